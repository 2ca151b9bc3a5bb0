#!/usr/bin/env python3
"""sha256 over what decides the blocked sweep's counters: csrc/xee_sweepq.hip, csrc/xee_sweepz.hip, csrc/xee_api.hip (the launch schedule),
csrc/xee_device.h and the HIPFLAGS line of the Makefile.  tools/profile_round.sh / pmc_only.sh record it on the GPU box
next to the PMC CSVs; tools/pmc_summarize.py stores it in profiles/traffic.json; bench.py recomputes it and quotes the
stored counters only when the two agree.
    python tools/build_hash.py            -> prints the hash of the working tree"""
import hashlib
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FILES = ("xlab-ee-fortran_amd/csrc/xee_sweepq.hip", "xlab-ee-fortran_amd/csrc/xee_sweepz.hip", "xlab-ee-fortran_amd/csrc/xee_api.hip",
         "xlab-ee-fortran_amd/csrc/xee_device.h")


def build_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in FILES:
        h.update(rel.encode())
        h.update((Path(root) / rel).read_bytes())
    text = (Path(root) / "xlab-ee-fortran_amd" / "Makefile").read_text().replace("\\\n", " ")
    m = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", text, re.M)
    h.update(b"HIPFLAGS")
    h.update(" ".join(m.group(1).split()).encode() if m else b"?")
    return h.hexdigest()


if __name__ == "__main__":
    print(build_hash())
