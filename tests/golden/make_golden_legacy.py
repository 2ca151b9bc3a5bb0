#!/usr/bin/env python3
"""Generates tests/golden/legacy_efficiency_64x48.json by RUNNING the reference's legacy program
(oracle/_ref/old-diagnose, built by oracle/Makefile from src/old-diagnose; this container only).

Purpose: the legacy program is the only place the reference computes the scalar
energy-efficiency numbers (`sum Q`, `sum_Qeta`), so its `efficiency.txt` pins the oracle's integral
restatements (orc_integrate_weight_B, orc_cal_sum_Qeta) to the 8 significant digits the program
prints.  The legacy program has undefined behaviour: Q_in and F_in are
allocated (nr-1,nz-1) but read as nr x nz records.  On this small grid the overflow is inert, and
the field it really integrates is the SHEARED one, Q_eff(i,j) = file_flat[(j-1)*(nr-1) + (i-1)],
which the test emulates.  The fixture stores the inputs' recipe (seeded, regenerable), the
program's efficiency.txt text and the eta / rchi fields it wrote for the L(B=0) = B0 solve.
"""
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orchelp as H  # noqa: E402

OUT = Path(__file__).resolve().parent
NR, NZ = 64, 48


def inputs():
    A, B, Cc, _ = H.member_inputs(NR, NZ, 0)
    r = np.linspace(0, 1, NR)[None, :]
    z = np.linspace(0, 1, NZ)[:, None]
    Qfile = np.exp(-((r - 0.3) / 0.1) ** 2 - ((z - 0.5) / 0.2) ** 2).astype(np.float32)   # nr x nz record
    return A, B, Cc, Qfile


def main():
    A, B, Cc, Qfile = inputs()
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "out").mkdir()
        A.tofile(td / "A.bin"); B.tofile(td / "B.bin"); Cc.tofile(td / "C.bin"); Qfile.tofile(td / "Q.bin")
        np.zeros((NZ, NR), np.float32).tofile(td / "F.bin")
        (td / "diag.txt").write_text(
            "CYLINDRICAL-TENDENCY-DENSITY_NORMAL-BAROTROPIC\n100.0 // testing_dt\n0.0 1.0 0.0 1.0\n%d %d\n.\nout\n"
            "A.bin\nB.bin\nC.bin\nQ.bin\nF.bin\n1 1e-7 2000 1.0\n1 1e-7 2000 1.0\nno\nno\n" % (NR, NZ))
        with open(td / "diag.txt") as fin:
            p = subprocess.run([str(H.REF_DIR / "old-diagnose")], stdin=fin, cwd=td, capture_output=True, text=True, timeout=300)
        eff = (td / "out" / "efficiency.txt").read_text()
        eta = np.fromfile(td / "out" / "eta-[0_B0]-A.bin", np.float32)
        rchi = np.fromfile(td / "out" / "rchi-[0_B0]-O.bin", np.float32)
    print(p.stdout[-600:])
    print(eff)
    rec = {"nr": NR, "nz": NZ, "efficiency_txt": eff, "eta_0_B0_md5": H.md5(eta), "rchi_0_B0_md5": H.md5(rchi),
           "returncode": p.returncode}
    np.savez_compressed(OUT / "legacy_64x48_fields.npz", eta_0_B0=eta.reshape(NZ, NR - 1), rchi_0_B0=rchi.reshape(NZ, NR))
    (OUT / "legacy_efficiency_64x48.json").write_text(json.dumps(rec, indent=1) + "\n")


if __name__ == "__main__":
    main()
