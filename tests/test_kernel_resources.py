"""No instantiation of the blocked sweep may use scratch memory (a kernel with scratch is throttled at dispatch: the
residual-writing variant once took 158 us instead of 105, and the nine-weight body lost 2 % to a 32-byte spill that went
unnoticed for several commits), and the instances that are launched with two workgroups per CU must fit 128 VGPRs.
Compiles csrc/xee_sweepq.hip for gfx950 with the optimisation remarks on (no GPU needed, ~40 s)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "xlab-ee-fortran_amd" / "csrc" / "xee_sweepq.hip"


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_sweep_kernels_use_no_scratch(tmp_path):
    text = (ROOT / "xlab-ee-fortran_amd" / "Makefile").read_text().replace("\\\n", " ")
    m = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", text, re.M)
    assert m, "HIPFLAGS not found in the Makefile"
    flags = [f.replace("$(ARCH)", "gfx950") for f in m.group(1).split()]
    cmd = ["hipcc", *flags, "-fno-slp-vectorize", "-c", str(SRC), "-o", str(tmp_path / "q.o"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    recs, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); recs[name] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                recs[name][key] = int(m.group(1))
    sweeps = {k: v for k, v in recs.items() if "k_sweep_q" in k and "vgpr" in v}
    assert len(sweeps) >= 24, sorted(recs)
    for k, v in sweeps.items():
        assert v["scratch"] == 0, (k, v)
    # two workgroups of 8 waves per CU = 4 waves per SIMD = at most 128 VGPRs: the star at every tile, the nine-weight body at
    # six and four rows per lane (template arguments NW, R, WRITE_R, BARO: ILi8ELi<R>ELb<W>ELb<B>)
    for k, v in sweeps.items():
        m = re.search(r"k_sweep_qILi8ELi(\d)ELb[01]ELb([01])", k)
        if m and (m.group(2) == "1" or int(m.group(1)) < 8):
            assert v["vgpr"] <= 128, (k, v)
