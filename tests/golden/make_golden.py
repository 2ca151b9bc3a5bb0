#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz and golden_hashes.json by RUNNING THE REFERENCE built into
oracle/_ref/ (make -C oracle ref; this container only -- /root/reference never travels).

Each .npz holds one small case: the inputs handed to the reference program and every output file
it wrote, plus the "Relaxation uses N steps. Final residue is r1 , r2" line(s) of its stdout.
Big cases are pinned by md5 only (golden_hashes.json); BASELINE.md section 4 lists the same hashes
as measured by the survey, which this script re-derives independently.

Usage: python tests/golden/make_golden.py [--big]     (--big adds the 1024x512 / 4096x2048 hashes)
"""
import json
import re
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orchelp as H  # noqa: E402

OUT = Path(__file__).resolve().parent
RELAX = re.compile(r"Relaxation uses\s+(\d+)\s+steps\. Final residue is\s+(\S+)\s*,\s*(\S+)")


def run(name, A, B, Cc, bc, forcing=None, keep=True, **cfg):
    with tempfile.TemporaryDirectory() as td:
        stdout, outs = H.run_ref_diagnose(td, A, B, Cc, bc, forcing, **cfg)
    relax = [(int(a), b, c) for a, b, c in RELAX.findall(stdout)]
    rec = {"relax": relax, "md5": {k: H.md5(v) for k, v in outs.items()}, "cfg": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
           "shape": [int(A.shape[1]), int(A.shape[0])], "input_md5": {"A": H.md5(A), "B": H.md5(B), "C": H.md5(Cc), "bc_init": H.md5(bc)}}
    if keep:
        arrs = {"in_A": A, "in_B": B, "in_C": Cc, "in_bc": bc}
        if forcing is not None:
            arrs["in_forcing"] = forcing
        for k, v in outs.items():
            arrs["out_" + k] = v
        np.savez_compressed(OUT / f"{name}.npz", **arrs)
    print(name, relax, flush=True)
    return rec


def main():
    big = "--big" in sys.argv
    old = OUT / "golden_hashes.json"
    hashes = json.loads(old.read_text()) if old.exists() else {}     # keep what this run does not regenerate
                                                                    # (--big cases, make_golden_deep.py's anchors)
    # 1. plumbing case P: test1 formula at 128x64, test1 criteria -> 6,300 iterations
    A, B, Cc, bc = H.test1_inputs(128, 64)
    hashes["p128x64_test1"] = run("p128x64_test1", A, B, Cc, bc, r1=5e-3, r2=5e-3, max_iter=100000, alpha=1.0)
    # 2. both operator variants, every plane varying, fixed work (pins K1, K2, K3, K5)
    A, B, Cc, bc = H.rich_inputs(64, 48, seed=7)
    hashes["s64x48_baroall"] = run(
        "s64x48_baroall", A, B, Cc, bc, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BARO_ALL"),
        domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=300, alpha=0.9)
    # 3. secondary circulation + Boussinesq + forcing file (pins K6 and the forcing path)
    A, B, Cc, bc = H.rich_inputs(64, 48, seed=11)
    forcing = (1e-9 * np.random.default_rng(5).standard_normal((48, 64))).astype(np.float32)
    hashes["s64x48_seccirc"] = run(
        "s64x48_seccirc", A, B, Cc, bc, forcing,
        mode=("SECONDARY_CIRCULATION", "CYLINDRICAL", "DENSITY_BOUSSINESQ", "BAROCLINIC"),
        domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=250, alpha=1.0)
    # 4. the convergence state machine with both criteria live on a rich case
    A, B, Cc, bc = H.rich_inputs(64, 48, seed=13)
    hashes["s64x48_converge"] = run(
        "s64x48_converge", A, B, Cc, bc, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"),
        domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-6, r2=2e-3, max_iter=100000, alpha=1.0)
    # 5. spherical geometry (latitudes in degrees fed to cos, quirk 9)
    A, B, Cc, bc = H.rich_inputs(64, 48, seed=17)
    hashes["s64x48_spherical"] = run(
        "s64x48_spherical", A, B, Cc, bc, mode=("DYNAMIC_EFFICIENCY", "SPHERICAL", "DENSITY_NORMAL", "BAROTROPIC"),
        domain=(6371000.0, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=200, alpha=1.0)
    # 6. the shipped fixture itself (md5 only): 200x200 -> 17,900 iterations
    A, B, Cc, bc = H.test1_inputs(200, 200)
    hashes["t200x200_test1"] = run("t200x200_test1", A, B, Cc, bc, keep=False, r1=5e-3, r2=5e-3, max_iter=100000, alpha=1.0)
    hashes["t200x200_baroclinic300"] = run(
        "t200x200_baroclinic300", A, B, Cc, bc, keep=False,
        mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC"), r1=1e-8, r2=-1.0, max_iter=300, alpha=0.9)
    if big:
        A, B, Cc, bc = H.test1_inputs(1024, 512)
        hashes["m1024x512_500"] = run("m1024x512_500", A, B, Cc, bc, keep=False, r1=1e-8, r2=-1.0, max_iter=500, alpha=1.0)
        hashes["m1024x512_20000"] = run("m1024x512_20000", A, B, Cc, bc, keep=False, r1=1e-8, r2=-1.0, max_iter=20000, alpha=1.0)
        A, B, Cc, bc = H.test1_inputs(4096, 2048)
        hashes["l4096x2048_100"] = run("l4096x2048_100", A, B, Cc, bc, keep=False, r1=1e-8, r2=-1.0, max_iter=100, alpha=1.0)
    (OUT / "golden_hashes.json").write_text(json.dumps(hashes, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
