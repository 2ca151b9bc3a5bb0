#!/usr/bin/env python3
"""Deep md5 anchors at BASELINE.json's full size, made by RUNNING THE REFERENCE (oracle/_ref/diagnose,
the unmodified sources compiled by oracle/Makefile; this container only).  Merged into
golden_hashes.json next to the anchors of make_golden.py.

  * m1024x512_100000 / m1024x512_baroclinic_100000  -- the protocol of SURVEY.md 8(d) config 2 at its full
    depth (src/diagnose/diagnose.f90:15-21 / :42-48), test1 formula at 1024x512, r1=1e-8, r2 off, alpha=1.
  * bench_m0_100000 / bench_m0_baroclinic_100000    -- member 0 of bench.py's ensemble (SURVEY.md 8d synthetic
    member formula) at the bench's own depth: bench.py prints the md5 of member 0's psi / eta after its
    timed region and tests/test_gpu_bench_contract.py ties that line to these anchors.
  * member{m}_1024x512_600[_baroclinic]              -- noisy members (m>0) at full size for the batched-vs-
    reference test (tests/test_gpu_fullsize.py).
  * m1024x512_baroclinic_2000                         -- full-B operator anchor at moderate depth.
  * member{m}_1024x512_200                            -- anchors for the whole 256-member batch of configs[2].

One reference process per case (the reference is single-threaded): ~40-50 min per 100,000-iteration case.
Usage: python tests/golden/make_golden_deep.py [--jobs 4] [--only name,name]
"""
import argparse
import json
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orchelp as H  # noqa: E402
from make_golden import RELAX  # noqa: E402

OUT = Path(__file__).resolve().parent
BT = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC")
BC = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC")


def cases():
    c = {}
    for tag, mode in (("", BT), ("_baroclinic", BC)):
        c[f"m1024x512{tag}_100000"] = ("test1", 0, mode, 100000)
        c[f"bench_m0{tag}_100000"] = ("member", 0, mode, 100000)
    c["m1024x512_baroclinic_2000"] = ("test1", 0, BC, 2000)
    for m in (1, 2, 3, 7):
        c[f"member{m}_1024x512_600"] = ("member", m, BT, 600)
    for m in (1, 6):
        c[f"member{m}_1024x512_600_baroclinic"] = ("member", m, BC, 600)
    for m in (1, 2, 3, 7, 255):          # the 256-member ensemble of configs[2] in ONE batch (tests/test_gpu_fullsize.py)
        c[f"member{m}_1024x512_200"] = ("member", m, BT, 200)
    # the highres workload of bench.py (configs[3]): member 0 at 4096x2048, 2,000 iterations (~25 min of the reference)
    c["bench_m0_4096x2048_2000"] = ("member4096", 0, BT, 2000)
    return c


def one(args):
    name, (kind, m, mode, iters) = args
    nr, nz = (4096, 2048) if kind == "member4096" else (1024, 512)
    A, B, Cc, bc = H.test1_inputs(nr, nz) if kind == "test1" else H.member_inputs(nr, nz, m)
    cfg = dict(mode=mode, r1=1e-8, r2=-1.0, max_iter=iters, alpha=1.0)
    with tempfile.TemporaryDirectory() as td:
        stdout, outs = H.run_ref_diagnose(td, A, B, Cc, bc, None, **cfg)
    rec = {"relax": [(int(a), b, c) for a, b, c in RELAX.findall(stdout)],
           "md5": {k: H.md5(v) for k, v in outs.items()},
           "cfg": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
           "inputs": [kind, m], "shape": [nr, nz],
           "input_md5": {"A": H.md5(A), "B": H.md5(B), "C": H.md5(Cc), "bc_init": H.md5(bc)}}
    print(name, rec["relax"], flush=True)
    return name, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = cases()
    if a.only:
        todo = {k: v for k, v in todo.items() if k in a.only.split(",")}
    # short cases first so a partial run already leaves anchors behind
    order = sorted(todo.items(), key=lambda kv: kv[1][3])
    path = OUT / "golden_hashes.json"
    with ProcessPoolExecutor(a.jobs) as ex:
        for name, rec in ex.map(one, order):
            hashes = json.loads(path.read_text())
            hashes[name] = rec
            path.write_text(json.dumps(hashes, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
