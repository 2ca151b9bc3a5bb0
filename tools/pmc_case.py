"""PMC calibration run (developer tool): a known-traffic float4 copy, the streaming sweep (single-step
solve) and the temporally blocked sweep on the bench workload shape.  Usage: nr nz members"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, orchelp as H, xee_pkg
xee = xee_pkg.load()
nr, nz, M = (int(v) for v in sys.argv[1:4])
A, B, Cc, bc = H.test1_inputs(nr, nz)
with xee.Ensemble(nr, nz, M) as e:
    e.set_profiles(xee.profiles(nr, nz))
    for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
        for m in range(M):
            e.upload(name, arr, first=m)
    e.forcing_from_B(); e.assemble(0)          # k_negate: float4 copy of M planes (known bytes)
    e.solve(96, 1e-8, -1.0, 1.0)               # 12 blocked launches (T=8)
    e.solve(20, 1e-8, -1.0, 1.0, flags=1)      # 20 streaming launches
    e.cal_eta()
