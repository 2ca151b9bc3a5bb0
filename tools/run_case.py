"""Run one solve of a synthetic ensemble (developer tool for profiling): nr nz members max_iter"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, orchelp as H, xee_pkg
xee = xee_pkg.load()
nr, nz, M, mi = (int(v) for v in sys.argv[1:5])
A, B, Cc, bc = H.test1_inputs(nr, nz)
with xee.Ensemble(nr, nz, M) as e:
    e.set_profiles(xee.profiles(nr, nz))
    for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
        for m in range(M):
            e.upload(name, arr, first=m)
    e.forcing_from_B(); e.assemble(0)
    for _ in range(2):
        e.solve(mi, 1e-8, -1.0, 1.0)
        print(e.timing())
