"""TB kernel tile-config probe (developer tool).  usage: probe_tb.py  (reads XEE_TB_CFG from env)"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, orchelp as H, xee_pkg
xee = xee_pkg.load()
cfg = os.environ.get("XEE_TB_CFG", "8x16")
for (nr, nz, M, iters) in [(1024, 512, 32, 240), (1024, 512, 1, 960), (4096, 2048, 1, 240)]:
    A, B, Cc, bc = H.test1_inputs(nr, nz)
    for T in (4, 6, 8, 10, 12):
        os.environ["XEE_TB_T"] = str(T)
        with xee.Ensemble(nr, nz, M) as e:
            e.set_profiles(xee.profiles(nr, nz))
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                for m in range(M):
                    e.upload(name, arr, first=m)
            e.forcing_from_B(); e.assemble(0)
            e.time_sweeps(iters, 1)
            ms, n = e.time_sweeps(iters, 1)
            pts = (nr - 2) * (nz - 2) * M
            print(f"cfg {cfg:5s} {nr}x{nz} M={M:3d} T={T:2d}: {ms/iters*1e3:8.2f} us/iter {ms/n*1e3:8.1f} us/launch  {48.0*pts*iters/(ms*1e-3)/1e12:6.2f} TB/s effective", flush=True)
