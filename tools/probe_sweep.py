"""Quick timing probe of the sweep kernels (developer tool, not part of the product)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, orchelp as H, xee_pkg
xee = xee_pkg.load()

def run(nr, nz, M, iters, variant=0, T=8):
    os.environ["XEE_TB_T"] = str(T)
    A, B, Cc, bc = H.test1_inputs(nr, nz)
    prof = xee.profiles(nr, nz)
    with xee.Ensemble(nr, nz, M) as e:
        e.set_profiles(prof)
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
            for m in range(M):
                e.upload(name, arr, first=m)
        e.forcing_from_B(); e.assemble(0)
        e.solve(200, 1e-8, -1.0, 1.0)
        t = e.timing()
        ms, n = e.time_sweeps(iters, variant)
        ms, n = e.time_sweeps(iters, variant)
        pts = (nr - 2) * (nz - 2) * M
        gbs = 48.0 * pts * iters / (ms * 1e-3) / 1e9
        print(f"{nr}x{nz} M={M:4d} variant={variant} T={T:2d}: {ms/iters*1e3:9.2f} us/iter {ms/n*1e3:9.2f} us/launch {gbs:8.1f} GB/s effective ({gbs/8000*100:6.1f}% of 8 TB/s)   solve200: {t['solve_ms']:.3f} ms sweep {t['sweep_ms']:.3f} ms", flush=True)

if __name__ == "__main__":
    cases = [(1024, 512, 1, 2000), (1024, 512, 8, 480), (1024, 512, 32, 240), (4096, 2048, 1, 240)]
    for (nr, nz, M, it) in cases:
        run(nr, nz, M, it, 0)
        for T in (4, 8, 12, 16):
            run(nr, nz, M, it, 1, T)
