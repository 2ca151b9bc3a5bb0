"""Developer probe: whole solves with the streaming k_sweep_z forced (option kernel = 4) against the tiled k_sweep_q (kernel = 3) for
several batch shapes on one box -- where the automatic choice's threshold (z_decide, xee_api.hip) comes from.
usage: python tools/auto_vs_tiled.py [baroclinic]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()
VARIANT = 1 if "baroclinic" in sys.argv[1:] else 0       # (the nine-weight operator streams for members that share AP and CQ)
SHAPES = (((1024, 512, 32, 5000), (1024, 512, 16, 5000), (1024, 512, 12, 5000), (1024, 512, 8, 5000), (2048, 1024, 4, 3000), (2048, 1024, 8, 3000), (512, 256, 64, 5000))
          if VARIANT else ((4096, 2048, 1, 2000), (1024, 512, 16, 5000), (1024, 512, 24, 5000), (2048, 1024, 4, 3000), (2048, 1024, 8, 3000), (1024, 512, 8, 5000)))
for (nr, nz, M, iters) in SHAPES:
    with xee.Ensemble(nr, nz, M) as e:
        e.set_profiles(xee.profiles(nr, nz))
        for m in range(M):
            A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                e.upload(name, arr, first=m)
        res = {}
        for rnd in range(3):
            for k in (4, 3):
                e.set_option("kernel", k)
                e.forcing_from_B(); e.assemble(VARIANT)
                t0 = time.perf_counter(); e.solve(iters, 1e-8, -1.0, 1.0); dt = time.perf_counter() - t0
                if rnd: res.setdefault(k, []).append((dt, e.timing()["z_launches"]))
        print(nr, nz, M, iters, "k_sweep_z", round(min(r[0] for r in res[4]) * 1e3, 2), "ms (z launches", res[4][0][1], ")  k_sweep_q", round(min(r[0] for r in res[3]) * 1e3, 2), "ms", flush=True)
