"""Developer probe: a batch whose last members cannot stream (zero right-hand side: the rim-boundary kind of configs[4]) -- the
members before them on k_sweep_z beside the others on k_sweep_q, for several chunkings, against the tiled sweep for all.
usage: python tools/elig_split_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import bench, xee_pkg
xee = xee_pkg.load()
nr, nz, iters = 1024, 512, 5000
for (M, nbad) in ((24, 8), (48, 16)):
    for variant in (0, 1):
        with xee.Ensemble(nr, nz, M) as e:
            e.set_profiles(xee.profiles(nr, nz))
            for m in range(M):
                A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
                f = (-B).astype(np.float32) if m < M - nbad else np.zeros_like(B)
                for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("FORCING", f)):
                    e.upload(name, arr, first=m)
            e.assemble(variant)
            for opts in ({"kernel": 3}, {}, {"z_pairs": 4}, {"z_pairs": 5}, {"z_pairs": 6}, {"z_pairs": 8}):
                for k in ("kernel", "z_pairs"):
                    e.set_option(k, 0)
                for k, v in opts.items():
                    e.set_option(k, v)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter(); e.solve(iters, 1e-8, -1.0, 1.0); best = min(best, time.perf_counter() - t0)
                t = e.timing()
                print(M, nbad, "variant", variant, opts, round(best * 1e3, 1), "ms  streams", t["sweep_streams"], "z", t["z_launches"], "of", t["sweep_launches"], flush=True)
