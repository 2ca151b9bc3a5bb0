"""Sweep-only timing probe (developer tool): bare blocked sweeps between HIP events, per kernel / tile / T option.
usage: python tools/probe_sweep.py [workload ...]      workloads: ensemble baroclinic single highres
       options tried per workload are listed in CONFIGS below (run-time options of include/xee.h)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()

SHAPES = {"ensemble": (1024, 512, 32, 0, 240), "baroclinic": (1024, 512, 32, 1, 240), "single": (1024, 512, 1, 0, 1920),
          "highres": (4096, 2048, 1, 0, 240)}
CONFIGS = [{}, {"tb_T": 4}, {"tb_T": 8}, {"kernel": 2}, {"kernel": 1}]


def main():
    for wl in (sys.argv[1:] or ["ensemble"]):
        nr, nz, M, variant, iters = SHAPES[wl]
        with xee.Ensemble(nr, nz, M) as e:
            e.set_profiles(xee.profiles(nr, nz))
            for m in range(M):
                A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
                for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                    e.upload(name, arr, first=m)
            e.forcing_from_B(); e.assemble(variant)
            for opts in CONFIGS:
                for k in ("tb_T", "kernel", "tile"):
                    e.set_option(k, {"tb_T": 6}.get(k, 0))
                for k, v in opts.items():
                    e.set_option(k, v)
                e.time_sweeps(iters, 1)
                ms, n = e.time_sweeps(iters, 1)
                info = e.sweep_info()
                pts = (nr - 2) * (nz - 2) * M
                print(f"{wl:10s} {str(opts):28s} {info['name']:18s} T={info['T']:2d} tile {info['tile']:7s}: {ms / iters * 1e3:8.2f} us/iter "
                      f"{ms / n * 1e3:8.1f} us/launch  {pts * iters / (ms * 1e-3) / 1e12:6.3f} T point-updates/s", flush=True)


if __name__ == "__main__":
    main()
