"""Sweep-only timing probe (developer tool): bare blocked sweeps between HIP events, per kernel / tile / T option.
usage: python tools/probe_sweep.py [workload ...] [-- key=value,key=value ...]
       workloads: ensemble baroclinic single highres; option sets after `--` replace CONFIGS below
Each option set is timed ROUNDS times, round-robin (clock ramp and thermal state spread evenly); the median is printed."""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()

SHAPES = {"ensemble": (1024, 512, 32, 0, 480), "baroclinic": (1024, 512, 32, 1, 480), "single": (1024, 512, 1, 0, 3840),
          "highres": (4096, 2048, 1, 0, 480), "ens16": (1024, 512, 16, 0, 480), "ens8": (1024, 512, 8, 0, 480), "ens64": (1024, 512, 64, 0, 240)}
CONFIGS = [{}, {"tb_T": 5}, {"tb_T": 7}, {"tb_T": 8}, {"tb_T": 10}, {"tile": 806}, {"tile": 806, "tb_T": 8}, {"kernel": 2}]
ROUNDS = 3


def main():
    args = sys.argv[1:]
    cfgs = CONFIGS
    if "--" in args:
        i = args.index("--")
        cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in args[i + 1:]]
        args = args[:i]
    for wl in (args or ["ensemble"]):
        nr, nz, M, variant, iters = SHAPES[wl]
        with xee.Ensemble(nr, nz, M) as e:
            e.set_profiles(xee.profiles(nr, nz))
            for m in range(M):
                A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
                for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                    e.upload(name, arr, first=m)
            e.forcing_from_B(); e.assemble(variant)
            for _ in range(3):
                e.time_sweeps(iters, 1)                   # warm the clocks
            times = {i: [] for i in range(len(cfgs))}
            infos = {}
            for _ in range(ROUNDS):
                for i, opts in enumerate(cfgs):
                    for k in ("tb_T", "kernel", "tile", "z_pairs", "share_planes"):
                        e.set_option(k, {"tb_T": 8, "share_planes": 1}.get(k, 0))
                    for k, v in opts.items():
                        e.set_option(k, v)
                    ms, n = e.time_sweeps(iters, 1)
                    times[i].append((ms, n))
                    infos[i] = e.sweep_info()
            pts = (nr - 2) * (nz - 2) * M
            for i, opts in enumerate(cfgs):
                ms = statistics.median(t[0] for t in times[i]); n = times[i][0][1]
                info = infos[i]
                print(f"{wl:10s} {str(opts):28s} {info['name']:16s} T={info['T']:2d} tile {info['tile']:6s}: {ms / iters * 1e3:7.2f} us/iter "
                      f"{ms / n * 1e3:7.1f} us/launch  {pts * iters / (ms * 1e-3) / 1e12:6.3f} T pt-upd/s  (min {min(t[0] for t in times[i]) / iters * 1e3:.2f})", flush=True)


if __name__ == "__main__":
    main()
