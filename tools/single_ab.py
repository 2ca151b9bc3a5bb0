"""A/B of the small-batch solve driver on ONE box (developer tool): the `single` workload (one 1024x512 case, 100,000
iterations) under several option sets, round-robin, median of ROUNDS.
usage: python tools/single_ab.py [key=value,key=value ...]"""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()
import os
ROUNDS = int(os.environ.get("AB_ROUNDS", "3"))
DEFAULTS = {"graph_blocks": 8, "graph_events": 0, "sync_checks": 0, "tb_T": 8}


def main():
    import os
    if os.environ.get("AB_TORCH"):             # the bench's process also holds torch's HIP context
        import torch
        torch.cuda.set_device(0); torch.cuda.synchronize()
    cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
    nr, nz = 1024, 512
    with xee.Ensemble(nr, nz, 1) as e:
        e.set_profiles(xee.profiles(nr, nz))
        A, B, Cc, bc, Q = bench.member_fields(nr, nz, 0)
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
            e.upload(name, arr)
        e.forcing_from_B(); e.assemble(0)
        e.solve(20000, 1e-8, -1.0, 1.0)
        times = {i: [] for i in range(len(cfgs))}
        for _ in range(ROUNDS):
            for i, opts in enumerate(cfgs):
                for k, v in DEFAULTS.items():
                    e.set_option(k, v)
                for k, v in opts.items():
                    e.set_option(k, v)
                e.solve(100000, 1e-8, -1.0, 1.0)
                times[i].append(e.timing()["solve_ms"])
        for i, opts in enumerate(cfgs):
            print(f"{str(opts):48s} solve_ms median {statistics.median(times[i]):8.2f}  min {min(times[i]):8.2f}  ({times[i]})", flush=True)


if __name__ == "__main__":
    main()
