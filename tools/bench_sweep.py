"""Source-term sweep (BASELINE.json configs[4]) on one GPU: this GPU's share of 64 timesteps over 8 GPUs = 8 timesteps x
6 solves (2 operators x 3 right-hand sides) at 1024x512, each to max_iter, with eta and the EE scalars.
usage: python tools/bench_sweep.py [n_timesteps] [max_iter]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import xee_pkg
xee = xee_pkg.load()
from xlab_ee_fortran_amd import sweep

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
sweep.run(1024, 512, 1, max_iter=200)                       # warm-up (library load, allocations)
t0 = time.perf_counter()
rec, _ = sweep.run(1024, 512, nt, max_iter=max_iter)
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"source-term sweep: {nt} timesteps x 6 solves, 1024x512, max_iter {max_iter}", "cases": len(rec),
                  "seconds": dt, "cases_per_s": len(rec) / dt, "all_ran_to_max_iter": bool(np.all(rec["iters"] == max_iter)),
                  "efficiency_first_cases": [float(x) for x in rec["efficiency"][:6]]}))
