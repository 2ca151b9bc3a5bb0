"""Developer probe: option split_halves (a large batch as two halves side by side) on / off for several batch sizes, one box.
usage: python tools/split_probe.py [iters=20000] [members ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()
nr, nz = 1024, 512
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sizes = [int(a) for a in sys.argv[2:]] or [8, 16, 24, 32, 64]
for M in sizes:
    e = xee.Ensemble(nr, nz, M)
    e.set_profiles(xee.profiles(nr, nz))
    for m in range(M):
        A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
            e.upload(name, arr, first=m)
    res = {}
    zinfo = {}
    for rnd in range(3):
        for split in (1, 0):
            e.set_option("split_halves", split)
            t0 = time.perf_counter()
            e.forcing_from_B(); e.assemble(0); e.solve(iters, 1e-8, -1.0, 1.0); e.cal_eta(); e.ee_scalars()
            dt = time.perf_counter() - t0
            if rnd:
                res.setdefault(split, []).append(dt)
                t = e.timing()
                zinfo[split] = (t["z_launches"], t["sweep_launches"], round(t["z_ms"] / max(t["z_timed_launches"] / t["sweep_streams"], 1) * 1e3, 1))
    print(f"M={M:4d}: split {min(res[1]) * 1e3:8.1f} ms ({M / min(res[1]) * iters / 1e5:6.2f} cases/s at 100k)   one stream {min(res[0]) * 1e3:8.1f} ms "
          f"({M / min(res[0]) * iters / 1e5:6.2f})   (k_sweep_z launches, all launches, us per k_sweep_z launch): split {zinfo[1]}  one {zinfo[0]}", flush=True)
    e.close()
