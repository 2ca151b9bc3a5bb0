"""Developer probe: does splitting the 32-case batch into K resident ensembles driven by K host threads fill the tail of each
sweep launch (7744 tiles = 15.1 rounds of 512 workgroups)?   usage: python tools/two_batch_probe.py [iters=20000]"""
import sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()
nr, nz, M = 1024, 512, 32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000


def make(first, count):
    e = xee.Ensemble(nr, nz, count)
    e.set_profiles(xee.profiles(nr, nz))
    for m in range(count):
        A, B, Cc, bc, Q = bench.member_fields(nr, nz, first + m)
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
            e.upload(name, arr, first=m)
    return e


def step(e):
    e.forcing_from_B(); e.assemble(0); e.solve(iters, 1e-8, -1.0, 1.0); e.cal_eta(); e.ee_scalars()


for K in (1, 2, 4, 1, 2):
    ens = [make(k * (M // K), M // K) for k in range(K)]
    for e in ens:
        step(e)                                   # warm
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ts = [threading.Thread(target=step, args=(e,)) for e in ens]
        for t in ts: t.start()
        for t in ts: t.join()
        best = min(best, time.perf_counter() - t0)
    print(f"K={K}: {best * 1e3:8.1f} ms for {M} cases x {iters} iterations -> {M / best * iters / 100000:6.2f} cases/s at 100,000", flush=True)
    for e in ens:
        e.close()
