"""developer probe: bare sweep time on the fresh state vs after 40,000 iterations (how much do rejected tiles cost?)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()
nr, nz, M = 1024, 512, 32
with xee.Ensemble(nr, nz, M) as e:
    e.set_profiles(xee.profiles(nr, nz))
    for m in range(M):
        A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
        for n, a in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
            e.upload(n, a, first=m)
    e.forcing_from_B(); e.assemble(0)
    e.solve(100, 1e-8, -1.0, 1.0)
    for tag in ("fresh", "after 40k"):
        for _ in range(3):
            ms, n = e.time_sweeps(480, 1)
        print(tag, f"{ms / n * 1e3:.1f} us/launch", e.timing()["redo_tiles"])
        if tag == "fresh":
            e.solve(40000, 1e-8, -1.0, 1.0, flags=2)      # KEEP_PSI: continue from the current field
            print("redo tiles at the last check:", e.timing()["redo_tiles"], "solve sweep avg us", e.timing()["sweep_ms"] * 1e3 / e.timing()["sweep_launches"])
