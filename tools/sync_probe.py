"""developer probe: 2000 iterations of the bench ensemble with synchronous checks (kernel durations without overlap)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg
xee = xee_pkg.load()
nr, nz, M = 1024, 512, 32
with xee.Ensemble(nr, nz, M) as e:
    e.set_option("sync_checks", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    e.set_profiles(xee.profiles(nr, nz))
    for m in range(M):
        A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
        for n, a in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
            e.upload(n, a, first=m)
    e.forcing_from_B(); e.assemble(0)
    e.solve(500, 1e-8, -1.0, 1.0)
    e.solve(4000, 1e-8, -1.0, 1.0)
    print(e.timing())
