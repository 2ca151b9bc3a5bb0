"""developer probe: which member / which EE integral falls off the signed exact sum's fast path on the bench inputs?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench, xee_pkg, numpy as np
xee = xee_pkg.load()
nr, nz, M = 1024, 512, 32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
with xee.Ensemble(nr, nz, M) as e:
    e.set_profiles(xee.profiles(nr, nz))
    for m in range(M):
        A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
        for n, a in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
            e.upload(n, a, first=m)
    e.forcing_from_B(); e.assemble(0); e.solve(iters, 1e-8, -1.0, 1.0); e.cal_eta(); rec = e.ee_scalars()
    sq, sqe = e.ee_slow_chunks()
    print("chunks per integral:", (nr - 1) * -(-(nz - 1) // 32))
    print("sum_Q slow chunks   :", sq.tolist())
    print("sum_Qeta slow chunks:", sqe.tolist())
    print("sum_Qeta:", rec["sum_Qeta"][:8])
