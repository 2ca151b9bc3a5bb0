"""k_sweep_z / k_sweep_zn against k_sweep_q, bit for bit (developer tool): python tools/ztest.py [pairs ...]"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import orchelp as H
import xee_pkg
xee = xee_pkg.load()
def biteq(a, b): return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
cases = [(200, 150, 106), (1024, 512, 112), (333, 270, 231), (67, 40, 120), (1000, 33, 250), (57, 45, 66)]
pairs = [int(a) for a in sys.argv[1:]] or [0, 1, 3]
MODES = {"BAROTROPIC": ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"),
         "BAROCLINIC": ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC")}
for (nr, nz, iters) in cases:
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=nr)
    for op, variant in (("BAROTROPIC", 0),):
        cfg = dict(r1=1e-8, r2=-1.0, max_iter=iters, alpha=1.0, mode=MODES[op])
        want = xee.diagnose_case(A, B, Cc, bc, options={"kernel": 3}, **cfg)[op]
        for P in pairs:
            for ex in (0, 1):
                got = xee.diagnose_case(A, B, Cc, bc, options={"kernel": 4, "z_pairs": P, "q_exact_only": ex, "tile": variant}, **cfg)[op]
                ok = biteq(got["psi"], want["psi"]) and got["iters"] == want["iters"]
                d = np.argwhere(got["psi"][0].view(np.uint32) != want["psi"][0].view(np.uint32))
                msg = "OK"
                if not ok:
                    rows = sorted(set(d[:, 0].tolist())); cols = sorted(set(d[:, 1].tolist()))
                    msg = f"DIFF {len(d)} dim0 {rows[:6]}..{rows[-3:]} (n={len(rows)}) dim1 {cols[:12]}..{cols[-3:]} (n={len(cols)})"
                t = got["timing"]
                print(nr, nz, iters, op[:6], "pairs", P, "exact" if ex else "fast ", got["sweep"]["name"], got["sweep"]["tile"], "z", t["z_launches"], "redo", t["inline_redo_tiles"], msg, flush=True)
