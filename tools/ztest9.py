"""k_sweep_z<nine> (BAROCLINIC, member pairs that share AP and CQ) against k_sweep_q<nine>, bit for bit (developer tool)"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import orchelp as H
import xee_pkg
xee = xee_pkg.load()
def biteq(a, b): return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
def members(nr, nz, M, share_b):
    A0, B0, C0, bc0 = H.rich_inputs(nr, nz, seed=nr + 7)
    rng = np.random.default_rng(3)
    out = []
    for m in range(M):
        B = B0 if share_b else (B0 + np.float32(1e-3 * (m + 1)) * rng.standard_normal(B0.shape)).astype(np.float32)
        out.append((A0, B, C0, (bc0 * np.float32(1.0 + 0.1 * m)).astype(np.float32)))
    return out
def solve(mem, n, opts, r1=1e-8):
    nz, nr = mem[0][0].shape
    with xee.Ensemble(nr, nz, len(mem)) as e:
        for k, v in opts.items(): e.set_option(k, v)
        e.set_profiles(xee.profiles(nr, nz))
        for m, (A, B, Cc, bc) in enumerate(mem):
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)): e.upload(name, arr, first=m)
        e.forcing_from_B(); e.assemble(1)
        rec = e.solve(n, r1, -1.0, 1.0)
        return e.download("PSI"), rec, e.timing(), e.sweep_info()
for (nr, nz, n, M, sb) in [(200, 150, 106, 2, 0), (333, 270, 231, 3, 0), (67, 40, 120, 5, 1), (1000, 33, 250, 2, 0), (57, 45, 66, 4, 0), (1024, 512, 208, 4, 0)]:
    mem = members(nr, nz, M, sb)
    want, rw, tw, iw = solve(mem, n, {"kernel": 3})
    for opts in ({"kernel": 4}, {"kernel": 4, "q_exact_only": 1}, {"kernel": 4, "z_pairs": 3}, {"kernel": 4, "sync_checks": 1}):
        got, rg, t, info = solve(mem, n, opts)
        ok = biteq(got, want) and rg["err_now"].tobytes() == rw["err_now"].tobytes() and rg["iters"].tolist() == rw["iters"].tolist()
        msg = "OK"
        if not ok:
            d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            msg = f"DIFF {len(d)} first {d[:4].tolist()} last {d[-2:].tolist()} err {rg['err_now'][:2]} {rw['err_now'][:2]}"
        print(nr, nz, n, "M", M, "shareB", sb, opts, info["name"], "z", t["z_launches"], "of", t["sweep_launches"], "shared", t["shared_planes"], "redo", t["inline_redo_tiles"], msg, flush=True)
