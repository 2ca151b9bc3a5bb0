"""The residual checks run pipelined behind the sweeps by default (the check of iteration k is evaluated
while the next block already iterates; results of members that stop come from a snapshot).  It must give
exactly what the synchronous driver (option "sync_checks") and the oracle give -- iteration counts, fields,
residues -- for members stopping at different checks, at max_iter on and off a check boundary."""
import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def run(A, B, Cc, bc, cfg, sync):
    return xee.diagnose_case(A, B, Cc, bc, options={"sync_checks": 1 if sync else 0}, **cfg)["BAROTROPIC"]


@pytest.mark.parametrize("max_iter", [20000, 437, 300, 57])
def test_pipelined_equals_synchronous_and_oracle(max_iter):
    nr, nz, M = 96, 64, 7
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    cfg = dict(r1=2e-4, r2=5e-3, max_iter=max_iter, alpha=1.0)
    a = run(A, B, Cc, bc, cfg, False)
    b = run(A, B, Cc, bc, cfg, True)
    assert np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["err"], b["err"])
    assert biteq(a["psi"], b["psi"]) and biteq(a["eta"], b["eta"])
    assert a["r1"].tobytes() == b["r1"].tobytes() and a["r2"].tobytes() == b["r2"].tobytes()
    for m in (0, 3, 6):
        want = H.diagnose_case(A[m], B[m], Cc[m], bc[m], **cfg)["BAROTROPIC"]
        assert int(a["iters"][m]) == want["iters"] and int(a["err"][m]) == want["err"]
        assert biteq(a["psi"][m], want["psi"])
        assert a["r1"][m].tobytes() == want["r1"].tobytes()
    if max_iter == 20000:
        assert len(set(int(v) for v in a["iters"])) > 1 and int(a["iters"].max()) < max_iter


@pytest.mark.parametrize("max_iter", [6400, 6437, 6600, 7057, 20000])
def test_graphed_small_batch_driver(max_iter):
    """Small batches with a long solve ahead run up to EIGHT check blocks per graph over a ring of 3 x 8 slots (solve_graphed).
    Members that stop at every slot index of the ring (and while later blocks of their graph are already enqueued), caps
    on and off a check boundary, check counts that are not a multiple of four: same bits as the one-block-at-a-time
    driver (option graph_blocks = 0), the synchronous driver and the oracle; XEE_F_RESIDUAL = each member's last check."""
    nr, nz, M = 96, 64, 12
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    # per-member tolerances cannot be set, so members differ by their inputs; r1 chosen so that stops spread over many checks
    cfg = dict(r1=1.5e-4, r2=4e-3, max_iter=max_iter, alpha=1.0)
    outs = {}
    for tag, opts in (("graphed", {}), ("graphed4e", {"graph_blocks": 4, "graph_events": 1}), ("graphed2", {"graph_blocks": 2}),
                      ("single", {"graph_blocks": 0}), ("sync", {"sync_checks": 1})):
        with xee.Ensemble(nr, nz, M) as e:
            for kk, v in opts.items():
                e.set_option(kk, v)
            e.set_profiles(xee.profiles(nr, nz))
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                e.upload(name, arr)
            e.forcing_from_B(); e.assemble(xee.BAROTROPIC)
            rec = e.solve(cfg["max_iter"], cfg["r1"], cfg["r2"], cfg["alpha"])
            outs[tag] = (rec.copy(), e.download("PSI"), e.download("RESIDUAL"), e.timing())
    ra, pa, resa, _ = outs["graphed"]
    for tag in ("graphed4e", "graphed2", "single", "sync"):
        rb, pb, resb, _ = outs[tag]
        assert np.array_equal(ra["iters"], rb["iters"]) and np.array_equal(ra["err_bits"], rb["err_bits"]), (tag, ra["iters"], rb["iters"])
        assert ra["err_now"].tobytes() == rb["err_now"].tobytes() and ra["ratio"].tobytes() == rb["ratio"].tobytes(), tag
        assert biteq(pa, pb), tag
        assert biteq(resa[:, 1:-1, 1:-1], resb[:, 1:-1, 1:-1]), tag
    for m in (0, 5, 11):
        want = H.diagnose_case(A[m], B[m], Cc[m], bc[m], **cfg)["BAROTROPIC"]
        assert int(ra["iters"][m]) == want["iters"] and int(ra["err_bits"][m]) == want["err"]
        assert biteq(pa[m], want["psi"]) and ra["err_now"][m].tobytes() == want["r1"].tobytes()
    if max_iter == 20000:
        stops = sorted(set(int(v) // 100 for v in ra["iters"] if v < max_iter))
        assert len(set(c % 24 for c in stops)) >= 6, stops       # stops land on many slots of the ring


@pytest.mark.parametrize("serial", [0, 1])
def test_large_batch_split_halves_and_serial_norm(serial):
    """A batch above the large-batch threshold (4 Mi points): by default its member range is solved as two halves side by
    side (option split_halves, from 32 members on), checks one deep behind the sweeps.  33 members (halves of 17 and 16) that stop at different
    checks; the two-halves driver, the one-stream driver and the synchronous driver give the same bits -- also with option
    exact_sum_serial, which keeps the large batch's depth and only moves the norm onto the check stream."""
    nr, nz, M = 1024, 512, 33
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    # (the members' residuals decline slowly past 4.7e-3 .. 4.9e-3 at different iterations: an absolute criterion spreads the stops)
    cfg = dict(r1=(4.9e-3 if serial else 4.74e-3), r2=-1.0, max_iter=(500 if serial else 3000), alpha=1.0)
    outs = {}
    for tag, opts in (("halves", {}), ("one", {"split_halves": 0}), ("sync", {"sync_checks": 1})):
        with xee.Ensemble(nr, nz, M) as e:
            e.set_option("exact_sum_serial", serial)
            for kk, v in opts.items():
                e.set_option(kk, v)
            e.set_profiles(xee.profiles(nr, nz))
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                e.upload(name, arr)
            e.forcing_from_B(); e.assemble(xee.BAROTROPIC)
            rec = e.solve(cfg["max_iter"], cfg["r1"], cfg["r2"], cfg["alpha"], converge_time=1)
            outs[tag] = (rec.copy(), e.download("PSI"), e.download("RESIDUAL"), e.timing())
    ra, pa, resa, ta = outs["halves"]
    assert ta["sweep_streams"] == 2 and outs["one"][3]["sweep_streams"] == 1
    # the pipelined drivers stream this batch's plain launches (k_sweep_z, members leaving at different checks included); the
    # synchronous driver is k_sweep_q throughout
    assert ta["z_launches"] > 0 and outs["one"][3]["z_launches"] > 0 and outs["sync"][3]["z_launches"] == 0
    for tag in ("one", "sync"):
        rb, pb, resb, _ = outs[tag]
        assert np.array_equal(ra["iters"], rb["iters"]) and np.array_equal(ra["err_bits"], rb["err_bits"]), (tag, ra["iters"], rb["iters"])
        assert ra["err_now"].tobytes() == rb["err_now"].tobytes() and ra["ratio"].tobytes() == rb["ratio"].tobytes(), tag
        assert biteq(pa, pb), tag
        assert biteq(resa[:, 1:-1, 1:-1], resb[:, 1:-1, 1:-1]), tag
    if not serial:
        assert len(set(int(v) for v in ra["iters"])) >= 3, ra["iters"]        # members stop at different checks

def test_pipelined_other_check_steps():
    """check_step that is not a multiple of the blocked launch length, tiny converge_time."""
    nr, nz = 80, 56
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=4)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
    f = (-B).astype(np.float32)
    for cs, ct, mi in ((7, 3, 5000), (25, 2, 1013), (1, 4, 40)):
        want = H.solve(bc, coe, f, mi, 1e-6, 5e-2, 1.0, check_step=cs, converge_time=ct)
        got = xee.solve_elliptic(bc, coe, f, mi, 1e-6, 5e-2, 1.0, check_step=cs, converge_time=ct)
        assert got[1] == want[1] and got[2] == want[2], (cs, ct, mi, got[1:3], want[1:3])
        assert biteq(got[0], want[0])
        assert got[3].tobytes() == want[3].tobytes() and got[4].tobytes() == want[4].tobytes()


def _oracle_residual(A, B, Cc, bc, cfg, c):
    """residual plane of check iteration c: do_elliptic(psi_{c-1}) - f on the interior (elliptic_tools.f90:189-190)"""
    import ctypes as C
    o = H.diagnose_case(A, B, Cc, bc, **dict(cfg, max_iter=c - 1))["BAROTROPIC"]
    assert o["iters"] == c - 1, "the case must not stop before the check whose residual is wanted"
    nz, nr = A.shape
    out = np.zeros((nz, nr), np.float32)
    err = C.c_int(0)
    H.oracle().orc_do_elliptic(o["psi"], o["coe"], out, nr, nz, C.byref(err))
    f = (-B).astype(np.float32)
    out[1:-1, 1:-1] = out[1:-1, 1:-1] - f[1:-1, 1:-1]
    return out


@pytest.mark.parametrize("sync", [0, 1])
@pytest.mark.parametrize("max_iter", [200, 300, 437])
def test_residual_field_is_the_last_checks_plane(sync, max_iter):
    """XEE_F_RESIDUAL after a solve = the residual plane of each member's last check, in both drivers (the pipelined
    one alternates two planes): members that stop at different checks, and a cap on / off a check boundary."""
    nr, nz, M = 96, 64, 5
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    for crit in (dict(r1=1e-8, r2=-1.0), dict(r1=2e-4, r2=5e-3)):
        cfg = dict(max_iter=max_iter if crit["r2"] < 0 else 20000, alpha=1.0, **crit)
        with xee.Ensemble(nr, nz, M) as e:
            e.set_option("sync_checks", sync)
            e.set_profiles(xee.profiles(nr, nz))
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                e.upload(name, arr)
            e.forcing_from_B()
            e.assemble(xee.BAROTROPIC)
            rec = e.solve(cfg["max_iter"], cfg["r1"], cfg["r2"], cfg["alpha"])
            res = e.download("RESIDUAL")
        for m in (0, 2, 4):
            c = (int(rec["iters"][m]) // 100) * 100
            want = _oracle_residual(A[m], B[m], Cc[m], bc[m], cfg, c)
            assert biteq(res[m][1:-1, 1:-1], want[1:-1, 1:-1]), (sync, max_iter, crit, m, int(rec["iters"][m]))
