"""The Tier-1 entry points keep their one-member device ensemble per thread and reuse it while the grid stays the same
(include/xee.h, xee_tier1_release).  A reused ensemble must behave exactly like a fresh one: interleaved calls of different
kinds on the same grid, a change of grid, changed process defaults, a failing call and an explicit release in between --
every result equal to the oracle's (and to the first call's)."""
import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _case(nr, nz, seed):
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=seed)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
    return dict(bc=bc, coe=coe, f=(-B).astype(np.float32), prof=prof, sabc=(sA, sB, sC), nr=nr, nz=nz)


def _solve_ok(c, mi=300, alpha=1.0):
    want = H.solve(c["bc"], c["coe"], c["f"], mi, 1e-8, -1.0, alpha)
    got = xee.solve_elliptic(c["bc"], c["coe"], c["f"], mi, 1e-8, -1.0, alpha)
    assert biteq(got[0], want[0]) and got[1] == want[1] and got[2] == want[2]
    assert got[3].tobytes() == want[3].tobytes() and got[4].tobytes() == want[4].tobytes()
    return got


def test_reused_tier1_ensemble_behaves_like_a_fresh_one():
    xee.tier1_release()
    a, b, other = _case(96, 72, 1), _case(96, 72, 2), _case(80, 56, 3)
    first = _solve_ok(a)
    # other kinds of calls on the same grid leave planes, flags and profiles behind in the kept ensemble
    assert biteq(xee.cal_eta(b["bc"], b["prof"]), H.cal_eta(b["bc"], b["prof"]))
    u, w = xee.cal_uw(a["f"], a["prof"]); uo, wo = H.cal_uw(a["f"], a["prof"])
    assert biteq(u, uo) and biteq(w, wo)
    want = np.full((72, 96), -3.5, np.float32)
    H.oracle().orc_do_elliptic(b["bc"], b["coe"], want, 96, 72, __import__("ctypes").c_int(0))
    out, _ = xee.do_elliptic(b["bc"], b["coe"], outdat=np.full((72, 96), -3.5, np.float32))   # (its rim keeps the caller's bytes)
    assert biteq(out, want)
    coe_g, err = xee.cal_coe(*b["sabc"], b["prof"]["dr"], b["prof"]["dz"], 96, 72)
    assert biteq(coe_g, b["coe"]) and err == 0
    _solve_ok(b, mi=250, alpha=0.8)                       # the coe this thread's last cal_coe made: fast kernel
    assert xee.last_sweep_info()["kernel"] == "k_sweep_q"
    again = _solve_ok(a)                                   # any other coe: nine-plane kernel, same bits as the first time
    assert biteq(again[0], first[0]) and again[1:3] == first[1:3]
    # another grid replaces the kept ensemble; coming back rebuilds it
    _solve_ok(other)
    _solve_ok(a)
    # process defaults are read again by a reused ensemble
    xee.set_default_option("tb_T", 3)
    try:
        _solve_ok(a)
        assert xee.last_sweep_info()["T"] == 3
    finally:
        xee.reset_default_options()
    _solve_ok(a)
    assert xee.last_sweep_info()["T"] == 8
    # a failing call drops the kept ensemble; the next one starts afresh
    with pytest.raises(xee.XeeError):
        xee.solve_elliptic(a["bc"], a["coe"], a["f"], 10, -1.0, 0.0, 1.0)
    _solve_ok(a)
    xee.tier1_release()
    _solve_ok(a)
    xee.tier1_release()


def test_repeated_tier1_calls_do_not_pay_for_a_new_ensemble():
    """counted, not timed: repeated Tier-1 calls on one grid create no further device ensemble and allocate no further
    staging buffer (xee_debug_counters)"""
    c = _case(64, 48, 7)
    xee.solve_elliptic(c["bc"], c["coe"], c["f"], 200, 1e-8, -1.0, 1.0)
    before = xee.debug_counters()
    for _ in range(5):
        xee.solve_elliptic(c["bc"], c["coe"], c["f"], 200, 1e-8, -1.0, 1.0)
        xee.cal_coe(*c["sabc"], c["prof"]["dr"], c["prof"]["dz"], 64, 48)
    after = xee.debug_counters()
    assert after["ensembles"] == before["ensembles"], (before, after)
    assert after["tier1_stage_allocs"] == before["tier1_stage_allocs"], (before, after)
    xee.tier1_release()
    xee.solve_elliptic(c["bc"], c["coe"], c["f"], 200, 1e-8, -1.0, 1.0)
    assert xee.debug_counters()["ensembles"] == after["ensembles"] + 1
    xee.tier1_release()


def test_tier1_from_two_threads_at_once():
    """SURVEY.md 8(b) "Threading": the reference's procedures are stateless, so a drop-in may be called from several host
    threads at once.  Two threads run cal_coe -> solve_elliptic -> cal_eta loops concurrently on DIFFERENT grids (each
    thread keeps its own device ensemble), a third flips a process default meanwhile; every result equals the serial one
    bit for bit, and a thread that ends gives its device memory back without being asked to."""
    import threading
    cases = {"a": _case(112, 80, 11), "b": _case(200, 96, 12)}
    serial = {}
    for name, c in cases.items():
        coe, _ = xee.cal_coe(*c["sabc"], c["prof"]["dr"], c["prof"]["dz"], c["nr"], c["nz"])
        serial[name] = (coe, xee.solve_elliptic(c["bc"], coe, c["f"], 400, 1e-8, -1.0, 0.9), xee.cal_eta(c["bc"], c["prof"]))
        want = H.solve(c["bc"], c["coe"], c["f"], 400, 1e-8, -1.0, 0.9)
        assert biteq(serial[name][1][0], want[0]) and biteq(coe, c["coe"])
    xee.tier1_release()
    errors, results = [], {"a": [], "b": []}
    stop = threading.Event()

    def worker(name):
        try:
            c = cases[name]
            for _ in range(6):
                coe, err = xee.cal_coe(*c["sabc"], c["prof"]["dr"], c["prof"]["dz"], c["nr"], c["nz"])
                got = xee.solve_elliptic(c["bc"], coe, c["f"], 400, 1e-8, -1.0, 0.9)
                results[name].append((coe, got, xee.cal_eta(c["bc"], c["prof"]), xee.last_sweep_info()["kernel"]))
        except Exception as exc:              # noqa: BLE001
            errors.append((name, repr(exc)))

    def flipper():
        while not stop.is_set():
            xee.set_default_option("elision", 0); xee.set_default_option("elision", 1)

    before = xee.debug_counters()
    ts = [threading.Thread(target=worker, args=(n,)) for n in ("a", "b")]
    fl = threading.Thread(target=flipper)
    fl.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set(); fl.join()
    xee.reset_default_options()
    assert not errors, errors
    for name in ("a", "b"):
        coe0, sol0, eta0 = serial[name]
        for coe, got, eta, kern in results[name]:
            assert biteq(coe, coe0) and biteq(got[0], sol0[0]) and got[1:3] == sol0[1:3] and biteq(eta, eta0), name
            assert got[3].tobytes() == sol0[3].tobytes() and kern == "k_sweep_q"
    after = xee.debug_counters()
    assert after["ensembles"] - before["ensembles"] == 2            # one kept ensemble per thread, reused across its calls
    # both threads ended: their caches are gone (join() returns when the Python side of a thread is done; the OS thread, whose
    # exit runs the C++ thread_local destructors, may take a moment longer)
    import time
    for _ in range(200):
        if xee.debug_counters()["thread_exit_releases"] - before["thread_exit_releases"] == 2:
            break
        time.sleep(0.01)
    assert xee.debug_counters()["thread_exit_releases"] - before["thread_exit_releases"] == 2
