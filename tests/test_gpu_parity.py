"""GPU parity tests: the HIP path, called through the C-ABI (include/xee.h via ctypes), against the
oracle on the same inputs and against the reference's own golden outputs.  fp32 fields and scalars are
compared BIT FOR BIT -- the meaning of the north star's 1e-10 / 1e-12 tolerances for real(4) data."""
import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert xee.device_count() >= 1, "GPU tests need a device; libxee has no CPU fallback"


# ---- kernel level -----------------------------------------------------------------------------

@pytest.mark.parametrize("nr,nz", [(64, 48), (37, 29), (200, 200), (257, 33), (1024, 64)])
def test_solver_coefficients_and_cal_coe(nr, nz):
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=nr + nz)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    gA, gB, gC = xee.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    assert biteq(gA, sA) and biteq(gB, sB) and biteq(gC, sC)
    want = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
    sentinel = np.full((nz, nr, 9), 7.25, np.float32)
    got, err = xee.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz, coe=sentinel)
    assert err == 0
    assert biteq(got[1:-1, 1:-1], want[1:-1, 1:-1])
    # rim entries are not written (the reference leaves them untouched)
    assert np.all(got[0] == 7.25) and np.all(got[-1] == 7.25) and np.all(got[:, 0] == 7.25) and np.all(got[:, -1] == 7.25)


@pytest.mark.parametrize("nr,nz", [(64, 48), (130, 21)])
def test_do_elliptic(nr, nz):
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=5)
    prof = H.profiles(nr, nz)
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
    want = np.full((nz, nr), -3.5, np.float32)
    e = __import__("ctypes").c_int(0)
    H.oracle().orc_do_elliptic(bc, coe, want, nr, nz, e)
    got, err = xee.do_elliptic(bc, coe, outdat=np.full((nz, nr), -3.5, np.float32))
    assert err == 1 and biteq(got, want)


@pytest.mark.parametrize("nr,nz,crit", [
    (64, 48, (1e-8, -1.0, 300, 0.9)), (37, 29, (5e-3, 5e-3, 5000, 1.0)), (37, 29, (-1.0, 1e-3, 3000, 0.8)),
    (64, 48, (1e-8, -1.0, 150, 1.0)), (64, 48, (1e-8, -1.0, 99, 1.0)), (200, 120, (1e-8, -1.0, 400, 1.0)),
    (300, 40, (1e-8, -1.0, 201, 0.7)),
])
def test_solve_elliptic_drop_in(nr, nz, crit):
    """xee_solve_elliptic vs the oracle incl. the in/out scalars (iterations, residues, err bits)."""
    r1, r2, mi, al = crit
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=nr * 3 + nz)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
    f = (-B).astype(np.float32)
    want = H.solve(bc, coe, f, mi, r1, r2, al)
    got = xee.solve_elliptic(bc, coe, f, mi, r1, r2, al)
    assert biteq(got[0], want[0])
    assert got[1] == want[1] and got[2] == want[2]
    assert got[3].tobytes() == want[3].tobytes() and got[4].tobytes() == want[4].tobytes()


def test_solve_rejects_both_criteria_off():
    A, B, Cc, bc = H.rich_inputs(16, 12)
    with pytest.raises(xee.XeeError) as ei:
        xee.solve_elliptic(bc, np.zeros((12, 16, 9), np.float32), B, 10, -1.0, 0.0, 1.0)
    assert ei.value.status == 1


def test_eta_uw_and_integrals():
    nr, nz = 96, 72
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=21)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    psi = bc
    assert biteq(xee.cal_eta(psi, prof), H.cal_eta(psi, prof))
    u, w = xee.cal_uw(psi, prof)
    uo, wo = H.cal_uw(psi, prof)
    assert biteq(u, uo) and biteq(w, wo)
    rng = np.random.default_rng(2)
    Q = np.exp(-((np.linspace(0, 1, nr - 1)[None, :] - 0.3) / 0.1) ** 2 - ((np.linspace(0, 1, nz - 1)[:, None] - 0.5) / 0.2) ** 2)
    Q = (Q + 1e-3 * rng.standard_normal(Q.shape)).astype(np.float32)
    eta = H.cal_eta(psi, prof)
    assert xee.integrate_weight_B(Q, prof).tobytes() == H.integrate_weight_B(Q, prof).tobytes()
    assert xee.cal_sum_Qeta(Q, eta, prof).tobytes() == H.cal_sum_Qeta(Q, eta, prof).tobytes()
    wt = np.zeros((nz - 1, nr - 1), np.float32)
    H.oracle().orc_cal_wtheta(wo, Q, wt, nr, nz)
    assert biteq(xee.cal_wtheta(wo, Q), wt)


def test_rhs_assembly():
    nr, nz = 80, 56
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    rng = np.random.default_rng(9)
    Q = rng.standard_normal((nz - 1, nr - 1)).astype(np.float32)
    JJ = np.zeros_like(Q)
    H.oracle().orc_JJ_B(Q, prof["exner"], JJ, nr, nz)
    rhs = np.zeros((nz, nr), np.float32)
    wk = np.zeros((nz - 1, nr), np.float32)
    H.oracle().orc_rhs_thm(JJ, rhs, wk, prof["ra"], nr, nz)
    assert biteq(xee.rhs_thermal(Q, prof["ra"], prof["exner"]), rhs)
    fO = np.zeros((nz, nr), np.float32)
    H.oracle().orc_f_from_B(Q, fO, nr, nz)
    assert biteq(xee.rhs_from_B(Q), fO)


# ---- whole case vs the reference's golden outputs ---------------------------------------------

GOLDEN_CASES = ["p128x64_test1", "s64x48_baroall", "s64x48_seccirc", "s64x48_converge", "s64x48_spherical"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_whole_case_matches_reference_golden(name, golden_hashes):
    g = np.load(H.GOLDEN / f"{name}.npz")
    rec = golden_hashes[name]
    cfg = rec["cfg"]
    out = xee.diagnose_case(g["in_A"], g["in_B"], g["in_C"], g["in_bc"],
                            g["in_forcing"] if "in_forcing" in g.files else None,
                            mode=tuple(cfg.get("mode", ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"))),
                            domain=tuple(cfg.get("domain", (0.0, 1.0, 0.0, 1.0))),
                            r1=cfg["r1"], r2=cfg["r2"], max_iter=cfg["max_iter"], alpha=cfg["alpha"])
    sA, sB, sC = out["solver"]
    assert biteq(sA[0], g["out_solver_a-sA.bin"]) and biteq(sB[0], g["out_solver_b-B.bin"]) and biteq(sC[0], g["out_solver_c-sC.bin"])
    variants = [v for v in ("BAROTROPIC", "BAROCLINIC") if v in out]
    assert len(variants) == len(rec["relax"])
    for v, (iters, r1s, r2s) in zip(variants, rec["relax"]):
        res = out[v]
        assert int(res["iters"][0]) == iters
        assert float(res["r1"][0]) == pytest.approx(float(r1s), rel=2e-7)
        assert float(res["r2"][0]) == pytest.approx(float(r2s), rel=2e-7)
        for key, fn in (("eta", f"eta-[{v}]-A.bin"), ("psi", f"rchi-[{v}]-O.bin"), ("psi", f"rpsi-[{v}]-O.bin"),
                        ("u", f"u-[{v}]-C.bin"), ("w", f"w-[{v}]-A.bin")):
            if "out_" + fn in g.files:
                assert biteq(res[key][0], g["out_" + fn]), (v, fn)


def test_test1_fixture_on_gpu():
    """The shipped fixture: 200x200 -> 17,900 iterations; md5 anchors of BASELINE.md section 4."""
    A, B, Cc, bc = H.test1_inputs(200, 200)
    out = xee.diagnose_case(A, B, Cc, bc, r1=5e-3, r2=5e-3, max_iter=100000, alpha=1.0)["BAROTROPIC"]
    assert int(out["iters"][0]) == 17900 and int(out["err"][0]) == 0
    assert H.md5(out["psi"][0]) == "3495ddd52bcb05c39a5c6685f8f1f039"
    assert H.md5(out["eta"][0]) == "a671b036d5e6f27fa1f9893076ae6e8a"


def test_m_grid_md5_anchors():
    """1024x512 fixed-work cases: 500 and 20,000 iterations."""
    A, B, Cc, bc = H.test1_inputs(1024, 512)
    assert H.md5(B) == "986b0fd66de089d7e32d3afc08bb53d0"
    out = xee.diagnose_case(A, B, Cc, bc, r1=1e-8, r2=-1.0, max_iter=500, alpha=1.0)["BAROTROPIC"]
    assert int(out["iters"][0]) == 500 and int(out["err"][0]) == 1
    assert H.md5(out["psi"][0]) == "96ac915d5789a73fdd61a47447916235"
    assert H.md5(out["eta"][0]) == "826fefae4b9f730278ab0cfbefcf4df9"
    out = xee.diagnose_case(A, B, Cc, bc, r1=1e-8, r2=-1.0, max_iter=20000, alpha=1.0)["BAROTROPIC"]
    assert H.md5(out["psi"][0]) == "f3c5da6719e63dbd39f59b82483ebc32"
    assert H.md5(out["eta"][0]) == "327954effa6b33ee14590ba58a50dfce"


def test_l_grid_md5_anchor():
    """4096x2048, 100 iterations."""
    A, B, Cc, bc = H.test1_inputs(4096, 2048)
    assert H.md5(B) == "5d8a980b60a467c7438ba05cd92e1cc8"
    out = xee.diagnose_case(A, B, Cc, bc, r1=1e-8, r2=-1.0, max_iter=100, alpha=1.0)["BAROTROPIC"]
    assert H.md5(out["psi"][0]) == "1cc37e77edff7db54a4dc62430c0e201"
    assert H.md5(out["eta"][0]) == "5690c1085e9714324340d4622341a62d"


# ---- ensemble -----------------------------------------------------------------------------------

def test_ensemble_members_equal_their_solo_runs():
    """Batched members are bit-identical to the same case run alone and to the oracle; members that
    converge at different iterations stop independently."""
    nr, nz, M = 96, 64, 6
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    cfg = dict(r1=2e-4, r2=5e-3, max_iter=20000, alpha=1.0)
    out = xee.diagnose_case(A, B, Cc, bc, **cfg)["BAROTROPIC"]
    iters = set()
    for m in range(M):
        want = H.diagnose_case(A[m], B[m], Cc[m], bc[m], **cfg)["BAROTROPIC"]
        assert int(out["iters"][m]) == want["iters"], m
        assert biteq(out["psi"][m], want["psi"]) and biteq(out["eta"][m], want["eta"]), m
        assert out["r1"][m].tobytes() == want["r1"].tobytes()
        iters.add(want["iters"])
    assert len(iters) > 1, "members should stop at different iterations in this test"


def test_ee_scalars_ensemble():
    nr, nz, M = 64, 48, 3
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    r = np.linspace(0, 1, nr - 1)[None, :]
    z = np.linspace(0, 1, nz - 1)[:, None]
    Q = np.stack([np.exp(-((r - 0.3) / 0.1) ** 2 - ((z - 0.5) / 0.2) ** 2) * (1 + 0.1 * m) for m in range(M)]).astype(np.float32)
    cfg = dict(r1=1e-8, r2=-1.0, max_iter=300, alpha=1.0)
    out = xee.diagnose_case(A, B, Cc, bc, Q=Q, **cfg)["BAROTROPIC"]
    for m in range(M):
        want = H.diagnose_case(A[m], B[m], Cc[m], bc[m], **cfg)
        prof = want["profiles"]
        sq = H.integrate_weight_B(Q[m], prof)
        sqe = H.cal_sum_Qeta(Q[m], want["BAROTROPIC"]["eta"], prof)
        assert out["sum_Q"][m].tobytes() == sq.tobytes()
        assert out["sum_Qeta"][m].tobytes() == sqe.tobytes()
        assert out["efficiency"][m].tobytes() == np.float32(sqe / sq).tobytes()


# ---- temporally blocked sweep (K3-TB) ----------------------------------------------------------

@pytest.mark.parametrize("T", [2, 3, 5, 8, 13, 16])
@pytest.mark.parametrize("nr,nz,iters", [(200, 120, 437), (64, 48, 250), (1024, 512, 203), (130, 300, 301)])
def test_temporal_blocking_is_bit_identical(T, nr, nz, iters):
    """T iterations per launch (overlapped tiles, register-resident coefficients) must reproduce the
    oracle bit for bit for every T, incl. tiles cut by the domain edge, ragged last tiles and a
    check iteration (multiple of 100) that falls inside a launch."""
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=T + nr)
    cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC"),
               domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=iters, alpha=0.9)
    got = xee.diagnose_case(A, B, Cc, bc, options={"tb_T": T}, **cfg)["BAROCLINIC"]
    assert got["sweep"]["T"] == T
    want = H.diagnose_case(A, B, Cc, bc, **cfg)["BAROCLINIC"]
    assert int(got["iters"][0]) == want["iters"] == iters
    assert biteq(got["psi"][0], want["psi"])
    assert got["r1"][0].tobytes() == want["r1"].tobytes() and got["r2"][0].tobytes() == want["r2"].tobytes()


@pytest.mark.parametrize("tile", [808, 806, 804])
@pytest.mark.parametrize("variant", ["BAROTROPIC", "BAROCLINIC"])
def test_every_tile_shape_is_bit_identical(tile, variant):
    """option "tile" = waves x rows per lane of the blocked sweep: every instantiated shape, both operator bodies, with a
    check (a residual-writing launch) inside the run, against the oracle"""
    nr, nz, iters = 333, 270, 231
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=tile)
    cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", variant),
               domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=iters, alpha=1.0)
    want = H.diagnose_case(A, B, Cc, bc, **cfg)[variant]
    for T in (8, 11):
        got = xee.diagnose_case(A, B, Cc, bc, options={"tile": tile, "tb_T": T}, **cfg)[variant]
        nw, r = divmod(tile, 100)
        assert got["sweep"]["tile"] == f"64x{nw * r}", got["sweep"]
        assert biteq(got["psi"][0], want["psi"]) and biteq(got["eta"][0], want["eta"]), (tile, variant, T)
        assert got["r1"][0].tobytes() == want["r1"].tobytes()


def test_single_step_flag_equals_blocked_path():
    nr, nz = 300, 200
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=77)
    cfg = dict(r1=1e-8, r2=-1.0, max_iter=333, alpha=1.0)
    a = xee.diagnose_case(A, B, Cc, bc, **cfg)["BAROTROPIC"]
    b = xee.diagnose_case(A, B, Cc, bc, flags=1, **cfg)["BAROTROPIC"]      # XEE_SOLVE_SINGLE_STEP
    assert biteq(a["psi"], b["psi"]) and a["r1"].tobytes() == b["r1"].tobytes()
    assert a["timing"]["sweep_launches"] < b["timing"]["sweep_launches"] == 333


# ---- parallel order-exact residual norm (xee_exactsum.hip) --------------------------------------

def _planes():
    rng = np.random.default_rng(1234)
    out = {}
    out["normal_37x29"] = rng.standard_normal((29, 37)).astype(np.float32)
    out["normal_200x200"] = (1e-3 * rng.standard_normal((200, 200))).astype(np.float32)
    out["normal_1024x512"] = (1e-4 * rng.standard_normal((512, 1024))).astype(np.float32)
    out["tall_130x2100"] = rng.standard_normal((2100, 130)).astype(np.float32)          # > 64 chunks per column
    out["zeros"] = np.zeros((64, 96), np.float32)
    one = np.zeros((64, 96), np.float32); one[40, 50] = 3.0
    out["single"] = one
    # squares that are exact small multiples of powers of two: round-to-even ties everywhere
    out["ties"] = (2.0 ** rng.integers(-6, 3, size=(150, 260)) * rng.integers(0, 4, size=(150, 260))).astype(np.float32)
    out["wide_range"] = (10.0 ** rng.uniform(-12, 6, size=(120, 300)) * rng.choice([-1, 1], size=(120, 300))).astype(np.float32)
    out["tiny"] = (1e-23 * rng.standard_normal((90, 70))).astype(np.float32)             # squares are subnormal / underflow
    grow = np.zeros((40, 50), np.float32)
    grow[1:-1, 1:-1] = (1.7 ** np.arange(48 * 38)).reshape(48, 38).T.clip(max=1e18)       # each term dwarfs the sum so far
    out["growing"] = grow
    shr = np.zeros((40, 50), np.float32)
    shr[1:-1, 1:-1] = (0.9 ** np.arange(48 * 38)).reshape(48, 38).T                       # terms vanish below half an ulp
    out["shrinking"] = shr
    big = rng.standard_normal((64, 64)).astype(np.float32); big[20, 20] = 1e19; big[21, 20] = 3e19   # overflow to +inf
    out["overflow"] = big
    nan = rng.standard_normal((64, 64)).astype(np.float32); nan[10, 33] = np.nan
    out["nan"] = nan
    return out


@pytest.mark.parametrize("name", list(_planes().keys()))
def test_exact_residual_norm_adversarial(name):
    res = _planes()[name]
    ny, nx = res.shape
    with np.errstate(all="ignore"):
        want = np.float32(H.oracle().orc_residual_sumsq(res, nx, ny))
        want_rms = np.float32(H.oracle().orc_residual_rms(res, nx, ny))
    got, rms, slow = xee.residual_norm(res)
    assert got.tobytes() == want.tobytes() or (np.isnan(got) and np.isnan(want)), (name, got, want, slow)
    assert rms.tobytes() == want_rms.tobytes() or (np.isnan(rms) and np.isnan(want_rms))
    nchunks = (nx - 2) * -(-(ny - 2) // 32)
    if name.startswith("normal") or name == "tall_130x2100":
        assert slow < 0.02 * nchunks + 40, (slow, nchunks)      # the fast path must actually be taken


def test_exact_norm_equals_serial_walk_in_solver():
    """Same solve with the single-lane reference-order walk and with the parallel algorithm."""
    A, B, Cc, bc = H.rich_inputs(64, 48, seed=13)
    cfg = dict(r1=1e-6, r2=2e-3, max_iter=100000, alpha=1.0, domain=(0.0, 2.0e5, 0.0, 1.5e4))
    a = xee.diagnose_case(A, B, Cc, bc, **cfg)["BAROTROPIC"]
    b = xee.diagnose_case(A, B, Cc, bc, options={"exact_sum_serial": 1}, **cfg)["BAROTROPIC"]
    assert int(a["iters"][0]) == int(b["iters"][0]) and biteq(a["psi"], b["psi"])
    assert a["r1"].tobytes() == b["r1"].tobytes() and a["r2"].tobytes() == b["r2"].tobytes()
    assert int(a["err"][0]) == 0, "this case is meant to converge before max_iter"


# ---- uniform coefficient planes are elided from the blocked kernel ------------------------------

def test_uniform_plane_elision_nonzero_constants():
    """Corner coefficient planes that are uniform (any value, here non-zero) are kept in scalars by the
    blocked kernel instead of being loaded: same bits as the oracle fed the full 9-plane array."""
    nr, nz = 150, 90
    rng = np.random.default_rng(3)
    coe = np.zeros((nz, nr, 9), np.float32)
    coe[..., 4] = -(4.0 + rng.random((nz, nr))).astype(np.float32)
    for k in (1, 3, 5, 7):
        coe[..., k] = (1.0 + 0.1 * rng.random((nz, nr))).astype(np.float32)
    for k, v in ((0, 0.015625), (2, -0.03125), (6, 0.0078125), (8, -0.0)):
        coe[..., k] = np.float32(v)
    f = (1e-2 * rng.standard_normal((nz, nr))).astype(np.float32)
    bc = (1e-3 * rng.standard_normal((nz, nr))).astype(np.float32)
    want = H.solve(bc, coe, f, 237, 1e-8, -1.0, 0.9)
    got = xee.solve_elliptic(bc, coe, f, 237, 1e-8, -1.0, 0.9)
    assert biteq(got[0], want[0]) and got[3].tobytes() == want[3].tobytes()


def test_elision_on_off_same_bits():
    A, B, Cc, bc = H.test1_inputs(200, 200)
    cfg = dict(r1=1e-8, r2=-1.0, max_iter=400, alpha=1.0)
    a = xee.diagnose_case(A, B, Cc, bc, options={"kernel": 1}, **cfg)["BAROTROPIC"]       # nine planes, uniform ones elided
    b = xee.diagnose_case(A, B, Cc, bc, options={"kernel": 1, "elision": 0}, **cfg)["BAROTROPIC"]
    want = H.diagnose_case(A, B, Cc, bc, **cfg)["BAROTROPIC"]
    assert a["sweep"]["name"] == "k_sweep_tb<elided>" and b["sweep"]["name"] == "k_sweep_tb"
    assert biteq(a["psi"][0], want["psi"])
    assert biteq(a["psi"], b["psi"]) and a["r1"].tobytes() == b["r1"].tobytes()


# ---- Tier 1 drop-in: cal_coe -> solve_elliptic takes the fast kernel ------------------------------------------------

@pytest.mark.parametrize("variant", ["BAROTROPIC", "BAROCLINIC"])
def test_tier1_solve_reuses_cal_coe_for_the_fast_kernel(variant):
    """The drop-in user's sequence (src/diagnose/diagnose.f90:7,15): cal_coe, then solve_elliptic with that coe.  When the
    coe handed to the solve is bit for bit what this thread's last cal_coe returned, the solve rebuilds the generating
    planes and runs k_sweep_q; a coe that differs in one bit takes the nine-plane kernel.  Same bits as the oracle both
    ways."""
    nr, nz = 200, 120
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=9)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    b_use = sB if variant == "BAROCLINIC" else np.zeros_like(sB)
    f = (-B).astype(np.float32)
    coe, err = xee.cal_coe(sA, b_use, sC, prof["dr"], prof["dz"], nr, nz)
    want = H.solve(bc, H.cal_coe(sA, b_use, sC, prof["dr"], prof["dz"], nr, nz), f, 437, 1e-8, -1.0, 0.9)
    got = xee.solve_elliptic(bc, coe, f, 437, 1e-8, -1.0, 0.9)
    assert xee.last_sweep_info()["name"] == ("k_sweep_q<nine>" if variant == "BAROCLINIC" else "k_sweep_q<star>")
    assert biteq(got[0], want[0]) and got[3].tobytes() == want[3].tobytes()
    coe2 = coe.copy()
    coe2[60, 100, 4] = np.nextafter(coe2[60, 100, 4], np.float32(0))       # one ulp off in one interior coefficient
    want2 = H.solve(bc, coe2, f, 437, 1e-8, -1.0, 0.9)
    got2 = xee.solve_elliptic(bc, coe2, f, 437, 1e-8, -1.0, 0.9)
    assert xee.last_sweep_info()["kernel"] == "k_sweep_tb"
    assert biteq(got2[0], want2[0]) and not biteq(got2[0], got[0])
