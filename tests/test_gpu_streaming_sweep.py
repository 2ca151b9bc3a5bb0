"""k_sweep_z (csrc/xee_sweepz.hip): the blocked sweep streamed along z that large BAROTROPIC batches take.
Bit for bit against k_sweep_q, the oracle and the reference's own files: its optimistic body, its exact (fallback) body,
every ragged edge of its strips and chunks, the cases the optimistic body must hand over (a -0 in the iterate, a result
that overflows), the host's precondition test (a forcing with zeros stays with k_sweep_q), the launch plan of a check
block, and the batches that select it by themselves."""
import json
from pathlib import Path

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()

GOLD = Path(__file__).parent / "golden"
Z_T = 6


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _run(A, B, Cc, bc, f, n, options, members=1, r1=1e-8, r2=-1.0, check_step=100):
    """the same case `members` times in one ensemble -> (psi of every member, records, timing, sweep_info)"""
    nz, nr = A.shape
    with xee.Ensemble(nr, nz, members) as e:
        for k, v in options.items():
            e.set_option(k, v)
        e.set_profiles(xee.profiles(nr, nz))
        for m in range(members):
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("FORCING", f)):
                e.upload(name, arr, first=m)
        e.assemble(0)
        rec = e.solve(n, r1, r2, 1.0, check_step=check_step)
        return e.download("PSI"), rec, e.timing(), e.sweep_info()


def _forcing(A, seed=5, lo=0.5):
    rng = np.random.default_rng(seed)
    f = (lo + rng.random(A.shape)).astype(np.float32) * rng.choice(np.float32([-1e-9, 1e-9]), size=A.shape)
    return f.astype(np.float32)


# nr x nz, iterations: strips of 52 own columns (one strip; a last strip of 1, 2, 3 columns: the partial fours; exact fits),
# chunks shorter than the run-in, grids smaller than a wave
SHAPES = [(200, 150, 106), (333, 270, 231), (67, 40, 120), (1000, 33, 250), (54, 70, 60), (55, 61, 70), (56, 50, 64),
          (57, 45, 66), (106, 41, 72), (107, 38, 78), (9, 9, 30), (12, 300, 60), (1024, 512, 112)]


@pytest.mark.parametrize("nr,nz,iters", SHAPES)
def test_streamed_sweep_is_bit_identical_to_the_tiled_one(nr, nz, iters):
    A, B, Cc, bc = H.member_inputs(nr, nz, 3)
    f = _forcing(A)
    want, rw, tw, iw = _run(A, B, Cc, bc, f, iters, {"kernel": 3})
    assert tw["z_launches"] == 0 and "k_sweep_q" in iw["name"]
    for pairs in (0, 1, 2, 5):
        for exact in (0, 1):
            got, rg, tg, ig = _run(A, B, Cc, bc, f, iters, {"kernel": 4, "z_pairs": pairs, "q_exact_only": exact})
            assert tg["z_launches"] > 0, (pairs, exact, tg)
            assert biteq(got, want), (nr, nz, pairs, exact, int(np.sum(got.view(np.uint32) != want.view(np.uint32))))
            assert rg["iters"].tolist() == rw["iters"].tolist() and rg["err_now"].tobytes() == rw["err_now"].tobytes()
            if exact:
                assert tg["inline_redo_tiles"] > 0          # every wave took the exact body
            else:
                assert tg["inline_redo_tiles"] == 0 and tg["redo_tiles"] == 0


def test_streamed_sweep_against_the_oracle():
    nr, nz, n = 160, 120, 137
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=21)
    f = _forcing(A, seed=9)
    prof = H.profiles(nr, nz)
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, np.zeros_like(sB), sC, prof["dr"], prof["dz"], nr, nz)
    want = H.solve(bc, coe, f, n, 1e-8, -1.0, 1.0)[0]
    for exact in (0, 1):
        got, _, t, _ = _run(A, B, Cc, bc, f, n, {"kernel": 4, "q_exact_only": exact})
        assert t["z_launches"] > 0 and biteq(got[0], want)


def test_check_block_plan():
    """A block of `check_step` iterations: k_sweep_z's launches of six and, where the length is a multiple of six or four more,
    its own residual-writing launch of six or four; else k_sweep_q's residual-writing launch, which takes the remainder along
    (tb_T .. tb_T + 5 iterations).  A last block without a check: the remainder first, never a single iteration."""
    nr, nz = 120, 90
    A, B, Cc, bc = H.member_inputs(nr, nz, 1)
    f = _forcing(A)
    for check_step, n, tb_T, want_z, want_all in (
            (100, 300, 8, 3 * 17, 3 * 17),        # 16 x 6 + 4, the last by k_sweep_z too
            (100, 250, 8, 2 * 17 + 8, 2 * 17 + 9),   # ... and a last block of 50 without a check: 2 + 8 x 6
            (100, 207, 8, 2 * 17, 2 * 17 + 1),    # ... of 7: one launch (a remainder of one takes a launch of six with it)
            (99, 198, 8, 2 * 15, 2 * 16),         # 15 x 6 + 9: the residual-writing launch is k_sweep_q's
            (50, 100, 8, 2 * 7, 2 * 8),           # 7 x 6 + 8
            (20, 40, 8, 2 * 2, 2 * 3),            # 2 x 6 + 8
            (13, 26, 8, 0, 2 * 2),                # 5 + 8: no room for a launch of six
            (40, 80, 6, 2 * 7, 2 * 7),            # 6 x 6 + 4
            (30, 60, 16, 2 * 5, 2 * 5)):          # 5 x 6
        ref, rr, _, _ = _run(A, B, Cc, bc, f, n, {"kernel": 3, "tb_T": tb_T}, check_step=check_step)
        got, rg, t, _ = _run(A, B, Cc, bc, f, n, {"kernel": 4, "tb_T": tb_T}, check_step=check_step)
        assert (t["z_launches"], t["sweep_launches"]) == (want_z, want_all), (check_step, n, tb_T, t)
        assert t["iterations"] == n and biteq(got, ref) and rg["err_now"].tobytes() == rr["err_now"].tobytes()


def test_negative_zero_iterate_takes_the_exact_body():
    """The optimistic body keeps idle points idle by a quotient of zero: psi + (+-0) is psi unless psi is -0.  A -0 anywhere a
    wave loads -- rim or interior first guess -- sends that wave to the exact body; the others stay optimistic."""
    nr, nz, n = 230, 140, 55
    A, B, Cc, bc = H.member_inputs(nr, nz, 2)
    f = _forcing(A)
    for where in ("rim", "interior"):
        b2 = bc.copy()
        if where == "rim":
            b2[0, 100:104] = np.float32(-0.0); b2[60:64, 0] = np.float32(-0.0); b2[nz - 1, 200] = np.float32(-0.0)
        else:
            b2[70:72, 150:160] = np.float32(-0.0)
        want, rw, _, _ = _run(A, B, Cc, b2, f, n, {"kernel": 3})
        got, rg, t, _ = _run(A, B, Cc, b2, f, n, {"kernel": 4, "z_pairs": 2})
        assert biteq(got, want), where
        waves = t["tiles"]
        assert 0 < t["inline_redo_tiles"] < waves / 2, (where, t)
        # the rim keeps its -0 for good (never written); an interior -0 is gone after the first sweep
        if where == "rim":
            assert np.signbit(got[0][0, 100]) and got[0][0, 100] == 0.0


def test_overflowing_iterate_takes_the_exact_body():
    """A first guess near the top of the range: the update overflows -- the optimistic division then leaves a NaN where the
    IEEE one leaves an infinity.  The wave sees the non-finite result in what it stores and walks again, exactly."""
    nr, nz, n = 150, 100, 30
    A, B, Cc, bc = H.member_inputs(nr, nz, 4)
    f = _forcing(A)
    b2 = bc.copy()
    b2[40:44, 60:70] = np.float32(3.0e38)
    b2[41, 62] = np.float32(-3.0e38)
    with np.errstate(all="ignore"):
        want, rw, _, _ = _run(A, B, Cc, b2, f, n, {"kernel": 3})
        got, rg, t, _ = _run(A, B, Cc, b2, f, n, {"kernel": 4, "z_pairs": 2})
    assert not np.all(np.isfinite(want)), "the case is meant to overflow"
    assert np.any(np.isnan(want))
    assert t["inline_redo_tiles"] > 0
    # infinities and everything finite bit for bit; NaNs where the tiled sweep has NaNs (the SIGN of a NaN that an operation
    # makes is not pinned: it differs between (-0) x inf and -(0 x inf), which the compiler is free to choose between, and
    # between this hardware and the reference's)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    # ... and the same through the exact body alone
    with np.errstate(all="ignore"):
        gx, _, _, _ = _run(A, B, Cc, b2, f, n, {"kernel": 4, "z_pairs": 2, "q_exact_only": 1})
    assert biteq(gx, got)


def test_forcing_with_zeros_stays_with_the_tiled_sweep():
    """min |f| = 0 fails the precondition (a residual could be arbitrarily small): every launch is k_sweep_q's, also under
    "kernel" = 4.  So does a forcing with one tiny value, and a coefficient plane out of range."""
    nr, nz, n = 200, 120, 100
    A, B, Cc, bc = H.member_inputs(nr, nz, 2)
    f = _forcing(A)
    for what in ("zero", "tiny", "fine"):
        f2 = f.copy()
        if what == "zero":
            f2[30:60, 40:90] = 0.0
        elif what == "tiny":
            f2[50, 77] = np.float32(1e-30)
        want, _, _, _ = _run(A, B, Cc, bc, f2, n, {"kernel": 3})
        got, _, t, info = _run(A, B, Cc, bc, f2, n, {"kernel": 4})
        assert biteq(got, want)
        assert (t["z_launches"] > 0) == (what == "fine"), (what, t)
    # the rim of f is never read: zeros there do not matter
    f3 = f.copy(); f3[0, :] = 0.0; f3[:, 0] = 0.0; f3[-1, :] = 0.0; f3[:, -1] = 0.0
    _, _, t, _ = _run(A, B, Cc, bc, f3, n, {"kernel": 4})
    assert t["z_launches"] > 0


def test_other_operators_and_relaxations_do_not_stream():
    nr, nz = 150, 100
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=3)
    cfg = dict(r1=1e-8, r2=-1.0, max_iter=100)
    a = xee.diagnose_case(A, B, Cc, bc, alpha=1.0, options={"kernel": 4}, **cfg)
    bc_mode = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC")
    a2 = xee.diagnose_case(A, B, Cc, bc, alpha=1.0, mode=bc_mode, options={"kernel": 4}, **cfg)
    assert a["BAROTROPIC"]["timing"]["z_launches"] > 0 and a2["BAROCLINIC"]["timing"]["z_launches"] == 0
    b = xee.diagnose_case(A, B, Cc, bc, alpha=0.9, options={"kernel": 4}, **cfg)
    assert b["BAROTROPIC"]["timing"]["z_launches"] == 0
    c = xee.diagnose_case(A, B, Cc, bc, alpha=1.0, **cfg)
    assert c["BAROTROPIC"]["timing"]["z_launches"] == 0       # one small case: the tiled sweep fills the device better
    assert biteq(a["BAROTROPIC"]["psi"], c["BAROTROPIC"]["psi"])


def test_a_large_batch_streams_by_itself_and_matches_the_reference_files():
    """32 cases of 1024x512 (configs[2]'s share of one GPU): the two halves of the member range run side by side, each on
    k_sweep_z for its plain launches.  Members 1, 2, 3, 7 against the md5 of the reference program's own output after 200
    iterations (tests/golden/golden_hashes.json), all of them against the tiled sweep."""
    nr, nz, M, n = 1024, 512, 32, 200
    gold = json.loads((GOLD / "golden_hashes.json").read_text())
    outs = {}
    for k in (0, 3):
        with xee.Ensemble(nr, nz, M) as e:
            e.set_option("kernel", k)
            e.set_profiles(xee.profiles(nr, nz))
            for m in range(M):
                A, B, Cc, bc = H.member_inputs(nr, nz, m)
                for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                    e.upload(name, arr, first=m)
            e.forcing_from_B(); e.assemble(xee.BAROTROPIC)
            rec = e.solve(n, 1e-8, -1.0, 1.0)
            outs[k] = (e.download("PSI"), rec, e.timing())
    t = outs[0][2]
    assert t["sweep_streams"] == 2 and t["z_launches"] == 2 * 2 * 17 and t["sweep_launches"] == 2 * 2 * 17, t
    assert outs[3][2]["z_launches"] == 0
    assert biteq(outs[0][0], outs[3][0]) and outs[0][1]["err_now"].tobytes() == outs[3][1]["err_now"].tobytes()
    assert t["inline_redo_tiles"] == 0 and t["redo_tiles"] == 0
    hit = 0
    for m in (1, 2, 3, 7):
        key = f"member{m}_1024x512_200"
        if key in gold:
            assert H.md5(outs[0][0][m]) == gold[key]["md5"]["rchi-[BAROTROPIC]-O.bin"], m
            hit += 1
    assert hit == 4, "reference anchors missing from golden_hashes.json"


def test_random_shapes_chunkings_and_lengths():
    """Forty random grids (8 .. 300 points a side, so every alignment of the last strip and of the chunks against the two-row
    fetch groups and the eight-tick loop turn occurs), random chunk pairs, check steps and iteration counts; members that differ."""
    rng = np.random.default_rng(20261005)
    for trial in range(40):
        nr, nz = int(rng.integers(8, 301)), int(rng.integers(8, 301))
        M = int(rng.integers(1, 4))
        cs = int(rng.integers(14, 60))
        n = int(rng.integers(cs, 4 * cs))
        pairs = int(rng.integers(0, 9))
        outs = {}
        for k in (3, 4):
            with xee.Ensemble(nr, nz, M) as e:
                e.set_option("kernel", k); e.set_option("z_pairs", pairs)
                e.set_profiles(xee.profiles(nr, nz))
                for m in range(M):
                    A, B, Cc, bc = H.member_inputs(nr, nz, 10 * trial + m)
                    f = _forcing(A, seed=trial + m)
                    for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("FORCING", f)):
                        e.upload(name, arr, first=m)
                e.assemble(0)
                rec = e.solve(n, 1e-8, -1.0, 1.0, check_step=cs)
                outs[k] = (e.download("PSI"), rec, e.timing())
        tz = outs[4][2]
        assert tz["z_launches"] > 0 and tz["inline_redo_tiles"] == 0, (trial, nr, nz, M, cs, n, pairs, tz)
        assert biteq(outs[4][0], outs[3][0]), (trial, nr, nz, M, cs, n, pairs)
        assert outs[4][1]["err_now"].tobytes() == outs[3][1]["err_now"].tobytes()


def test_streamed_launches_inside_block_graphs():
    """A small batch with a long solve ahead replays its check blocks as HIP graphs (solve_pipelined); under "kernel" = 4 the
    graphs hold k_sweep_z launches.  Same bits, same counts, as the tiled sweep."""
    nr, nz, n = 130, 80, 7000
    A, B, Cc, bc = H.member_inputs(nr, nz, 5)
    f = _forcing(A)
    want, rw, tw, _ = _run(A, B, Cc, bc, f, n, {"kernel": 3}, r1=1e-30)
    got, rg, tg, _ = _run(A, B, Cc, bc, f, n, {"kernel": 4}, r1=1e-30)
    assert tg["z_launches"] == 70 * 17 and tg["sweep_launches"] == 70 * 17 and tg["iterations"] == n
    assert biteq(got, want) and rg["err_now"].tobytes() == rw["err_now"].tobytes() and rg["iters"].tolist() == rw["iters"].tolist()


@pytest.mark.parametrize("members", [1, 3, 4])
def test_streamed_residual_writing_launch(members):
    """A check block whose length is 16 x 6 + 4 (the default 100) or a multiple of six ends in k_sweep_z's own residual-writing
    launch (4 or 6 iterations): the residual plane, bit for bit, and the iterate against k_sweep_q's; every launch streamed.  With
    several members (sharing their operator: member pairs) one of which has converged and stopped, nothing of the stopped one is
    written by the later blocks."""
    nr, nz = 140, 95
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=31)
    outs = {}
    for cs, n in ((100, 300), (60, 180), (22, 66), (50, 150)):
        for k in (3, 4):
            with xee.Ensemble(nr, nz, members) as e:
                e.set_option("kernel", k)
                e.set_profiles(xee.profiles(nr, nz))
                for m in range(members):
                    f = _forcing(A, seed=m) * np.float32(1e3 if m == 1 else 1.0)
                    for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc * np.float32(1 + m)), ("FORCING", f)):
                        e.upload(name, arr, first=m)
                e.assemble(0)
                # (member 0's residual falls below r1 at the first check of the longer blocks: converge_time = 1 stops it there)
                rec = e.solve(n, 3e-7, -1.0, 1.0, check_step=cs, converge_time=1)
                outs[k] = (e.download("PSI"), e.download("RESIDUAL"), rec, e.timing())
        (pz, rz, recz, tz), (pq, rq, recq, tq) = outs[4], outs[3]
        assert recz["iters"].tolist() == recq["iters"].tolist() and recz["err_now"].tobytes() == recq["err_now"].tobytes(), (cs, recz["iters"], recq["iters"])
        assert biteq(pz, pq), cs
        assert biteq(rz[:, 1:-1, 1:-1], rq[:, 1:-1, 1:-1]), cs
        blocks = int(max(recz["iters"])) // cs
        if cs in (100, 60, 22):          # 16 x 6 + 4, 10 x 6, 3 x 6 + 4: all of them k_sweep_z's
            assert tz["z_launches"] == tz["sweep_launches"] > 0, (cs, tz)
        else:                            # 50 = 7 x 6 + 8: the residual-writing launch is k_sweep_q's
            assert 0 < tz["z_launches"] == tz["sweep_launches"] - blocks, (cs, tz)


@pytest.mark.parametrize("variant", [0, 1])
def test_members_that_cannot_stream_leave_the_others_streaming(variant):
    """24 cases of 1024x512, the last 8 with a zero right-hand side (the rim-boundary kind of a source-term sweep, whose cases come last
    in the batch: old-diagnose/diagnose.f90:548-568): the first 16 stream, the 8 take the tiled sweep, side by side -- the same bits
    as the tiled sweep for all; both operators."""
    nr, nz, M, n = 1024, 512, 24, 300
    ins = [H.member_inputs(nr, nz, m) for m in range(M)]
    outs = {}
    for tag, opts in (("auto", {}), ("tiled", {"kernel": 3}), ("one", {"split_halves": 0})):
        with xee.Ensemble(nr, nz, M) as e:
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_profiles(xee.profiles(nr, nz))
            for m, (A, B, Cc, bc) in enumerate(ins):
                f = (-B).astype(np.float32) if m < 16 else np.zeros_like(B)
                for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("FORCING", f)):
                    e.upload(name, arr, first=m)
            e.assemble(variant)
            rec = e.solve(n, 1e-8, -1.0, 1.0)
            outs[tag] = (e.download("PSI"), rec.copy(), e.timing())
    pa, ra, ta = outs["auto"]
    assert ta["sweep_streams"] == 2 and 0 < ta["z_launches"] < ta["sweep_launches"], ta
    assert ta["z_launches"] == 3 * (17 if variant == 0 else 25)
    for tag in ("tiled", "one"):
        pb, rb, tb = outs[tag]
        assert tb["z_launches"] == 0 and tb["sweep_streams"] == 1
        assert biteq(pa, pb) and ra["err_now"].tobytes() == rb["err_now"].tobytes() and ra["iters"].tolist() == rb["iters"].tolist(), tag
