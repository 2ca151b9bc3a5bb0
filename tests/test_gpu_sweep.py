"""GPU test of the source-term sweep (BASELINE.json configs[4], SURVEY.md 8d "Config 5"): every
case of a small sweep -- 2 operators x 3 right-hand sides per timestep -- against the oracle, bit for
bit, fields and EE scalars."""
import ctypes as C

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
from xlab_ee_fortran_amd import sweep as S  # noqa: E402
from xlab_ee_fortran_amd.xee_tools import ensemble_member  # noqa: E402


def biteq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def oracle_case(nr, nz, t, op, rhs, max_iter, alpha):
    A, B, Cc, _ = ensemble_member(nr, nz, t)
    prof = H.profiles(nr, nz)
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB if op == 1 else np.zeros_like(sB), sC, prof["dr"], prof["dz"], nr, nz)
    Q = S.heating_Q(nr, nz)
    if rhs == 0:
        f = (-B).astype(np.float32)
    elif rhs == 1:
        JJ = np.zeros_like(Q)
        H.oracle().orc_JJ_B(Q, prof["exner"], JJ, nr, nz)
        f = np.zeros((nz, nr), np.float32)
        H.oracle().orc_rhs_thm(JJ, f, np.zeros((nz - 1, nr), np.float32), prof["ra"], nr, nz)
    else:
        f = np.zeros((nz, nr), np.float32)
    bc = S.rim_bc(nr, nz) if rhs == 2 else np.zeros((nz, nr), np.float32)
    psi, it, eb, r1o, r2o, _ = H.solve(bc, coe, f, max_iter, 1e-8, -1.0, alpha)
    eta = H.cal_eta(psi, prof)
    sq = H.integrate_weight_B(Q, prof)
    sqe = H.cal_sum_Qeta(Q, eta, prof)
    return psi, eta, it, r1o, sq, sqe


@pytest.mark.parametrize("fused", [True, False])
def test_sweep_cases_match_oracle(fused):
    nr, nz, nt, mi = 96, 64, 3, 300
    rec, fields = S.run(nr, nz, nt, max_iter=mi, keep_fields=True, fused=fused)
    assert len(rec) == nt * S.CASES_PER_TIMESTEP
    assert list(rec["case_id"]) == list(range(nt * S.CASES_PER_TIMESTEP))
    for t in range(nt):
        for op in (0, 1):
            for rhs in (0, 1, 2):
                psi, eta, it, r1o, sq, sqe = oracle_case(nr, nz, t, op, rhs, mi, 1.0)
                got = fields[(t, op, rhs)]
                r = rec[S.CASES_PER_TIMESTEP * t + 3 * op + rhs]
                assert biteq(got["psi"], psi), (t, op, rhs)
                assert biteq(got["eta"], eta), (t, op, rhs)
                assert int(r["iters"]) == it
                assert r["err_now"].tobytes() == r1o.tobytes()
                assert r["sum_Q"].tobytes() == sq.tobytes()
                assert r["sum_Qeta"].tobytes() == sqe.tobytes()
                with np.errstate(all="ignore"):
                    assert r["efficiency"].tobytes() == np.float32(sqe / sq).tobytes()
    # the three right-hand sides really are different problems
    assert not biteq(fields[(0, 0, 0)]["psi"], fields[(0, 0, 1)]["psi"])
    assert not biteq(fields[(0, 0, 0)]["psi"], fields[(0, 0, 2)]["psi"])
    assert np.any(fields[(0, 1, 2)]["psi"][:, -1] != 0)


def test_sweep_fullsize_kinds_match_the_reference_program():
    """configs[4] at its real size: the six (operator, right-hand side) kinds of sweep timestep 1 -- and the zero-RHS /
    rim-boundary kind of timestep 0 -- at 1024x512, 600 iterations.  psi against the md5 of what the REFERENCE PROGRAM wrote
    for the same inputs (tests/golden/make_golden_sweep.py: SECONDARY_CIRCULATION mode with the right-hand side as its
    forcing file); eta and the EE scalars against the oracle's stencils applied to that psi; the residue the program printed."""
    import hashlib
    import json
    from pathlib import Path
    gh = json.loads((Path(__file__).parent / "golden" / "golden_hashes.json").read_text())
    nr, nz, mi = 1024, 512, 600
    prof = H.profiles(nr, nz)
    Q = S.heating_Q(nr, nz)
    sq = H.integrate_weight_B(Q, prof)
    sw = S.Sweep(nr, nz, [0, 1])
    got = {}

    def keep(op, rhs, ens, first):
        got[(op, rhs)] = (ens.download("PSI", first, 2), ens.download("ETA", first, 2))
    try:
        rec = sw.step(mi, on_stage=keep)
        frac = {k: (t["redo_tiles"] + t["inline_redo_tiles"]) / t["tiles"] for k, t in sw.stage_timing.items()}
    finally:
        sw.close()
    for t, kinds in ((1, (0, 1, 2)), (0, (2,))):
        for rhs in kinds:
            ref = gh[f"sweep_t{t}_{S.RHS_NAMES[rhs]}_1024x512_600"]
            for op, opn in enumerate(S.OP_NAMES):
                psi, eta = (x[t] for x in got[(op, rhs)])
                assert hashlib.md5(psi.tobytes()).hexdigest() == ref["md5"][f"rpsi-[{opn}]-O.bin"], (t, rhs, opn)
                assert biteq(eta, H.cal_eta(psi, prof)), (t, rhs, opn)
                r = rec[S.CASES_PER_TIMESTEP * t + 3 * op + rhs]
                assert int(r["iters"]) == ref["relax"][op][0] == mi
                assert abs(float(r["err_now"]) - float(ref["relax"][op][1])) <= 2e-7 * abs(float(ref["relax"][op][1]))   # 7 printed digits
                assert r["sum_Q"].tobytes() == sq.tobytes()
                assert r["sum_Qeta"].tobytes() == H.cal_sum_Qeta(Q, eta, prof).tobytes()
    # most of a rim-driven plane still holds the all-zero first guess after 600 iterations: those tiles take the cheap exit,
    # not the exact body (round 2: every one of them ran twice)
    assert all(f < 0.12 for f in frac.values()), frac
