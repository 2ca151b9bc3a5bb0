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


def test_sweep_cases_match_oracle():
    nr, nz, nt, mi = 96, 64, 3, 300
    rec, fields = S.run(nr, nz, nt, max_iter=mi, keep_fields=True)
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
