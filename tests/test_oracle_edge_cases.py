"""Pins the oracle on the edge-case families of tests/edge_cases.py against the REFERENCE ITSELF
(oracle/_ref/libref_elliptic.so = xtt-lib-fortran/elliptic_tools.f90 compiled unmodified): cal_coe and
solve_elliptic on zero residuals / signed zeros, Inf and NaN iterates, 2^-90- and 2^60-scaled coefficients and
negative alpha must agree bit for bit (NaN payloads canonicalised), in/out scalars included.  The GPU tests
then compare every sweep body with the oracle on the same inputs (tests/test_gpu_quotient.py)."""
import numpy as np
import pytest

import edge_cases as E
import orchelp as H

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (reference sources absent)")


@pytest.mark.parametrize("kind", E.KINDS)
def test_oracle_equals_reference_on_edge_cases(kind):
    c = E.build(kind)
    A, B, Cc, bc, cfg = c["A"], c["B"], c["Cc"], c["bc"], c["cfg"]
    nz, nr = A.shape
    prof = H.profiles(nr, nz, 0, 0, cfg["domain"])
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    b_use = sB if c["variant"] == "BAROCLINIC" else np.zeros_like(sB)
    with np.errstate(all="ignore"):
        coe_o = H.cal_coe(sA, b_use, sC, prof["dr"], prof["dz"], nr, nz)
        coe_r = H.ref_cal_coe(sA, b_use, sC, prof["dr"], prof["dz"], nr, nz)
        assert np.array_equal(coe_o[1:-1, 1:-1].view(np.uint32), coe_r[1:-1, 1:-1].view(np.uint32))
        f = c["forcing"] if c["forcing"] is not None else (-B).astype(np.float32)
        o = H.solve(bc, coe_o, f, cfg["max_iter"], cfg["r1"], -1.0, cfg["alpha"])
        r = H.ref_solve(bc, coe_o, f, cfg["max_iter"], cfg["r1"], -1.0, cfg["alpha"])
    assert o[1] == r[1] and o[2] == r[2], (kind, o[1:3], r[1:3])
    assert E.canon(o[0]).tobytes() == E.canon(r[0]).tobytes(), kind
    for a, b in ((o[3], r[3]), (o[4], r[4])):
        assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (kind, a, b)
