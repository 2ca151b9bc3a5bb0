"""The blocked sweep fed by the operator's four generating planes (xee_sweep4.hip) against the same
sweep fed by the nine stored coefficient planes and against the oracle: bit for bit, both operator
variants, ragged grids, every T."""
import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("variant", ["BAROTROPIC", "BAROCLINIC"])
@pytest.mark.parametrize("nr,nz,iters,T", [(200, 120, 311, 6), (64, 48, 250, 8), (1024, 512, 203, 6), (130, 300, 301, 3), (97, 67, 199, 13)])
def test_generating_planes_path_is_bit_identical(variant, nr, nz, iters, T):
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=T + nr)
    cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", variant),
               domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=iters, alpha=0.9)
    got4 = xee.diagnose_case(A, B, Cc, bc, options={"tb_T": T}, **cfg)[variant]
    got9 = xee.diagnose_case(A, B, Cc, bc, options={"tb_T": T, "kernel": 1}, **cfg)[variant]
    gotx = xee.diagnose_case(A, B, Cc, bc, options={"tb_T": T, "kernel": 2}, **cfg)[variant]
    assert got4["sweep"]["kernel"] == "k_sweep_q" and got9["sweep"]["kernel"] == "k_sweep_tb" and gotx["sweep"]["kernel"] == "k_sweep_tb4"
    assert biteq(gotx["psi"][0], want_psi := H.diagnose_case(A, B, Cc, bc, **cfg)[variant]["psi"])
    want = H.diagnose_case(A, B, Cc, bc, **cfg)[variant]
    assert biteq(got4["psi"][0], want["psi"]) and biteq(got9["psi"][0], want["psi"])
    assert got4["r1"][0].tobytes() == want["r1"].tobytes() == got9["r1"][0].tobytes()
    assert biteq(got4["eta"][0], want["eta"])


def test_signed_zero_coefficients_survive():
    """Cp = -0 style corner cases: planes with negative zeros and zeros in A/C go through the rebuilt
    weights exactly as through the stored ones."""
    nr, nz = 80, 60
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=2)
    A[10:20, 5:30] = 0.0
    Cc[30:40, 40:70] = -0.0
    cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BARO_ALL"),
               domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=37, alpha=1.0)
    with np.errstate(all="ignore"):
        got = xee.diagnose_case(A, B, Cc, bc, **cfg)
        want = H.diagnose_case(A, B, Cc, bc, **cfg)
    for v in ("BAROTROPIC", "BAROCLINIC"):
        a, b = got[v]["psi"][0], want[v]["psi"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) or (np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)))
