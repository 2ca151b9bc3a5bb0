"""The remaining finite-difference / integral stencils of the path (quick-tools2.f90, the momentum RHS and
the exchange-conversion line integral of the legacy program) on the GPU, bit for bit against the oracle."""
import ctypes as C

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
from xlab_ee_fortran_amd import xee as X  # noqa: E402


def biteq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("nr,nz", [(64, 48), (37, 29), (130, 21), (5, 6)])
def test_fd_stencils(nr, nz):
    rng = np.random.default_rng(nr + nz)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    O = rng.standard_normal((nz, nr)).astype(np.float32)
    Bc = rng.standard_normal((nz - 1, nr - 1)).astype(np.float32)
    L = H.oracle()
    cases = [("d_dz_B2A", Bc, (nz, nr - 1), lambda f, t: L.orc_d_dz_B2A(f, t, prof["za"], nr, nz)),
             ("d_dz_O2C", O, (nz - 1, nr), lambda f, t: L.orc_d_dz_O2C(f, t, prof["za"], nr, nz)),
             ("d_dr_B2B", Bc, (nz - 1, nr - 1), lambda f, t: L.orc_d_dr_B2B(f, t, prof["ra"], nr, nz)),
             ("d_dr_B2C", Bc, (nz - 1, nr), lambda f, t: L.orc_d_dr_B2C(f, t, prof["ra"], nr, nz)),
             ("d_dr_O2A", O, (nz, nr - 1), lambda f, t: L.orc_d_dr_O2A(f, t, prof["ra"], nr, nz)),
             ("d_rcuvdr_O2A", O, (nz, nr - 1), lambda f, t: L.orc_d_rcuvdr_O2A(f, t, prof["ra"], prof["rcuva"], nr, nz))]
    for name, src, shape, ref in cases:
        want = np.full(shape, 7.5, np.float32)          # sentinel: untouched entries must stay untouched
        with np.errstate(all="ignore"):
            ref(src, want)
            got = X.fd_stencil(name, src, prof, to_dat=np.full(shape, 7.5, np.float32))
        assert biteq(got, want), name


def test_rhs_momentum_and_exchange_conversion():
    nr, nz = 72, 50
    rng = np.random.default_rng(5)
    prof = H.profiles(nr, nz, 0, 0, (1.0e3, 2.0e5, 0.0, 1.5e4))
    m2 = (1.0 + rng.random((nz - 1, nr - 1))).astype(np.float32)
    F = rng.standard_normal((nz - 1, nr - 1)).astype(np.float32)
    want = np.zeros((nz, nr), np.float32)
    H.oracle().orc_rhs_mom(m2, F, want, np.zeros((nz - 1, nr - 1), np.float32), np.zeros((nz, nr - 1), np.float32),
                           prof["za"], prof["rcuva"], nr, nz)
    assert biteq(X.rhs_momentum(m2, F, prof), want)

    rpsi = rng.standard_normal((nz, nr)).astype(np.float32)
    rchi = rng.standard_normal((nz, nr)).astype(np.float32)
    rhoC = (1.0 + rng.random((nz, nr))).astype(np.float32)
    bnd_w = np.zeros((2, nr - 1), np.float32)
    H.oracle().orc_exchange_conversion.restype = C.c_float
    s_w = np.float32(H.oracle().orc_exchange_conversion(rpsi, rchi, rhoC, bnd_w, prof["ra"], prof["za"], prof["rho"], nr, nz))
    bnd, s = X.exchange_conversion(rpsi, rchi, rhoC, prof)
    assert biteq(bnd, bnd_w) and s.tobytes() == s_w.tobytes()


@pytest.mark.parametrize("nr,nz", [(72, 50), (37, 29), (130, 21)])
def test_legacy_stage_two_thermal_forecast(nr, nz):
    """Stage II of the legacy TENDENCY mode (old-diagnose/diagnose.f90:462-518): dtheta/dt, the anomaly update chain and
    relativeTheta on the GPU, bit for bit against the oracle's restatement (parity unpinned against the legacy binary)."""
    rng = np.random.default_rng(nr * 7 + nz)
    prof = H.profiles(nr, nz, 0, 0, (1.0e3, 2.0e5, 0.0, 1.5e4))
    f32 = lambda *shape: rng.standard_normal(shape).astype(np.float32)
    JJ, rhoA, wA, rhoBC, uC = f32(nz - 1, nr - 1), f32(nz, nr - 1) * 1e-4, f32(nz, nr - 1), f32(nz - 1, nr) * 1e-6, f32(nz - 1, nr)
    L = H.oracle()
    want = np.zeros((nz - 1, nr - 1), np.float32)
    L.orc_theta_tendency(JJ, rhoA, wA, rhoBC, uC, want, nr, nz)
    got = X.theta_tendency(JJ, rhoA, wA, rhoBC, uC)
    assert biteq(got, want)

    dzA, drC = f32(nz, nr - 1) * 1e-3, f32(nz - 1, nr) * 1e-5
    want = np.zeros((nz - 1, nr - 1), np.float32)
    L.orc_relative_theta(want, dzA, drC, prof["ra"], prof["za"], nr, nz)
    assert biteq(X.relative_theta(dzA, drC, prof), want)

    rhoBB = f32(nz - 1, nr - 1) * 1e-6
    th, bb, ra_, rc = got.copy(), rhoBB.copy(), rhoA.copy(), rhoBC.copy()
    ban, sb = np.zeros_like(th), np.zeros_like(th)
    L.orc_theta_anomaly_update(th, 3600.0, bb, ra_, rc, ban, sb, prof["ra"], prof["za"], prof["rcuva"], prof["rho"], nr, nz)
    out = X.theta_anomaly_update(got, 3600.0, rhoBB, rhoA, rhoBC, prof)
    for name, ref in (("theta", th), ("rhoB_B", bb), ("rhoA_A", ra_), ("rhoB_C", rc), ("b_anomaly", ban), ("solver_b_anomaly", sb)):
        assert biteq(out[name], ref), name


@pytest.mark.parametrize("spherical", [False, True])
def test_m2_from_C_intended_form(spherical):
    """old-diagnose/diagnose.f90:359-380 restated as intended (the legacy lines read out of bounds): a sequential sum along r"""
    nr, nz = 96, 40
    rng = np.random.default_rng(9)
    prof = H.profiles(nr, nz, 1 if spherical else 0, 0, (10.0, 60.0, 0.0, 1.5e4) if spherical else (1.0e3, 2.0e5, 0.0, 1.5e4))
    rhoC = (1.0 + rng.random((nz - 1, nr))).astype(np.float32) * 1e-9
    sin_t = prof["sin_table"] if spherical else None
    want = np.zeros((nz - 1, nr - 1), np.float32)
    H.oracle().orc_m2_from_C(rhoC, want, prof["ra"], prof["rcuva"], None if sin_t is None else sin_t.ctypes.data, nr, nz)
    assert biteq(X.m2_from_C(rhoC, prof, sin_t), want)
