"""The EE integrals are sequential fp32 sums of terms of either sign (old-diagnose/diagnose.f90:1029-1092).  The
GPU reproduces them bit for bit with a parallel algorithm (csrc/xee_exactsum.hip, signed part); these cases are built
to break it: cancellation, sign changes of the running sum, ties, huge dynamic range, power-of-two sums, NaN/Inf."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
ROOT = Path(__file__).resolve().parent.parent


def unit_profiles(nr, nz):
    """metric factors of exactly 1: the summed term is the weight itself"""
    return {"ra": np.arange(nr, dtype=np.float32), "za": np.arange(nz, dtype=np.float32),
            "rcuva": np.ones(nr, np.float32), "rho": np.ones(nz, np.float32)}


def cases(nr, nz):
    rng = np.random.default_rng(nr * 1000 + nz)
    shape = (nz - 1, nr - 1)
    n = shape[0] * shape[1]
    yield "normal", rng.standard_normal(shape)
    yield "cancel", np.where(np.arange(n).reshape(shape) % 2 == 0, 1.0, -1.0) + 1e-4 * rng.standard_normal(shape)
    yield "range", rng.choice([-1.0, 1.0], shape) * 10.0 ** rng.uniform(-20, 20, shape)
    sparse = np.zeros(shape); idx = rng.integers(0, n, 40); sparse.flat[idx] = rng.standard_normal(40) * 1e3
    yield "sparse", sparse
    osc = rng.standard_normal(shape) * 0.01 + 5.0 * np.sin(np.arange(n).reshape(shape) * 0.01)
    yield "oscillating", osc                                   # the running sum changes sign again and again
    yield "negative", -np.abs(rng.standard_normal(shape))
    yield "ties", rng.integers(-7, 8, shape) * 2.0 ** -3 + rng.integers(-1, 2, shape) * 2.0 ** -24 * 1.5
    yield "ones", np.ones(shape)                               # sums that sit exactly on powers of two
    yield "minus_halves", np.full(shape, -0.5)
    grow = np.sign(rng.standard_normal(shape)) * 1.3 ** (np.arange(n).reshape(shape) % 160 - 80)
    yield "growing", grow
    tiny = rng.standard_normal(shape) * 1e-42                  # subnormal terms only
    yield "subnormal", tiny
    bad = rng.standard_normal(shape); bad.flat[n // 2] = np.inf; bad.flat[n // 2 + 7] = -np.inf
    yield "inf", bad
    down = np.concatenate([np.full(n // 2, 1.0), np.full(n - n // 2, -1.0 + 2.0 ** -20)]).reshape(shape)
    yield "up_then_down", down                                 # climbs the binades, then cancels back down through them


@pytest.mark.parametrize("nr,nz", [(65, 49), (130, 71), (300, 200), (1024, 512)])
def test_signed_sequential_sum_bit_exact(nr, nz):
    prof = unit_profiles(nr, nz)
    for name, w in cases(nr, nz):
        w = np.ascontiguousarray(w, np.float32)
        with np.errstate(all="ignore"):
            want = H.integrate_weight_B(w, prof)
        got = xee.integrate_weight_B(w, prof)
        assert got.tobytes() == want.tobytes() or (np.isnan(got) and np.isnan(want)), (name, got, want)


def test_signed_sum_with_metric_factors_and_serial_switch():
    nr, nz = 257, 129
    rng = np.random.default_rng(3)
    prof = H.profiles(nr, nz, 0, 0, (1.0e3, 2.0e5, 0.0, 1.5e4))
    Q = rng.standard_normal((nz - 1, nr - 1)).astype(np.float32)
    eta = rng.standard_normal((nz, nr - 1)).astype(np.float32)
    assert xee.integrate_weight_B(Q, prof).tobytes() == H.integrate_weight_B(Q, prof).tobytes()
    assert xee.cal_sum_Qeta(Q, eta, prof).tobytes() == H.cal_sum_Qeta(Q, eta, prof).tobytes()
    # the single-lane walk (option "exact_sum_serial") gives the same bits
    par = xee.integrate_weight_B(Q, prof)
    xee.set_default_option("exact_sum_serial", 1)          # (conftest resets the defaults after the test)
    assert xee.integrate_weight_B(Q, prof).tobytes() == par.tobytes()
    assert xee.cal_sum_Qeta(Q, eta, prof).tobytes() == H.cal_sum_Qeta(Q, eta, prof).tobytes()


def test_ee_integrals_take_the_fast_path_except_under_cancellation():
    """xee_ensemble_ee_slow_chunks: how many 32-term chunks each EE integral had to add term by term.  sum_Q (terms of one
    sign) stays on the parallel path for every member; so does sum(Q eta) unless the integral CANCELS: for the bench members
    whose B has an even number of half-waves in z (m % 5 in {1, 3}) eta is antisymmetric about the level where the heating
    is symmetric, every column's running sum climbs and falls back through ~20 binades, and a sequential fp32 sum is then
    serial by nature (each binade crossing changes the rounding unit) -- that is the 8.9 ms k_sx_walk outlier of the round-1
    profile.  Values are bit-exact against the oracle either way."""
    import bench
    nr, nz, M = 512, 256, 10
    prof = xee.profiles(nr, nz)
    with xee.Ensemble(nr, nz, M) as e:
        e.set_profiles(prof)
        ins = [bench.member_fields(nr, nz, m) for m in range(M)]
        for m, (A, B, Cc, bc, Q) in enumerate(ins):
            for n, a in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
                e.upload(n, a, first=m)
        e.forcing_from_B(); e.assemble(xee.BAROTROPIC); e.solve(1500, 1e-8, -1.0, 1.0); e.cal_eta()
        rec = e.ee_scalars()
        eta = e.download("ETA")
        sq, sqe = e.ee_slow_chunks()
    nchunks = (nr - 1) * -(-(nz - 1) // 32)
    assert np.all(sq < 0.02 * nchunks), sq
    for m in range(M):
        if m % 5 in (0, 2, 4):
            assert sqe[m] < 0.08 * nchunks, (m, sqe[m], nchunks)
        hp = H.profiles(nr, nz)
        assert rec["sum_Q"][m].tobytes() == H.integrate_weight_B(ins[m][4], hp).tobytes()
        assert rec["sum_Qeta"][m].tobytes() == H.cal_sum_Qeta(ins[m][4], eta[m], hp).tobytes(), m
    assert max(sqe[m] for m in range(M) if m % 5 in (1, 3)) > 0.1 * nchunks      # the cancelling members are the slow ones
