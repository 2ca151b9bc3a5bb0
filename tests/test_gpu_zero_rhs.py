"""Zero right-hand sides (the legacy sweep's boundary-driven solves, src/old-diagnose/diagnose.f90:548-568: f = 0,
rchi = rchi_bc; SURVEY.md 8d "zero-RHS with bc_init = 1e-5 sin(pi z) on the r = 1 rim").  Most of such a plane holds the
all-zero first guess for thousands of iterations: every residual there is EXACTLY zero.  The blocked sweep's optimistic
body takes those tiles through a cheap exit and admits zero residuals elsewhere (csrc/xee_sweepq.hip "Zero residuals");
here: iterates, residue, iteration counts AND the stored residual plane (signs of its zeros included) against the oracle
and against the reference program's own files, both drivers, both operator bodies."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
from xlab_ee_fortran_amd import sweep as S  # noqa: E402
from xlab_ee_fortran_amd.xee_tools import ensemble_member  # noqa: E402

GOLD = Path(__file__).parent / "golden"


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _oracle(A, B, Cc, bc, f, op, n, alpha=1.0):
    """-> psi after n iterations, and the residual plane of iteration n (do_elliptic(psi_{n-1}) - f, :189-190)"""
    nz, nr = A.shape
    prof = H.profiles(nr, nz)
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB if op == 1 else np.zeros_like(sB), sC, prof["dr"], prof["dz"], nr, nz)
    prev = H.solve(bc, coe, f, n - 1, 1e-8, -1.0, alpha)[0]
    out = np.zeros((nz, nr), np.float32)
    err = C.c_int(0)
    H.oracle().orc_do_elliptic(prev, coe, out, nr, nz, C.byref(err))
    out[1:-1, 1:-1] = out[1:-1, 1:-1] - f[1:-1, 1:-1]
    full = H.solve(bc, coe, f, n, 1e-8, -1.0, alpha)
    return full, out


def _gpu(A, B, Cc, bc, f, op, n, sync, alpha=1.0, options=None):
    nz, nr = A.shape
    with xee.Ensemble(nr, nz, 1) as e:
        e.set_option("sync_checks", sync)
        for k, v in (options or {}).items():
            e.set_option(k, v)
        e.set_profiles(xee.profiles(nr, nz))
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("FORCING", f)):
            e.upload(name, arr)
        e.assemble(op)
        rec = e.solve(n, 1e-8, -1.0, alpha)
        return e.download("PSI")[0], e.download("RESIDUAL")[0], rec[0], e.timing()


@pytest.mark.parametrize("sync", [0, 1])
@pytest.mark.parametrize("op", [0, 1])
@pytest.mark.parametrize("n", [100, 300])
def test_rim_driven_zero_rhs_matches_oracle(sync, op, n):
    nr, nz = 400, 150                        # several tile columns stay all-zero at 100 and at 300 iterations
    A, B, Cc, _ = ensemble_member(nr, nz, 3)
    bc, f = S.rim_bc(nr, nz), np.zeros((nz, nr), np.float32)
    (psi, it, eb, r1, r2, _), res = _oracle(A, B, Cc, bc, f, op, n)
    gpsi, gres, rec, tm = _gpu(A, B, Cc, bc, f, op, n, sync)
    assert biteq(gpsi, psi)
    assert biteq(gres[1:-1, 1:-1], res[1:-1, 1:-1])          # bit patterns: the sign of every zero residual too
    assert int(rec["iters"]) == it and int(rec["err_bits"]) == eb and rec["err_now"].tobytes() == r1.tobytes()
    assert np.count_nonzero(psi[:, : nr - n - 2]) == 0       # the untouched part really is there
    # only the tiles at the front (values below the optimistic body's range) may take the exact body
    assert (tm["redo_tiles"] + tm["inline_redo_tiles"]) <= 0.5 * tm["tiles"], tm


@pytest.mark.parametrize("op", [0, 1])
@pytest.mark.parametrize("fsign", [0.0, -0.0])
def test_all_zero_problem(op, fsign):
    """psi = 0, f = +-0 everywhere: the fixed point from the first iteration on; no tile may need the exact body, and the
    stored residual is the reference's +0 (elliptic_tools.f90:77-85,190: the nine-term sum of signed zeros, minus +-0)."""
    nr, nz = 200, 150
    A, B, Cc, _ = ensemble_member(nr, nz, 2)
    bc = np.zeros((nz, nr), np.float32)
    f = np.full((nz, nr), fsign, np.float32)
    (psi, it, eb, r1, r2, _), res = _oracle(A, B, Cc, bc, f, op, 200)
    for opts in ({}, {"q_exact_only": 1}):
        gpsi, gres, rec, tm = _gpu(A, B, Cc, bc, f, op, 200, 0, options=opts)
        assert biteq(gpsi, psi) and np.count_nonzero(gpsi.view(np.uint32)) == 0
        assert biteq(gres[1:-1, 1:-1], res[1:-1, 1:-1]), opts
        assert rec["err_now"].tobytes() == r1.tobytes()
        if not opts:
            assert tm["redo_tiles"] + tm["inline_redo_tiles"] == 0, tm


def test_partly_zero_with_negative_zero_and_patch():
    """zeros of both signs in the first guess, a small patch of forcing, negative damping: tiles that are all-zero, tiles
    that hold -0 (no zero may pass there) and live tiles side by side"""
    nr, nz = 300, 200
    A, B, Cc, _ = ensemble_member(nr, nz, 1)
    for op in (0, 1):
        for alpha in (1.0, -0.05):
            bc = np.zeros((nz, nr), np.float32)
            bc[40:80, 150:200] = -0.0
            bc[:, -1] = 1e-5
            f = np.zeros((nz, nr), np.float32)
            f[100:110, 20:60] = 1e-9
            f[150:160, 200:240] = -0.0
            (psi, it, eb, r1, r2, _), res = _oracle(A, B, Cc, bc, f, op, 200, alpha)
            gpsi, gres, rec, tm = _gpu(A, B, Cc, bc, f, op, 200, 0, alpha)
            assert biteq(gpsi, psi), (op, alpha)
            assert biteq(gres[1:-1, 1:-1], res[1:-1, 1:-1]), (op, alpha)
            assert rec["err_now"].tobytes() == r1.tobytes()


def test_rim_case_against_the_reference_programs_files():
    """tests/golden/s256x96_rimbc.npz: the reference program run in SECONDARY_CIRCULATION mode with an all-zero forcing file
    and bc_init = the rim field, BARO_ALL, 100 iterations -- its rpsi, u, w files bit for bit"""
    z = np.load(GOLD / "s256x96_rimbc.npz")
    A, B, Cc, bc, f = (z[k] for k in ("in_A", "in_B", "in_C", "in_bc", "in_forcing"))
    assert np.count_nonzero(f) == 0 and np.count_nonzero(bc[:, :-1]) == 0 and np.count_nonzero(bc[:, -1]) > 0
    out = xee.diagnose_case(A, B, Cc, bc, forcing=f, mode=("SECONDARY_CIRCULATION", "CYLINDRICAL", "DENSITY_NORMAL", "BARO_ALL"),
                            r1=1e-8, r2=-1.0, max_iter=100, alpha=1.0)
    for opn in ("BAROTROPIC", "BAROCLINIC"):
        assert biteq(out[opn]["psi"][0], z[f"out_rpsi-[{opn}]-O.bin"]), opn
        assert biteq(out[opn]["u"][0], z[f"out_u-[{opn}]-C.bin"]), opn
        assert biteq(out[opn]["w"][0], z[f"out_w-[{opn}]-A.bin"]), opn
        assert out[opn]["sweep"]["kernel"] == "k_sweep_q"
