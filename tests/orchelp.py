"""Test-side helpers: ctypes bindings for the oracle (oracle/libxee_oracle.so), for the reference
built into oracle/_ref/ (when present), synthetic-input generators and the
reference's wire formats.

TEST INFRASTRUCTURE ONLY -- nothing under xlab-ee-fortran_amd/ imports this.

Array convention: numpy arrays are (n2, n1) C-order float32 == the reference's Fortran (n1, n2)
column-major arrays == the layout of its .bin files (xtt-lib-fortran/field_tools.f90:30-76).
"""
import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "libxee_oracle.so"
REF_DIR = ROOT / "oracle" / "_ref"
GOLDEN = ROOT / "tests" / "golden"

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32 = C.c_int
f32 = C.c_float
pi32 = C.POINTER(C.c_int)
pf32 = C.POINTER(C.c_float)


class Constants(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("g0", "theta0", "Rd", "Cv", "Cp", "kappa", "h0", "p0")]


_oracle = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "oracle"], check=True)


def oracle():
    """ctypes handle of the C restatement, argtypes set."""
    global _oracle
    if _oracle is not None:
        return _oracle
    if not ORACLE_SO.exists():
        build_oracle()
    L = C.CDLL(str(ORACLE_SO))
    L.orc_constants.argtypes = [C.POINTER(Constants)]
    L.orc_profiles.argtypes = [i32, i32, f32p, i32, i32, f32p, f32p, f32p, f32p, f32p, f32p, pf32, pf32]
    L.orc_solver_coefficients.argtypes = [f32p] * 8 + [i32, i32]
    L.orc_cal_coe.argtypes = [f32p, f32p, f32p, f32p, f32, f32, i32, i32, pi32]
    L.orc_do_elliptic.argtypes = [f32p, f32p, f32p, i32, i32, pi32]
    for fn in (L.orc_solve_elliptic, L.orc_solve_elliptic_reforder):
        fn.argtypes = [pi32, i32, i32, i32, pf32, pf32, f32, f32p, f32p, f32p, f32p, i32, i32, pi32, i32]
        fn.restype = i32
    L.orc_residual_rms.argtypes = [f32p, i32, i32]
    L.orc_residual_rms.restype = f32
    L.orc_residual_sumsq.argtypes = [f32p, i32, i32]
    L.orc_residual_sumsq.restype = f32
    for fn in (L.orc_d_dz_B2A, L.orc_d_dz_O2C, L.orc_d_dr_B2B, L.orc_d_dr_B2C, L.orc_d_dr_O2A):
        fn.argtypes = [f32p, f32p, f32p, i32, i32]
    L.orc_d_rcuvdr_O2A.argtypes = [f32p, f32p, f32p, f32p, i32, i32]
    L.orc_cal_eta.argtypes = [f32p] * 6 + [i32, i32]
    L.orc_cal_uw.argtypes = [f32p] * 7 + [i32, i32]
    L.orc_integrate_weight_B.argtypes = [f32p] * 5 + [i32, i32]
    L.orc_integrate_weight_B.restype = f32
    L.orc_cal_sum_Qeta.argtypes = [f32p] * 6 + [i32, i32]
    L.orc_cal_sum_Qeta.restype = f32
    L.orc_cal_wtheta.argtypes = [f32p] * 3 + [i32, i32]
    L.orc_JJ_B.argtypes = [f32p] * 3 + [i32, i32]
    L.orc_rhs_thm.argtypes = [f32p] * 4 + [i32, i32]
    L.orc_rhs_mom.argtypes = [f32p] * 7 + [i32, i32]
    L.orc_cell_average_B.argtypes = [f32p, f32p, i32, i32]
    L.orc_f_from_B.argtypes = [f32p, f32p, i32, i32]
    L.orc_exchange_conversion.argtypes = [f32p] * 7 + [i32, i32]
    L.orc_exchange_conversion.restype = f32
    L.orc_m2_from_C.argtypes = [f32p, f32p, f32p, f32p, C.c_void_p, i32, i32]
    L.orc_theta_tendency.argtypes = [f32p] * 6 + [i32, i32]
    L.orc_relative_theta.argtypes = [f32p] * 5 + [i32, i32]
    L.orc_theta_anomaly_update.argtypes = [f32p, f32] + [f32p] * 9 + [i32, i32]
    _oracle = L
    return L


# ---------------------------------------------------------------------------------------------
# oracle convenience wrappers (numpy in / numpy out)

def constants():
    k = Constants()
    oracle().orc_constants(C.byref(k))
    return k


def profiles(nr, nz, geometry=0, density_mode=0, domain=(0.0, 1.0, 0.0, 1.0)):
    """A13 1-D profiles -> dict(ra, za, rcuva, exner, rho, sin_table, dr, dz)."""
    dom = np.zeros(4, np.float32)
    dom[: len(domain)] = domain
    out = {k: np.zeros(nr if k in ("ra", "rcuva", "sin_table") else nz, np.float32)
           for k in ("ra", "za", "rcuva", "exner", "rho", "sin_table")}
    dr, dz = C.c_float(), C.c_float()
    oracle().orc_profiles(geometry, density_mode, dom, nr, nz, out["ra"], out["za"], out["rcuva"],
                          out["exner"], out["rho"], out["sin_table"], C.byref(dr), C.byref(dz))
    out["dr"], out["dz"] = np.float32(dr.value), np.float32(dz.value)
    return out


def solver_coefficients(A, B, Cc, rcuva, rho):
    nz, nr = A.shape
    sA = np.zeros((nz - 2, nr - 1), np.float32)
    sB = np.zeros((nz - 1, nr - 1), np.float32)
    sC = np.zeros((nz - 1, nr - 2), np.float32)
    oracle().orc_solver_coefficients(A, B, Cc, rcuva, rho, sA, sB, sC, nr, nz)
    return sA, sB, sC


def cal_coe(sA, sB, sC, dx, dy, nx, ny, lib=None):
    """-> coe as (ny, nx, 9) float32 (== Fortran coe(9,nx,ny)); boundary entries zero."""
    coe = np.zeros((ny, nx, 9), np.float32)
    err = C.c_int(-1)
    (lib or oracle()).orc_cal_coe(sA, sB, sC, coe, float(dx), float(dy), nx, ny, C.byref(err))
    assert err.value == 0
    return coe


def solve(dat, coe, f, max_iter, r1, r2, alpha, check_step=100, converge_time=10, lost_rate=5,
          debug=-1, reforder=False):
    """-> (psi, iters, err_bits, r1_out, r2_out, status).  dat is not modified."""
    ny, nx = dat.shape
    psi = np.ascontiguousarray(dat, np.float32).copy()
    wk = np.zeros_like(psi)
    mi, e = C.c_int(int(max_iter)), C.c_int(0)
    a, b = C.c_float(r1), C.c_float(r2)
    fn = oracle().orc_solve_elliptic_reforder if reforder else oracle().orc_solve_elliptic
    st = fn(C.byref(mi), check_step, converge_time, lost_rate, C.byref(a), C.byref(b), float(alpha),
            psi, np.ascontiguousarray(coe), np.ascontiguousarray(f), wk, nx, ny, C.byref(e), debug)
    return psi, mi.value, e.value, np.float32(a.value), np.float32(b.value), st


def cal_eta(psi, prof):
    nz, nr = psi.shape
    eta = np.zeros((nz, nr - 1), np.float32)
    oracle().orc_cal_eta(np.ascontiguousarray(psi), eta, prof["ra"], prof["rcuva"], prof["rho"],
                         prof["exner"], nr, nz)
    return eta


def cal_uw(psi, prof):
    nz, nr = psi.shape
    u = np.zeros((nz - 1, nr), np.float32)
    w = np.zeros((nz, nr - 1), np.float32)
    oracle().orc_cal_uw(np.ascontiguousarray(psi), u, w, prof["ra"], prof["za"], prof["rcuva"],
                        prof["rho"], nr, nz)
    return u, w


def integrate_weight_B(wB, prof):
    nzm, nrm = wB.shape
    return np.float32(oracle().orc_integrate_weight_B(np.ascontiguousarray(wB), prof["ra"], prof["za"],
                                                      prof["rcuva"], prof["rho"], nrm + 1, nzm + 1))


def cal_sum_Qeta(Q, eta, prof):
    nzm, nrm = Q.shape
    return np.float32(oracle().orc_cal_sum_Qeta(np.ascontiguousarray(Q), np.ascontiguousarray(eta),
                                                prof["ra"], prof["za"], prof["rcuva"], prof["rho"],
                                                nrm + 1, nzm + 1))


def diagnose_case(A, B, Cc, bc, forcing=None, *, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL",
                                                       "DENSITY_NORMAL", "BAROTROPIC"),
                  domain=(0.0, 1.0, 0.0, 1.0), r1=5e-3, r2=5e-3, max_iter=100000, alpha=1.0,
                  reforder=False):
    """Whole-case oracle restating src/diagnose/{initialize-variables,diagnose}.f90 for ONE operator
    variant per entry of the returned dict ('BAROTROPIC' and/or 'BAROCLINIC')."""
    nz, nr = A.shape
    geometry = 0 if mode[1] == "CYLINDRICAL" else 1
    dens = 0 if mode[2] == "DENSITY_NORMAL" else 1
    prof = profiles(nr, nz, geometry, dens, domain)
    sA, sB, sC = solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    if mode[0] == "DYNAMIC_EFFICIENCY":
        f = (-B).astype(np.float32)                      # initialize-variables.f90:41
    elif mode[0] == "SECONDARY_CIRCULATION":
        f = forcing
    else:
        f = np.zeros_like(A)                             # quirk 13: defined as zero RHS
    out = {"profiles": prof, "solver": (sA, sB.copy(), sC)}
    variants = {"BAROTROPIC": ["BAROTROPIC"], "BAROCLINIC": ["BAROCLINIC"],
                "BARO_ALL": ["BAROTROPIC", "BAROCLINIC"]}[mode[3]]
    for v in variants:
        b_use = np.zeros_like(sB) if v == "BAROTROPIC" else sB      # diagnose.f90:6,33
        coe = cal_coe(sA, b_use, sC, prof["dr"], prof["dz"], nr, nz)
        psi, it, eb, o1, o2, st = solve(bc, coe, f, max_iter, r1, r2, alpha, reforder=reforder)
        res = {"psi": psi, "iters": it, "err": eb, "r1": o1, "r2": o2, "coe": coe}
        if mode[0] == "DYNAMIC_EFFICIENCY":
            res["eta"] = cal_eta(psi, prof)
        elif mode[0] == "SECONDARY_CIRCULATION":
            res["u"], res["w"] = cal_uw(psi, prof)
        out[v] = res
    return out


# ---------------------------------------------------------------------------------------------
# synthetic inputs, SURVEY.md 8d

def member_inputs(nr, nz, m=0, noise=None):
    """Ensemble member m: A=C=1, bc_init=0,
    B = 1e-2 sin(2 pi (1+m%4) r) sin(pi (1+m%5) z) + 1e-4 xi_m, float64 math then cast.
    m=0 without noise is the test1 formula up to its 3*pi (test1 uses sin(3 pi z): see test1_inputs)."""
    r = np.linspace(0.0, 1.0, nr)
    z = np.linspace(0.0, 1.0, nz)
    B = 1e-2 * np.sin(2.0 * np.pi * (1 + m % 4) * r)[None, :] * np.sin(np.pi * (1 + m % 5) * z)[:, None]
    if noise is None:
        noise = m > 0
    if noise:
        xi = np.random.default_rng(20140911 + m).uniform(-1.0, 1.0, size=(nz, nr))
        B = B + 1e-4 * xi
    A = np.ones((nz, nr), np.float32)
    Cc = np.ones((nz, nr), np.float32)
    bc = np.zeros((nz, nr), np.float32)
    return A, B.astype(np.float32), Cc, bc


def test1_inputs(nr=200, nz=200):
    """The reference fixture formula, test/test1/test-setup.py:42-55 (float64 math, cast to f32)."""
    r = np.linspace(0.0, 1.0, nr)
    z = np.linspace(0.0, 1.0, nz)
    B = np.zeros((nz, nr), np.float32)
    for i in range(nr):
        B[:, i] = 1e-2 * np.sin(2.0 * np.pi * (r[i] - r[0]) / 1.0) * np.sin(3.0 * np.pi * (z - z[0]) / 1.0)
    return (np.ones((nz, nr), np.float32), B, np.ones((nz, nr), np.float32), np.zeros((nz, nr), np.float32))


def rich_inputs(nr, nz, seed=7):
    """Non-trivial A, B, C, bc (all planes vary, bc non-zero on the rim) for kernel-level parity."""
    rng = np.random.default_rng(seed)
    r = np.linspace(0.0, 1.0, nr)[None, :]
    z = np.linspace(0.0, 1.0, nz)[:, None]
    A = (1.0 + 0.3 * np.cos(2 * np.pi * r) * np.sin(np.pi * z) + 0.01 * rng.standard_normal((nz, nr)))
    Cc = (1.5 + 0.4 * np.sin(np.pi * r) * np.cos(2 * np.pi * z) + 0.01 * rng.standard_normal((nz, nr)))
    B = 0.05 * np.sin(2 * np.pi * r) * np.sin(3 * np.pi * z) + 0.002 * rng.standard_normal((nz, nr))
    bc = 1e-3 * np.sin(np.pi * z) * (0.5 + r) + 1e-4 * rng.standard_normal((nz, nr))
    return tuple(np.ascontiguousarray(x, np.float32) for x in (A, B, Cc, bc))


def md5(arr):
    return hashlib.md5(np.ascontiguousarray(arr).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# the reference, when oracle/_ref/ was built (this container; prebuilt files travel to the GPU box)

def have_ref():
    return (REF_DIR / "diagnose").exists() and (REF_DIR / "libref_elliptic.so").exists()


_ref = None


def ref_lib():
    """ctypes handle of the reference's own elliptic_tools (flang-mangled module procedures,
    every argument by reference: xtt-lib-fortran/elliptic_tools.f90:8,64,93)."""
    global _ref
    if _ref is None:
        L = C.CDLL(str(REF_DIR / "libref_elliptic.so"))
        L.cal_coe = getattr(L, "_QMelliptic_toolsPcal_coe")
        L.do_elliptic = getattr(L, "_QMelliptic_toolsPdo_elliptic")
        L.solve_elliptic = getattr(L, "_QMelliptic_toolsPsolve_elliptic")
        L.cal_coe.argtypes = [f32p, f32p, f32p, f32p, pf32, pf32, pi32, pi32, pi32]
        L.do_elliptic.argtypes = [f32p, f32p, f32p, pi32, pi32, pi32]
        L.solve_elliptic.argtypes = [pi32, pi32, pi32, pi32, pf32, pf32, pf32, f32p, f32p, f32p, f32p,
                                     pi32, pi32, pi32, pi32]
        _ref = L
    return _ref


def ref_cal_coe(sA, sB, sC, dx, dy, nx, ny):
    coe = np.zeros((ny, nx, 9), np.float32)
    e = C.c_int(-1)
    ref_lib().cal_coe(sA, sB, sC, coe, C.byref(C.c_float(dx)), C.byref(C.c_float(dy)),
                      C.byref(C.c_int(nx)), C.byref(C.c_int(ny)), C.byref(e))
    return coe


def ref_solve(dat, coe, f, max_iter, r1, r2, alpha, check_step=100, converge_time=10, lost_rate=5, debug=0):
    ny, nx = dat.shape
    psi = np.ascontiguousarray(dat, np.float32).copy()
    wk = np.zeros_like(psi)
    mi, e = C.c_int(int(max_iter)), C.c_int(0)
    a, b = C.c_float(r1), C.c_float(r2)
    ref_lib().solve_elliptic(C.byref(mi), C.byref(C.c_int(check_step)), C.byref(C.c_int(converge_time)),
                             C.byref(C.c_int(lost_rate)), C.byref(a), C.byref(b), C.byref(C.c_float(alpha)),
                             psi, np.ascontiguousarray(coe), np.ascontiguousarray(f), wk,
                             C.byref(C.c_int(nx)), C.byref(C.c_int(ny)), C.byref(e), C.byref(C.c_int(debug)))
    return psi, mi.value, e.value, np.float32(a.value), np.float32(b.value)


def write_diag_txt(path, mode, domain, nr, nz, r1, r2, max_iter, alpha, forcing=False,
                   in_dir=".", out_dir="."):
    """diag.txt for the current program (src/diagnose/read-input.f90)."""
    lines = ["-".join(mode) + "   // mode",
             " ".join("%.9e" % v for v in domain) + " // domain", "%d %d // grid" % (nr, nz),
             "%s // in" % in_dir, "%s // out" % out_dir, "A.bin // A", "B.bin // B", "C.bin // C"]
    if forcing:
        lines.append("forcing.bin // forcing")
    lines += ["bc_init.bin // bc", "%.9e %.9e %d %.9e // criteria" % (r1, r2, max_iter, alpha)]
    Path(path).write_text("\n".join(lines) + "\n")


def run_ref_diagnose(workdir, A, B, Cc, bc, forcing=None, **cfg):
    """Run the reference program on the given inputs; returns (stdout, {filename: array})."""
    workdir = Path(workdir)
    workdir.mkdir(parents=True, exist_ok=True)
    nz, nr = A.shape
    for name, arr in (("A", A), ("B", B), ("C", Cc), ("bc_init", bc)):
        np.ascontiguousarray(arr, np.float32).tofile(workdir / f"{name}.bin")
    if forcing is not None:
        np.ascontiguousarray(forcing, np.float32).tofile(workdir / "forcing.bin")
    mode = cfg.get("mode", ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"))
    write_diag_txt(workdir / "diag.txt", mode, cfg.get("domain", (0.0, 1.0, 0.0, 1.0)), nr, nz,
                   cfg.get("r1", 5e-3), cfg.get("r2", 5e-3), cfg.get("max_iter", 100000),
                   cfg.get("alpha", 1.0), forcing=forcing is not None)
    with open(workdir / "diag.txt") as fin:
        p = subprocess.run([str(REF_DIR / "diagnose")], stdin=fin, cwd=workdir, capture_output=True,
                           text=True, check=True)
    shapes = {"eta": (nz, nr - 1), "rchi": (nz, nr), "rpsi": (nz, nr), "w": (nz, nr - 1), "u": (nz - 1, nr),
              "solver_a": (nz - 2, nr - 1), "solver_b": (nz - 1, nr - 1), "solver_c": (nz - 1, nr - 2)}
    outs = {}
    for fn in os.listdir(workdir):
        if fn.endswith(".bin") and fn not in ("A.bin", "B.bin", "C.bin", "bc_init.bin", "forcing.bin"):
            key = fn.split("-")[0]
            outs[fn] = np.fromfile(workdir / fn, np.float32).reshape(shapes[key])
    return p.stdout, outs
