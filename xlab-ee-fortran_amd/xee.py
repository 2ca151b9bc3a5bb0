"""ctypes view of libxee's C-ABI (include/xee.h).  Plumbing only: numpy arrays <-> C pointers.

There is deliberately no fallback: if lib/libxee.so is missing or a call fails, XeeError is raised
(the product path must fail loudly when the HIP extension is absent).

Array convention (same as the C-ABI): a Fortran array f(n1,n2) is a numpy (n2,n1) C-order float32
array -- the byte layout of the reference's .bin files (xtt-lib-fortran/field_tools.f90:30-76).
"""
import ctypes as C
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

__all__ = ["debug_counters", "Ensemble", "Comm", "comm_unique_id", "comm_available", "comm_library", "comm_set_timeout", "COMM_ID_BYTES", "XeeError", "lib", "lib_path", "device_count", "set_default_option", "reset_default_options", "last_sweep_info", "tier1_release",
           "profiles", "cal_coe", "do_elliptic", "solve_elliptic", "solver_coefficients", "residual_norm", "cal_eta", "cal_uw",
           "integrate_weight_B", "cal_sum_Qeta", "cal_wtheta", "rhs_thermal", "rhs_from_B", "rhs_momentum", "fd_stencil",
           "exchange_conversion", "legacy_averages", "JJ_B", "add_fields", "diagnose_case", "CaseRecord", "RECORD_DTYPE", "Timing", "FIELDS", "FD_STENCILS",
           "BAROTROPIC", "BAROCLINIC", "SOLVE_DEFAULT", "SOLVE_SINGLE_STEP", "SOLVE_KEEP_PSI", "SOLVE_TRACE",
           "quotient_selfcheck", "m2_from_C", "theta_tendency", "relative_theta", "theta_anomaly_update"]

BAROTROPIC, BAROCLINIC = 0, 1
SOLVE_DEFAULT, SOLVE_SINGLE_STEP, SOLVE_KEEP_PSI, SOLVE_TRACE = 0, 1, 2, 4

FIELDS = {"A": 0, "B": 1, "C": 2, "BC_INIT": 3, "FORCING": 4, "PSI": 5, "SOLVER_A": 6, "SOLVER_B": 7,
          "SOLVER_C": 8, "ETA": 9, "U": 10, "W": 11, "Q": 12, "RESIDUAL": 13}
FIELDS.update({f"COE{k}": 16 + k for k in range(9)})

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_pi = C.POINTER(C.c_int)
_pf = C.POINTER(C.c_float)


class XeeError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"libxee status {status}: {text}")
        self.status = status


class CaseRecord(C.Structure):
    """xee_case_record_t (128 bytes) -- the unit of the multi-GPU gather."""
    _fields_ = [("case_id", C.c_int32), ("iters", C.c_int32), ("err_bits", C.c_int32), ("reserved", C.c_int32),
                ("err_now", C.c_float), ("ratio", C.c_float), ("sum_Q", C.c_float), ("sum_Qeta", C.c_float),
                ("efficiency", C.c_float), ("scalars", C.c_float * 23)]


RECORD_DTYPE = np.dtype([("case_id", "<i4"), ("iters", "<i4"), ("err_bits", "<i4"), ("reserved", "<i4"),
                         ("err_now", "<f4"), ("ratio", "<f4"), ("sum_Q", "<f4"), ("sum_Qeta", "<f4"),
                         ("efficiency", "<f4"), ("scalars", "<f4", (23,))])
assert RECORD_DTYPE.itemsize == 128 and C.sizeof(CaseRecord) == 128


class Timing(C.Structure):
    _fields_ = [("solve_ms", C.c_double), ("sweep_ms", C.c_double), ("sweep_launches", C.c_int64),
                ("iterations", C.c_int64), ("check_launches", C.c_int64), ("redo_tiles", C.c_int64),
                ("inline_redo_tiles", C.c_int64), ("tiles", C.c_int64), ("sweep_streams", C.c_int64), ("z_launches", C.c_int64),
                ("z_ms", C.c_double), ("z_timed_launches", C.c_int64), ("shared_planes", C.c_int64)]


_lib = None


def lib_path():
    return _HERE / "lib" / "libxee.so"


def lib():
    """Load lib/libxee.so (built by `make -C xlab-ee-fortran_amd lib` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise XeeError(-1, f"{p} is missing: build it with `make -C {_HERE} lib` (no CPU fallback exists)")
    L = C.CDLL(str(p))
    L.xee_last_error.restype = C.c_char_p
    L.xee_device_count.argtypes = [_pi]
    L.xee_profiles.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _pf, _pf]
    L.xee_cal_coe.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int, _pi]
    L.xee_do_elliptic.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, _pi]
    L.xee_solve_elliptic.argtypes = [_pi, C.c_int, C.c_int, C.c_int, _pf, _pf, C.c_float, _f32p, _f32p, _f32p, _f32p,
                                     C.c_int, C.c_int, _pi, C.c_int]
    L.xee_residual_norm.argtypes = [_f32p, C.c_int, C.c_int, _pf, _pf, _pi]
    L.xee_judge_error.argtypes = [C.c_int]
    L.xee_judge_error.restype = None
    L.xee_solver_coefficients.argtypes = [_f32p] * 8 + [C.c_int, C.c_int]
    L.xee_cal_eta.argtypes = [_f32p] * 6 + [C.c_int, C.c_int]
    L.xee_cal_uw.argtypes = [_f32p] * 7 + [C.c_int, C.c_int]
    L.xee_integrate_weight_B.argtypes = [_f32p] * 5 + [C.c_int, C.c_int, _pf]
    L.xee_cal_sum_Qeta.argtypes = [_f32p] * 6 + [C.c_int, C.c_int, _pf]
    L.xee_cal_wtheta.argtypes = [_f32p] * 3 + [C.c_int, C.c_int]
    L.xee_rhs_thermal.argtypes = [_f32p] * 4 + [C.c_int, C.c_int]
    L.xee_rhs_from_B.argtypes = [_f32p] * 2 + [C.c_int, C.c_int]
    L.xee_rhs_momentum.argtypes = [_f32p] * 5 + [C.c_int, C.c_int]
    L.xee_fd_stencil.argtypes = [C.c_int] + [_f32p] * 5 + [C.c_int, C.c_int]
    L.xee_exchange_conversion.argtypes = [_f32p] * 7 + [_pf, C.c_int, C.c_int]
    L.xee_exchange_conversion_legacy.argtypes = [_f32p] * 7 + [_pf, C.c_int, C.c_int]
    vp = C.c_void_p
    L.xee_ensemble_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    L.xee_ensemble_destroy.argtypes = [vp]
    L.xee_ensemble_set_stream.argtypes = [vp, vp]
    L.xee_ensemble_set_case_ids.argtypes = [vp, C.c_int]
    L.xee_set_default_option.argtypes = [C.c_char_p, C.c_int]
    L.xee_ensemble_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.xee_ensemble_sweep_info.argtypes = [vp, C.c_char_p, C.c_int, _pi, _pi, _pi]
    L.xee_ensemble_set_profiles.argtypes = [vp, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float]
    L.xee_ensemble_upload.argtypes = [vp, C.c_int, C.c_int, C.c_int, _f32p]
    L.xee_ensemble_download.argtypes = [vp, C.c_int, C.c_int, C.c_int, _f32p]
    L.xee_ensemble_device_plane.argtypes = [vp, C.c_int, C.POINTER(vp), _pi, C.POINTER(C.c_size_t), _pi, _pi]
    L.xee_ensemble_forcing_from_B.argtypes = [vp]
    L.xee_ensemble_forcing_thermal.argtypes = [vp]
    L.xee_ensemble_forcing_zero.argtypes = [vp]
    L.xee_ensemble_member_window.argtypes = [vp, C.c_int, C.c_int]
    L.xee_ensemble_field_save.argtypes = [vp, C.c_int, C.c_int]
    L.xee_ensemble_field_restore.argtypes = [vp, C.c_int, C.c_int]
    L.xee_ensemble_assemble.argtypes = [vp, C.c_int]
    L.xee_ensemble_assemble_solver.argtypes = [vp, C.c_int]
    L.xee_legacy_averages.argtypes = [_f32p] * 7 + [C.c_int, C.c_int]
    L.xee_JJ_B.argtypes = [_f32p] * 3 + [C.c_int, C.c_int]
    L.xee_add_fields.argtypes = [_f32p] * 3 + [C.c_size_t]
    L.xee_ensemble_solve.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
    L.xee_ensemble_solve_sor.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
    L.xee_ensemble_cal_eta.argtypes = [vp]
    L.xee_ensemble_cal_uw.argtypes = [vp]
    L.xee_ensemble_ee_scalars.argtypes = [vp]
    L.xee_ensemble_records.argtypes = [vp, C.c_void_p]
    L.xee_ensemble_timing.argtypes = [vp, C.POINTER(Timing)]
    L.xee_ensemble_time_sweeps.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.xee_m2_from_C.argtypes = [_f32p, _f32p, _f32p, C.c_void_p, _f32p, C.c_int, C.c_int]
    L.xee_theta_tendency.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int]
    L.xee_relative_theta.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int]
    L.xee_theta_anomaly_update.argtypes = [_f32p, C.c_float, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                           C.c_int, C.c_int]
    L.xee_comm_unique_id.argtypes = [C.c_char_p]
    L.xee_comm_library.argtypes = [C.c_char_p, C.c_int]
    L.xee_comm_set_timeout.argtypes = [C.c_double]
    L.xee_comm_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.xee_comm_destroy.argtypes = [vp]
    L.xee_gather_scalars.argtypes = [vp, C.c_void_p, C.c_int, C.c_void_p]
    L.xee_quotient_selfcheck.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    _lib = L
    return L


def _chk(status):
    if status != 0:
        raise XeeError(status, lib().xee_last_error().decode(errors="replace"))


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def device_count():
    n = C.c_int(0)
    _chk(lib().xee_device_count(C.byref(n)))
    return n.value


def last_sweep_info():
    """The sweep kernel this thread's last Tier-1 solve_elliptic launched (xee_last_sweep_info)."""
    buf = C.create_string_buffer(64)
    T, tw, th = C.c_int(), C.c_int(), C.c_int()
    lib().xee_last_sweep_info.argtypes = [C.c_char_p, C.c_int, _pi, _pi, _pi]
    _chk(lib().xee_last_sweep_info(buf, 64, C.byref(T), C.byref(tw), C.byref(th)))
    name = buf.value.decode()
    return {"name": name, "kernel": name.split("<")[0], "T": T.value, "tile": f"{tw.value}x{th.value}"}


def tier1_release():
    """Free this thread's cached Tier-1 ensemble and cal_coe copy (xee_tier1_release)."""
    _chk(lib().xee_tier1_release())


def set_default_option(key, value):
    """Process-wide default of a run-time option (include/xee.h); copied into ensembles created afterwards."""
    _chk(lib().xee_set_default_option(key.encode(), int(value)))


def reset_default_options():
    _chk(lib().xee_reset_default_options())


# ---- Tier 1 wrappers (same names and argument meaning as the reference procedures) ----------

def profiles(nr, nz, geometry=0, density_mode=0, domain=(0.0, 1.0, 0.0, 1.0)):
    """initialize-variables.f90:44-67 -> dict(ra, za, rcuva, exner, rho, sin_table, dr, dz)."""
    dom = np.zeros(4, np.float32)
    dom[: len(domain)] = domain
    out = {k: np.zeros(nr if k in ("ra", "rcuva", "sin_table") else nz, np.float32)
           for k in ("ra", "za", "rcuva", "exner", "rho", "sin_table")}
    dr, dz = C.c_float(), C.c_float()
    _chk(lib().xee_profiles(geometry, density_mode, dom, nr, nz, out["ra"], out["za"], out["rcuva"],
                            out["exner"], out["rho"], out["sin_table"], C.byref(dr), C.byref(dz)))
    out["dr"], out["dz"] = np.float32(dr.value), np.float32(dz.value)
    return out


def cal_coe(a, b, c, dx, dy, nx, ny, coe=None):
    """elliptic_tools.f90:8-60 -> coe (ny,nx,9); rim entries keep the bytes of `coe` (zeros by default)."""
    coe = np.zeros((ny, nx, 9), np.float32) if coe is None else _a(coe)
    err = C.c_int(-1)
    _chk(lib().xee_cal_coe(_a(a), _a(b), _a(c), coe, float(dx), float(dy), nx, ny, C.byref(err)))
    return coe, err.value


def do_elliptic(psi, coe, outdat=None):
    ny, nx = psi.shape
    out = np.zeros((ny, nx), np.float32) if outdat is None else _a(outdat).copy()
    err = C.c_int(0)
    _chk(lib().xee_do_elliptic(_a(psi), _a(coe), out, nx, ny, C.byref(err)))
    return out, err.value


def solve_elliptic(dat, coe, f, max_iter, r1, r2, alpha, check_step=100, converge_time=10, lost_rate=5, debug=0):
    """elliptic_tools.f90:93-265 -> (psi, iters, err_bits, r1_out, r2_out)."""
    ny, nx = dat.shape
    psi = _a(dat).copy()
    wk = np.zeros_like(psi)
    mi, e = C.c_int(int(max_iter)), C.c_int(0)
    a, b = C.c_float(r1), C.c_float(r2)
    _chk(lib().xee_solve_elliptic(C.byref(mi), check_step, converge_time, lost_rate, C.byref(a), C.byref(b),
                                  float(alpha), psi, _a(coe), _a(f), wk, nx, ny, C.byref(e), debug))
    return psi, mi.value, e.value, np.float32(a.value), np.float32(b.value)


def residual_norm(res):
    """elliptic_tools.f90:193-199 -> (sum_sq, rms, slow_chunks), order-exact."""
    ny, nx = res.shape
    s, r, slow = C.c_float(), C.c_float(), C.c_int()
    _chk(lib().xee_residual_norm(_a(res), nx, ny, C.byref(s), C.byref(r), C.byref(slow)))
    return np.float32(s.value), np.float32(r.value), slow.value


def solver_coefficients(A, B, Cc, rcuva, rho):
    nz, nr = A.shape
    sA = np.zeros((nz - 2, nr - 1), np.float32)
    sB = np.zeros((nz - 1, nr - 1), np.float32)
    sC = np.zeros((nz - 1, nr - 2), np.float32)
    _chk(lib().xee_solver_coefficients(_a(A), _a(B), _a(Cc), _a(rcuva), _a(rho), sA, sB, sC, nr, nz))
    return sA, sB, sC


def cal_eta(rchi, prof):
    nz, nr = rchi.shape
    eta = np.zeros((nz, nr - 1), np.float32)
    _chk(lib().xee_cal_eta(_a(rchi), eta, prof["ra"], prof["rcuva"], prof["rho"], prof["exner"], nr, nz))
    return eta


def cal_uw(rpsi, prof):
    nz, nr = rpsi.shape
    u = np.zeros((nz - 1, nr), np.float32)
    w = np.zeros((nz, nr - 1), np.float32)
    _chk(lib().xee_cal_uw(_a(rpsi), u, w, prof["ra"], prof["za"], prof["rcuva"], prof["rho"], nr, nz))
    return u, w


def integrate_weight_B(weight, prof):
    nzm, nrm = weight.shape
    s = C.c_float()
    _chk(lib().xee_integrate_weight_B(_a(weight), prof["ra"], prof["za"], prof["rcuva"], prof["rho"],
                                      nrm + 1, nzm + 1, C.byref(s)))
    return np.float32(s.value)


def cal_sum_Qeta(Q, eta, prof):
    nzm, nrm = Q.shape
    s = C.c_float()
    _chk(lib().xee_cal_sum_Qeta(_a(Q), _a(eta), prof["ra"], prof["za"], prof["rcuva"], prof["rho"],
                                nrm + 1, nzm + 1, C.byref(s)))
    return np.float32(s.value)


def cal_wtheta(w_A, theta_B):
    nz, nrm = w_A.shape
    out = np.zeros((nz - 1, nrm), np.float32)
    _chk(lib().xee_cal_wtheta(_a(w_A), _a(theta_B), out, nrm + 1, nz))
    return out


def rhs_thermal(Q_B, ra, exner):
    nzm, nrm = Q_B.shape
    out = np.zeros((nzm + 1, nrm + 1), np.float32)
    _chk(lib().xee_rhs_thermal(_a(Q_B), _a(ra), _a(exner), out, nrm + 1, nzm + 1))
    return out


def rhs_from_B(b_B):
    nzm, nrm = b_B.shape
    out = np.zeros((nzm + 1, nrm + 1), np.float32)
    _chk(lib().xee_rhs_from_B(_a(b_B), out, nrm + 1, nzm + 1))
    return out


FD_STENCILS = {"d_dz_B2A": 0, "d_dz_O2C": 1, "d_dr_B2B": 2, "d_dr_B2C": 3, "d_dr_O2A": 4, "d_rcuvdr_O2A": 5}


def fd_stencil(name, from_dat, prof, to_dat=None):
    """quick-tools2.f90:1-85.  `to_dat` seeds the entries the reference leaves untouched (default zeros)."""
    which = FD_STENCILS[name]
    n2f, n1f = from_dat.shape
    nr, nz = (n1f + 1, n2f + 1) if name in ("d_dz_B2A", "d_dr_B2B", "d_dr_B2C") else (n1f, n2f)
    shape = {"d_dz_B2A": (nz, nr - 1), "d_dz_O2C": (nz - 1, nr), "d_dr_B2B": (nz - 1, nr - 1),
             "d_dr_B2C": (nz - 1, nr), "d_dr_O2A": (nz, nr - 1), "d_rcuvdr_O2A": (nz, nr - 1)}[name]
    out = np.zeros(shape, np.float32) if to_dat is None else _a(to_dat).copy()
    _chk(lib().xee_fd_stencil(which, _a(from_dat), out, prof["ra"], prof["za"], prof["rcuva"], nr, nz))
    return out


def rhs_momentum(m2_B, F_B, prof):
    nzm, nrm = m2_B.shape
    out = np.zeros((nzm + 1, nrm + 1), np.float32)
    _chk(lib().xee_rhs_momentum(_a(m2_B), _a(F_B), prof["za"], prof["rcuva"], out, nrm + 1, nzm + 1))
    return out


def exchange_conversion(rpsi, rchi, rhoC, prof, legacy_int=False):
    """old-diagnose/diagnose.f90:1143-1174 -> (bndconv (2, nr-1), sum); legacy_int reproduces the reference's
    INTEGER r, dr, dz (:1146), default is the documented fix with real metric factors."""
    nz, nr = rpsi.shape
    bnd = np.zeros((2, nr - 1), np.float32)
    s = C.c_float()
    fn = lib().xee_exchange_conversion_legacy if legacy_int else lib().xee_exchange_conversion
    _chk(fn(_a(rpsi), _a(rchi), _a(rhoC), prof["ra"], prof["za"], prof["rho"], bnd,
                                       C.byref(s), nr, nz))
    return bnd, np.float32(s.value)


def legacy_averages(A, B, Cc):
    """old-diagnose/diagnose.f90:329-354 -> (rhoA_A (nz,nr-1), rhoB_C (nz-1,nr), rhoB_B (nz-1,nr-1), rhoC_C (nz-1,nr))."""
    nz, nr = A.shape
    out = (np.zeros((nz, nr - 1), np.float32), np.zeros((nz - 1, nr), np.float32), np.zeros((nz - 1, nr - 1), np.float32),
           np.zeros((nz - 1, nr), np.float32))
    _chk(lib().xee_legacy_averages(_a(A), _a(B), _a(Cc), *out, nr, nz))
    return out


def JJ_B(Q_B, exner):
    """old-diagnose/diagnose.f90:383-387 -> Q / (Cp exner) on B cells."""
    nzm, nrm = Q_B.shape
    out = np.zeros((nzm, nrm), np.float32)
    _chk(lib().xee_JJ_B(_a(Q_B), _a(exner), out, nrm + 1, nzm + 1))
    return out


def add_fields(a, b):
    a, b = _a(a), _a(b)
    assert a.shape == b.shape
    out = np.zeros_like(a)
    _chk(lib().xee_add_fields(a, b, out, a.size))
    return out


# ---- Tier 2: device-resident ensemble ---------------------------------------------------------

def m2_from_C(rhoC_C, prof, sin_table=None):
    """old-diagnose/diagnose.f90:359-380 (intended form) -> m2 on B cells (nz-1, nr-1)."""
    nzm, nr = rhoC_C.shape
    out = np.zeros((nzm, nr - 1), np.float32)
    st = None if sin_table is None else _a(sin_table).ctypes.data_as(C.c_void_p)
    _chk(lib().xee_m2_from_C(_a(rhoC_C), prof["ra"], prof["rcuva"], st, out, nr, nzm + 1))
    return out


def theta_tendency(JJ_B, rhoA_A, w_A, rhoB_C, u_C):
    """old-diagnose/diagnose.f90:470-479 -> dtheta/dt on B cells (nz-1, nr-1)."""
    nzm, nrm = JJ_B.shape
    out = np.zeros((nzm, nrm), np.float32)
    _chk(lib().xee_theta_tendency(_a(JJ_B), _a(rhoA_A), _a(w_A), _a(rhoB_C), _a(u_C), out, nrm + 1, nzm + 1))
    return out


def relative_theta(dtheta_dz_A, dtheta_dr_C, prof):
    """old-diagnose/diagnose.f90:893-912 (relativeTheta) -> theta on B cells."""
    nz, nrm = dtheta_dz_A.shape
    out = np.zeros((nz - 1, nrm), np.float32)
    _chk(lib().xee_relative_theta(_a(dtheta_dz_A), _a(dtheta_dr_C), prof["ra"], prof["za"], out, nrm + 1, nz))
    return out


def theta_anomaly_update(dtheta_dt_B, testing_dt, rhoB_B, rhoA_A, rhoB_C, prof):
    """old-diagnose/diagnose.f90:489-518 -> dict(theta, rhoB_B, rhoA_A, rhoB_C, b_anomaly, solver_b_anomaly); inputs are not modified."""
    nzm, nrm = dtheta_dt_B.shape
    th, bb, ra_, rc = (np.array(_a(x), copy=True) for x in (dtheta_dt_B, rhoB_B, rhoA_A, rhoB_C))
    ban, sb = np.zeros((nzm, nrm), np.float32), np.zeros((nzm, nrm), np.float32)
    _chk(lib().xee_theta_anomaly_update(th, float(testing_dt), bb, ra_, rc, ban, sb, prof["ra"], prof["za"], prof["rcuva"],
                                        prof["rho"], nrm + 1, nzm + 1))
    return {"theta": th, "rhoB_B": bb, "rhoA_A": ra_, "rhoB_C": rc, "b_anomaly": ban, "solver_b_anomaly": sb}


def quotient_selfcheck(seed, blocks, per_thread, d_exp_lo, d_exp_hi, device=-1):
    """Count the operand pairs on which the blocked sweep's hoisted-reciprocal quotient differs from the IEEE
    division (csrc/xee_sweepq.hip); returns (mismatches, (a, d) of one failing pair)."""
    bad, first = C.c_uint64(), (C.c_float * 2)()
    _chk(lib().xee_quotient_selfcheck(device, seed, blocks, per_thread, d_exp_lo, d_exp_hi, C.byref(bad), first))
    return bad.value, (first[0], first[1])


COMM_ID_BYTES = 128


def comm_unique_id():
    """Rank 0: the 128-byte RCCL rendezvous id to hand to the other ranks (xee_comm_unique_id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(lib().xee_comm_unique_id(buf))
    return buf.raw


def debug_counters():
    """-> dict(ensembles, tier1_stage_allocs, thread_exit_releases): process-wide counts (xee_debug_counters)"""
    out = (C.c_int64 * 4)()
    _chk(lib().xee_debug_counters(out))
    return {"ensembles": out[0], "tier1_stage_allocs": out[1], "thread_exit_releases": out[2]}


def comm_available():
    """-> (ok, reason): can this process join a communicator (librccl loadable, a device visible)?  Not collective."""
    st = lib().xee_comm_available()
    return st == 0, ("" if st == 0 else (lib().xee_last_error() or b"").decode())


def comm_library():
    """the file the RCCL entry points were resolved from"""
    buf = C.create_string_buffer(512)
    _chk(lib().xee_comm_library(buf, 512))
    return buf.value.decode()


def comm_set_timeout(seconds):
    _chk(lib().xee_comm_set_timeout(float(seconds)))


class Comm:
    """RCCL communicator of the ensemble ranks (one process per GPU) for the record gather, xee_comm_* / xee_gather_scalars."""

    def __init__(self, unique_id, rank, world, device=-1):
        assert len(unique_id) == COMM_ID_BYTES
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        _chk(lib().xee_comm_create(C.byref(self._h), unique_id, rank, world, device))

    def gather(self, records, per_rank):
        """records: this rank's RECORD_DTYPE array (<= per_rank entries); returns the world*per_rank records of all ranks,
        rank by rank, short blocks padded with case_id = -1."""
        pad = np.zeros(per_rank, RECORD_DTYPE)
        pad["case_id"] = -1
        pad[: len(records)] = records
        out = np.zeros(self.world * per_rank, RECORD_DTYPE)
        _chk(lib().xee_gather_scalars(self._h, pad.ctypes.data_as(C.c_void_p), per_rank, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().xee_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ensemble:
    """N independent cases of one grid on one GPU (xee_ensemble_* of include/xee.h)."""

    def __init__(self, nr, nz, members=1, device=-1):
        self.nr, self.nz, self.members = nr, nz, members
        self._h = C.c_void_p()
        _chk(lib().xee_ensemble_create(C.byref(self._h), nr, nz, members, device))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().xee_ensemble_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _shape(self, field):
        nr, nz = self.nr, self.nz
        return {"SOLVER_A": (nz - 2, nr - 1), "SOLVER_B": (nz - 1, nr - 1), "SOLVER_C": (nz - 1, nr - 2),
                "ETA": (nz, nr - 1), "U": (nz - 1, nr), "W": (nz, nr - 1), "Q": (nz - 1, nr - 1)}.get(field, (nz, nr))

    def set_stream(self, hip_stream):
        _chk(lib().xee_ensemble_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_case_ids(self, first):
        _chk(lib().xee_ensemble_set_case_ids(self._h, first))

    def set_option(self, key, value):
        _chk(lib().xee_ensemble_set_option(self._h, key.encode(), int(value)))

    def sweep_info(self):
        """-> dict(name, kernel, T, tile): the sweep kernel the last solve launched."""
        buf = C.create_string_buffer(64)
        T, tw, th = C.c_int(), C.c_int(), C.c_int()
        _chk(lib().xee_ensemble_sweep_info(self._h, buf, 64, C.byref(T), C.byref(tw), C.byref(th)))
        name = buf.value.decode()
        return {"name": name, "kernel": name.split("<")[0], "T": T.value, "tile": f"{tw.value}x{th.value}"}

    def set_profiles(self, prof):
        self.prof = prof
        _chk(lib().xee_ensemble_set_profiles(self._h, prof["ra"], prof["za"], prof["rcuva"], prof["rho"],
                                             prof["exner"], float(prof["dr"]), float(prof["dz"])))

    def upload(self, field, arr, first=0):
        arr = _a(arr)
        shp = self._shape(field)
        if arr.ndim == 2:
            arr = arr[None]
        assert arr.shape[1:] == shp, (field, arr.shape, shp)
        _chk(lib().xee_ensemble_upload(self._h, FIELDS[field], first, arr.shape[0], arr))

    def download(self, field, first=0, count=None):
        count = self.members - first if count is None else count
        out = np.zeros((count,) + self._shape(field), np.float32)
        _chk(lib().xee_ensemble_download(self._h, FIELDS[field], first, count, out))
        return out

    def device_plane(self, field):
        """-> (device pointer, pitch, member_stride, n1, n2) for in-place fills by a GPU producer."""
        p, pitch, ms, n1, n2 = C.c_void_p(), C.c_int(), C.c_size_t(), C.c_int(), C.c_int()
        _chk(lib().xee_ensemble_device_plane(self._h, FIELDS[field], C.byref(p), C.byref(pitch), C.byref(ms),
                                             C.byref(n1), C.byref(n2)))
        return p.value, pitch.value, ms.value, n1.value, n2.value

    def forcing_from_B(self):
        _chk(lib().xee_ensemble_forcing_from_B(self._h))

    def forcing_thermal(self):
        _chk(lib().xee_ensemble_forcing_thermal(self._h))

    def forcing_zero(self):
        _chk(lib().xee_ensemble_forcing_zero(self._h))

    def member_window(self, first=0, count=0):
        """forcing_* / field_save / field_restore act on members [first, first+count) only (count 0: all)."""
        _chk(lib().xee_ensemble_member_window(self._h, first, count))

    def field_save(self, field, slot):
        """Keep a device-resident copy of `field` (all members) in stash slot `slot` (0..63)."""
        _chk(lib().xee_ensemble_field_save(self._h, FIELDS[field], slot))

    def field_restore(self, field, slot):
        _chk(lib().xee_ensemble_field_restore(self._h, FIELDS[field], slot))

    def assemble(self, variant):
        _chk(lib().xee_ensemble_assemble(self._h, variant))

    def assemble_solver(self, use_b):
        """cal_coe from the solver coefficients currently in SOLVER_A/_B/_C (use_b False: solverB_B = 0)."""
        _chk(lib().xee_ensemble_assemble_solver(self._h, 1 if use_b else 0))

    def solve(self, max_iter, r1, r2, alpha, check_step=100, converge_time=10, lost_rate=5, flags=0):
        _chk(lib().xee_ensemble_solve(self._h, int(max_iter), check_step, converge_time, lost_rate,
                                      float(r1), float(r2), float(alpha), flags))
        return self.records()

    def solve_sor(self, max_iter, r1, r2, omega, check_step=100, converge_time=10, lost_rate=5, flags=0):
        """Opt-in, non-parity 4-colour SOR (same fixed point, far fewer iterations)."""
        _chk(lib().xee_ensemble_solve_sor(self._h, int(max_iter), check_step, converge_time, lost_rate,
                                          float(r1), float(r2), float(omega), flags))
        return self.records()

    def cal_eta(self):
        _chk(lib().xee_ensemble_cal_eta(self._h))

    def cal_uw(self):
        _chk(lib().xee_ensemble_cal_uw(self._h))

    def ee_scalars(self):
        _chk(lib().xee_ensemble_ee_scalars(self._h))
        return self.records()

    def ee_slow_chunks(self):
        """-> (sum_Q_slow, sum_Qeta_slow) int arrays: chunks the last ee_scalars added term by term (fast-path diagnostic)."""
        a, b = np.zeros(self.members, np.int32), np.zeros(self.members, np.int32)
        lib().xee_ensemble_ee_slow_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(lib().xee_ensemble_ee_slow_chunks(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        return a, b

    def records(self):
        rec = np.zeros(self.members, RECORD_DTYPE)
        _chk(lib().xee_ensemble_records(self._h, rec.ctypes.data_as(C.c_void_p)))
        return rec

    def timing(self):
        t = Timing()
        _chk(lib().xee_ensemble_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def time_sweeps(self, iters, variant=0, alpha=1.0):
        ms, n = C.c_double(), C.c_int64()
        _chk(lib().xee_ensemble_time_sweeps(self._h, iters, variant, float(alpha), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def diagnose_case(A, B, Cc, bc, forcing=None, *, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"),
                  domain=(0.0, 1.0, 0.0, 1.0), r1=5e-3, r2=5e-3, max_iter=100000, alpha=1.0, flags=0, Q=None, options=None):
    """Host-side mirror of the reference driver for a batch of cases sharing one grid:
    src/diagnose/initialize-variables.f90:33-95 + diagnose.f90:2-55, on the GPU through the
    ensemble API.  A, B, Cc, bc (and forcing, Q) are (members, nz, nr) or (nz, nr)."""
    A, B, Cc, bc = (_a(x)[None] if np.ndim(x) == 2 else _a(x) for x in (A, B, Cc, bc))
    members, nz, nr = A.shape
    geometry = 0 if mode[1] == "CYLINDRICAL" else 1
    dens = 0 if mode[2] == "DENSITY_NORMAL" else 1
    prof = profiles(nr, nz, geometry, dens, domain)
    out = {"profiles": prof}
    with Ensemble(nr, nz, members) as ens:
        for k, v in (options or {}).items():
            ens.set_option(k, v)
        ens.set_profiles(prof)
        ens.upload("A", A); ens.upload("B", B); ens.upload("C", Cc); ens.upload("BC_INIT", bc)
        if mode[0] == "DYNAMIC_EFFICIENCY":
            ens.forcing_from_B()                                   # initialize-variables.f90:41
        elif mode[0] == "SECONDARY_CIRCULATION":
            ens.upload("FORCING", _a(forcing)[None] if np.ndim(forcing) == 2 else _a(forcing))
        else:
            ens.upload("FORCING", np.zeros_like(A))                # quirk 13: zero RHS
        if Q is not None:
            ens.upload("Q", _a(Q)[None] if np.ndim(Q) == 2 else _a(Q))
        variants = {"BAROTROPIC": [("BAROTROPIC", BAROTROPIC)], "BAROCLINIC": [("BAROCLINIC", BAROCLINIC)],
                    "BARO_ALL": [("BAROTROPIC", BAROTROPIC), ("BAROCLINIC", BAROCLINIC)]}[mode[3]]
        for name, v in variants:
            ens.assemble(v)                                        # K1 + cal_coe, diagnose.f90:6-7,33-34
            if "solver" not in out:
                out["solver"] = (ens.download("SOLVER_A"), ens.download("SOLVER_B"), ens.download("SOLVER_C"))
            rec = ens.solve(max_iter, r1, r2, alpha, flags=flags)  # diagnose.f90:9-15
            res = {"psi": ens.download("PSI"), "iters": rec["iters"].copy(), "err": rec["err_bits"].copy(),
                   "r1": rec["err_now"].copy(), "r2": rec["ratio"].copy(), "timing": ens.timing(), "sweep": ens.sweep_info()}
            if mode[0] == "DYNAMIC_EFFICIENCY":
                ens.cal_eta()
                res["eta"] = ens.download("ETA")
                if Q is not None:
                    rec = ens.ee_scalars()
                    res["sum_Q"], res["sum_Qeta"], res["efficiency"] = rec["sum_Q"].copy(), rec["sum_Qeta"].copy(), rec["efficiency"].copy()
            elif mode[0] == "SECONDARY_CIRCULATION":
                ens.cal_uw()
                res["u"], res["w"] = ens.download("U"), ens.download("W")
            out[name] = res
    return out
