"""The reference's LEGACY program -- the "energy-efficiency diagnostic sweep" proper -- on the GPU, for a BATCH of cases:
src/old-diagnose/diagnose.f90:283-841 (TENDENCY mode).

    source terms (A11)   sum_Q :284; solver coefficients :289-312; staggered averages :329-354; m2 :359-380 (intended form);
                         J = Q/(Cp exner) :383-387; RHS_rpsi_thm :395-406; RHS_rpsi_mom :412-436
    Stage I              psi for thm + mom on the full-B operator :449-463
    Stage II             u, w :468; dtheta/dt :470-479; anomaly dB, updated rhoA/rhoB, theta :489-518
    Stage III            f_basic, f_anomaly :523-541; up to six chi solves on {L(B=0), L(B0+dB)} x {0 with the chi boundary,
                         -dB, -B0} :548-673, each followed by cal_eta and sum(Q eta)
    integral check       psi solves on both operators, w*theta sums :679-725
    exchange conversion  methods 1 and 2 :730-772
    efficiency.txt       :779-841, every label (readable by xee_tools.read_efficiency / the reference's XEffReader)

The stage order lives in libxee itself -- xee_legacy_run, csrc/xee_legacy.hip, which the Fortran host bin/xee-old-diagnose
calls too -- where the solves run as one device-resident ensemble whose members are the cases of the batch (each stops on
its own criterion) and the O(N) stencils go through the Tier-1 entry points.  This module packs numpy arrays into that call
and unpacks its result; there is no CPU fallback (libxee raises without a GPU).

First guesses.  `chained=True` (default) is the program's own order -- rchi / rpsi are carried from one solve to the next
("the order ... is such that the initial guessing field is better", :545) -- and reproduces the legacy binary's files bit
for bit and its efficiency.txt to the printed digits (tests/test_gpu_legacy_diagnose.py against
tests/golden/legacy_64x48_*.npz).  `chained=False` starts every solve from its own boundary field: the
solves of one stage are then independent.

Solver mapping (old-diagnose/xtt-lib/elliptic_tools.f90:93-302 onto xee_ensemble_solve): strategy 1 == r1 = r, r2 off,
converge_time 1; strategy 2 == r1 off, r2 = r, converge_time 10, lost_rate 5; strategies 3/4 (max norm over the whole
array) are not offered; max_iter must be a multiple of 100.

Defined where the legacy program has undefined behaviour: Q and F are B-cell fields as given (to
reproduce a legacy run pass `sheared(record)`: the program reads nr x nz records into (nr-1,nz-1) arrays); m2 in its
intended form; the "w/ boundary" sums are 0 without a chi boundary file; `legacy_int=True` keeps the program's INTEGER
r, dr, dz in the exchange conversion (False: the documented fix with real metric factors).
"""
import ctypes as C
import os
import time

import numpy as np

from . import xee as X

f32 = np.float32


def sheared(record, nr, nz):
    """What the legacy program holds in Q_in / F_in after reading an nr x nz record into an (nr-1,nz-1) array
    (:211-212 vs :241-242): the first (nr-1)(nz-1) floats of the record, column-major."""
    flat = np.ascontiguousarray(record, f32).reshape(-1)
    return np.ascontiguousarray(flat[: (nr - 1) * (nz - 1)].reshape(nz - 1, nr - 1))


class _Config(C.Structure):
    """xee_legacy_config_t (include/xee.h)"""
    _fields_ = [("geometry", C.c_int), ("density_mode", C.c_int), ("domain", C.c_float * 4), ("testing_dt", C.c_float),
                ("op_mode", C.c_int),
                ("psi_strategy", C.c_int), ("psi_r", C.c_float), ("psi_max_iter", C.c_int), ("psi_alpha", C.c_float),
                ("chi_strategy", C.c_int), ("chi_r", C.c_float), ("chi_max_iter", C.c_int), ("chi_alpha", C.c_float),
                ("legacy_int", C.c_int), ("chained", C.c_int), ("device", C.c_int), ("first_case_id", C.c_int)]


def _lib():
    L = X.lib()
    if not getattr(L, "_legacy_bound", False):
        vp, fp = C.c_void_p, np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        L.xee_legacy_run.argtypes = [C.POINTER(vp), C.POINTER(_Config), C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, vp, vp]
        L.xee_legacy_destroy.argtypes = [vp]
        L.xee_legacy_field_count.argtypes = [vp, C.POINTER(C.c_int)]
        L.xee_legacy_field_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.xee_legacy_field.argtypes = [vp, C.c_int, C.c_char_p, fp]
        L.xee_legacy_scalar_count.argtypes = [vp, C.POINTER(C.c_int)]
        L.xee_legacy_scalar_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
        L.xee_legacy_scalar.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(C.c_float)]
        L.xee_legacy_relax.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
        L.xee_legacy_record.argtypes = [vp, C.c_int, C.c_void_p]
        L.xee_legacy_efficiency_text.argtypes = [vp, C.c_int, C.c_double, C.c_char_p, C.c_int]
        L.xee_legacy_write_outputs.argtypes = [vp, C.c_int, C.c_char_p, C.c_double]
        L._legacy_bound = True
    return L


def _chk(status):
    if status != 0:
        raise X.XeeError(status, (X.lib().xee_last_error() or b"").decode())


def run(cases, *, geometry=0, density_mode=0, domain=(0.0, 1.0, 0.0, 1.0), testing_dt=100.0, operator="BARO_ALL",
        psi_solver=(1, 1e-7, 2000, 1.0), chi_solver=(1, 1e-7, 2000, 1.0), legacy_int=True, chained=True, device=-1,
        first_case_id=0):
    """cases: list of dict(A, B, C (nz,nr) O-grid planes; Q, F (nz-1,nr-1) B cells; optional psi_bc, chi_bc (nz,nr)) --
    all with or all without each boundary field.  Returns a list (one per case) of
    dict(files={legacy file name: array}, scalars, relax=[(iterations, err_now)], lines=[(label, value, ratio)], record).
    The stage order itself is xee_legacy_run's (csrc/xee_legacy.hip); this function packs the arrays and unpacks the result."""
    L = _lib()
    M = len(cases)
    nz, nr = cases[0]["A"].shape
    stack = lambda key: np.ascontiguousarray(np.stack([np.ascontiguousarray(c[key], f32) for c in cases]))
    A, B, Cc, Q, F = (stack(k) for k in ("A", "B", "C", "Q", "F"))
    psi_bc = stack("psi_bc") if cases[0].get("psi_bc") is not None else None
    chi_bc = stack("chi_bc") if cases[0].get("chi_bc") is not None else None
    cfg = _Config()
    cfg.geometry, cfg.density_mode, cfg.testing_dt = int(geometry), int(density_mode), float(testing_dt)
    for k, v in enumerate(domain):
        cfg.domain[k] = float(v)
    cfg.op_mode = {"BAROTROPIC": 0, "BAROCLINIC": 1, "BARO_ALL": 2}[operator]
    cfg.psi_strategy, cfg.psi_r, cfg.psi_max_iter, cfg.psi_alpha = int(psi_solver[0]), float(psi_solver[1]), int(psi_solver[2]), float(psi_solver[3])
    cfg.chi_strategy, cfg.chi_r, cfg.chi_max_iter, cfg.chi_alpha = int(chi_solver[0]), float(chi_solver[1]), int(chi_solver[2]), float(chi_solver[3])
    cfg.legacy_int, cfg.chained, cfg.device, cfg.first_case_id = int(bool(legacy_int)), int(bool(chained)), int(device), int(first_case_id)
    for spec in (psi_solver, chi_solver):
        if spec[2] % 100:
            raise ValueError("max_iter must be a multiple of the check step (100), as the legacy loop assumes")
        if spec[0] not in (1, 2):
            raise ValueError("legacy strategies 3 and 4 (max norm over the whole array) are not supported")
    h = C.c_void_p()
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _chk(L.xee_legacy_run(C.byref(h), C.byref(cfg), nr, nz, M, A, B, Cc, Q, F, ptr(psi_bc), ptr(chi_bc)))
    try:
        n = C.c_int()
        _chk(L.xee_legacy_field_count(h, C.byref(n)))
        fields = []
        for k in range(n.value):
            name, n1, n2 = C.create_string_buffer(96), C.c_int(), C.c_int()
            _chk(L.xee_legacy_field_info(h, k, name, 96, C.byref(n1), C.byref(n2)))
            fields.append((name.value.decode(), n1.value, n2.value))
        _chk(L.xee_legacy_scalar_count(h, C.byref(n)))
        keys = []
        for k in range(n.value):
            key = C.create_string_buffer(96)
            _chk(L.xee_legacy_scalar_info(h, k, key, 96))
            keys.append(key.value.decode())
        prof = X.profiles(nr, nz, geometry, density_mode, domain)
        out = []
        for m in range(M):
            res = dict(files={}, scalars={}, relax=[], profiles=prof)
            for name, n1, n2 in fields:
                arr = np.zeros((n2, n1), f32)
                _chk(L.xee_legacy_field(h, m, name.encode(), arr))
                res["files"][name] = arr
            for key in keys:
                v = C.c_float()
                if L.xee_legacy_scalar(h, m, key.encode(), C.byref(v)) == 0:
                    res["scalars"][key] = f32(v.value)
            ns = C.c_int()
            _chk(L.xee_legacy_relax(h, m, C.byref(ns), None, None, 0))
            its, errs = np.zeros(ns.value, np.int32), np.zeros(ns.value, f32)
            _chk(L.xee_legacy_relax(h, m, C.byref(ns), its.ctypes.data_as(C.c_void_p), errs.ctypes.data_as(C.c_void_p), ns.value))
            res["relax"] = [(int(i), f32(e)) for i, e in zip(its, errs)]
            rec = np.zeros(1, X.RECORD_DTYPE)
            _chk(L.xee_legacy_record(h, m, rec.ctypes.data_as(C.c_void_p)))
            res["record"] = rec[0]
            buf = C.create_string_buffer(8192)
            _chk(L.xee_legacy_efficiency_text(h, m, 0.0, buf, 8192))
            res["efficiency_text"] = buf.value.decode()
            res["lines"] = efficiency_lines(res["efficiency_text"])
            out.append(res)
        return out
    finally:
        L.xee_legacy_destroy(h)


def efficiency_lines(text):
    """efficiency.txt as [(label, value, ratio | None) | ("#", comment)] in the program's order, without its first line
    (the elapsed time): parsed back from the text xee_legacy_efficiency_text wrote"""
    out = []
    for ln in text.splitlines()[1:]:
        if ln.startswith(" # "):
            out.append(("#", ln[3:]))
            continue
        label, _, rest = ln[1:].partition(" : ")
        parts = rest.split(" , ")
        out.append((label, f32(float(parts[0])), f32(float(parts[1])) if len(parts) > 1 else None))
    return out


def write_outputs(result, outdir, elapsed=None):
    """Write one case's files under the legacy names and its efficiency.txt (the text xee_legacy_efficiency_text made --
    ES16.9 numbers: an fp32 value round-trips; the reference's list-directed output prints 8 digits -- with the elapsed
    time filled in)."""
    os.makedirs(outdir, exist_ok=True)
    for name, arr in result["files"].items():
        np.ascontiguousarray(arr, f32).tofile(os.path.join(outdir, name))
    head, _, body = result["efficiency_text"].partition("\n")
    with open(os.path.join(outdir, "efficiency.txt"), "w") as fh:
        fh.write(head.partition(" : ")[0] + " : %16.9E\n" % (0.0 if elapsed is None else elapsed))
        fh.write(body)
    return os.path.join(outdir, "efficiency.txt")


def run_and_write(cases, outdirs, **kw):
    t0 = time.perf_counter()
    res = run(cases, **kw)
    dt = time.perf_counter() - t0
    return [write_outputs(r, d, elapsed=dt) for r, d in zip(res, outdirs)], res
