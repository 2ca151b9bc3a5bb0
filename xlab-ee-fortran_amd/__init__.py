"""xlab-ee-fortran_amd -- MI355X-native elliptic streamfunction solve + energy-efficiency diagnostic.

This directory holds only what the hot path needs:
  csrc/      hand-written HIP kernels for gfx950 and the C-ABI (include/xee.h)  -> lib/libxee.so
  fortran/   the Fortran host side: drop-in module `elliptic_tools` (ISO_C_BINDING -> libxee) and the
             `xee-diagnose` program that keeps the reference's stdin config and file formats
  xee.py     thin ctypes view of the C-ABI used by tests/bench (plumbing: numpy <-> C pointers)

The directory name is not a Python identifier; load it with the repo-root helper `xee_pkg.load()`
(importlib), which registers it as the module `xlab_ee_fortran_amd`.
"""
from .xee import (  # noqa: F401
    Ensemble, XeeError, lib, lib_path, device_count, profiles, cal_coe, do_elliptic, solve_elliptic,
    solver_coefficients, residual_norm, cal_eta, cal_uw, integrate_weight_B, cal_sum_Qeta, cal_wtheta, rhs_thermal,
    rhs_from_B, diagnose_case, CaseRecord, FIELDS, BAROTROPIC, BAROCLINIC, quotient_selfcheck, m2_from_C, theta_tendency, relative_theta, theta_anomaly_update,
)
