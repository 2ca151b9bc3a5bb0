"""xlab-ee-fortran_amd -- MI355X-native elliptic streamfunction solve + energy-efficiency diagnostic.

This directory holds only what the hot path needs:
  csrc/      hand-written HIP kernels for gfx950 and the C-ABI (include/xee.h)  -> lib/libxee.so
  fortran/   the Fortran host side: drop-in module `elliptic_tools` (ISO_C_BINDING -> libxee) and the
             `xee-diagnose` program that keeps the reference's stdin config and file formats
  xee.py     thin ctypes view of the C-ABI used by tests/bench (plumbing: numpy <-> C pointers)

The directory name is not a Python identifier; load it with the repo-root helper `xee_pkg.load()`
(importlib), which registers it as the module `xlab_ee_fortran_amd`.
"""
from .xee import *  # noqa: F401,F403  (xee.py defines __all__)
