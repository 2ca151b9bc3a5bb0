"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/xee.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import xee_pkg

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "xee.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xee_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_declares_the_reference_interface():
    syms = declared_symbols()
    for must in ("xee_cal_coe", "xee_do_elliptic", "xee_solve_elliptic", "xee_judge_error",
                 "xee_solver_coefficients", "xee_cal_eta", "xee_cal_uw", "xee_integrate_weight_B",
                 "xee_cal_sum_Qeta", "xee_ensemble_create", "xee_ensemble_solve", "xee_ensemble_records"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    xee = xee_pkg.load()
    L = C.CDLL(str(xee.lib_path()))
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    assert L.xee_version() >= 100


def test_record_is_128_bytes():
    xee = xee_pkg.load()
    from xlab_ee_fortran_amd import xee as x
    assert C.sizeof(x.CaseRecord) == 128 and x.RECORD_DTYPE.itemsize == 128


def test_no_cpu_fallback_without_gpu():
    """Without a device every compute entry fails loudly with XEE_ERR_NO_DEVICE."""
    xee = xee_pkg.load()
    if xee.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(xee.XeeError) as ei:
        xee.Ensemble(16, 16, 1)
    assert ei.value.status == 3
    with pytest.raises(xee.XeeError):
        xee.cal_coe(np.zeros((14, 15), np.float32), np.zeros((15, 15), np.float32),
                    np.zeros((15, 14), np.float32), 1.0, 1.0, 16, 16)


def test_host_profiles_match_oracle_and_golden():
    """xee_profiles is host code inside libxee (glibc powf/cosf): no GPU needed.  The golden solver
    dumps were produced by the reference from exactly these profiles, so bitwise agreement with
    the oracle's profiles (which reproduce those dumps) pins them."""
    import orchelp as H
    xee = xee_pkg.load()
    for geom, dens, dom in ((0, 0, (0.0, 2.0e5, 0.0, 1.5e4)), (0, 1, (0.0, 1.0, 0.0, 1.0)),
                            (1, 0, (6371000.0, 0.0, 1.5e4))):
        a = xee.profiles(64, 48, geom, dens, dom)
        b = H.profiles(64, 48, geom, dens, dom)
        for k in ("ra", "za", "rcuva", "exner", "rho"):
            assert a[k].tobytes() == b[k].tobytes(), (geom, dens, k)
        assert a["dr"].tobytes() == b["dr"].tobytes() and a["dz"].tobytes() == b["dz"].tobytes()
