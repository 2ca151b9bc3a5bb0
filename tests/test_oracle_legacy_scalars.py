"""The oracle's EE-integral restatements (A10) pinned against the reference's LEGACY program, the only
place the reference computes scalar energy-efficiency numbers (src/old-diagnose/diagnose.f90).

tests/golden/legacy_efficiency_64x48.json holds the `efficiency.txt` that oracle/_ref/old-diagnose wrote
for a 64x48 case (make_golden_legacy.py).  The program prints 8 significant digits, so this pins the
fp32 scalars to ~1 ulp; bit-level parity of the scalars is then carried by the oracle itself
(SURVEY.md 8c).  The legacy program reads its nr x nz Q record into an (nr-1, nz-1) array (undefined
behaviour, quirk 14): what it integrates is the sheared field reproduced below."""
import json

import numpy as np
import pytest

import orchelp as H

NR, NZ = 64, 48


def legacy_values():
    rec = json.loads((H.GOLDEN / "legacy_efficiency_64x48.json").read_text())
    vals = {}
    for line in rec["efficiency_txt"].splitlines():
        if ":" in line and not line.lstrip().startswith("#"):
            label, rest = line.split(":", 1)
            vals[label.strip()] = [float(x) for x in rest.split(",")]
    return vals


def sheared_Q():
    r = np.linspace(0, 1, NR)[None, :]
    z = np.linspace(0, 1, NZ)[:, None]
    qfile = np.exp(-((r - 0.3) / 0.1) ** 2 - ((z - 0.5) / 0.2) ** 2).astype(np.float32).reshape(-1)
    return np.ascontiguousarray(qfile[: (NR - 1) * (NZ - 1)].reshape(NZ - 1, NR - 1))


def test_sum_Q_matches_legacy_program():
    """integrate_weight_B / cal_sum_Q, old-diagnose/diagnose.f90:1029-1071."""
    prof = H.profiles(NR, NZ)
    got = float(H.integrate_weight_B(sheared_Q(), prof))
    assert got == pytest.approx(legacy_values()["sum Q"][0], rel=1.5e-7)
    # and the un-sheared field is a different number: the emulation is doing real work
    r = np.linspace(0, 1, NR)[None, :]
    z = np.linspace(0, 1, NZ)[:, None]
    plain = np.exp(-((r - 0.3) / 0.1) ** 2 - ((z - 0.5) / 0.2) ** 2).astype(np.float32)[: NZ - 1, : NR - 1]
    assert abs(float(H.integrate_weight_B(np.ascontiguousarray(plain), prof)) - got) > 1e-3 * got


def test_sum_Qeta_matches_legacy_program():
    """cal_sum_Qeta, old-diagnose/diagnose.f90:1073-1092, on the eta field the legacy program wrote for its
    L(B=0) = B0 solve; and that eta is the oracle's cal_eta of the rchi it wrote (bitwise)."""
    g = np.load(H.GOLDEN / "legacy_64x48_fields.npz")
    prof = H.profiles(NR, NZ)
    eta = np.ascontiguousarray(g["eta_0_B0"])
    assert np.array_equal(H.cal_eta(np.ascontiguousarray(g["rchi_0_B0"]), prof).view(np.uint32), eta.view(np.uint32))
    vals = legacy_values()["eta [L(B=0)    = B0]     wo/ boundary"]
    sqe = float(H.cal_sum_Qeta(sheared_Q(), eta, prof))
    assert sqe == pytest.approx(vals[0], rel=1.5e-7)
    sq = H.integrate_weight_B(sheared_Q(), prof)
    assert float(np.float32(sqe) / sq) == pytest.approx(vals[1], rel=3e-7)
