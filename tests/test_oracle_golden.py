"""The oracle (oracle/xee_oracle.c) pinned against the reference's own outputs.

Golden .npz files hold inputs and the files the reference program wrote (tests/golden/make_golden.py
ran oracle/_ref/diagnose, i.e. the unmodified reference compiled with amdflang -O2).  Everything is
compared BIT FOR BIT (fp32 fields) -- that is what the 1e-10 / 1e-12 tolerances of the north star
mean for real(4) data (finding 2).
"""
import numpy as np
import pytest

import orchelp as H

CASES = ["p128x64_test1", "s64x48_baroall", "s64x48_seccirc", "s64x48_converge", "s64x48_spherical"]


def biteq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def run_oracle_case(g, rec, **kw):
    cfg = rec["cfg"]
    return H.diagnose_case(g["in_A"], g["in_B"], g["in_C"], g["in_bc"],
                           g["in_forcing"] if "in_forcing" in g.files else None,
                           mode=tuple(cfg.get("mode", ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"))),
                           domain=tuple(cfg.get("domain", (0.0, 1.0, 0.0, 1.0))),
                           r1=cfg["r1"], r2=cfg["r2"], max_iter=cfg["max_iter"], alpha=cfg["alpha"], **kw)


def check_against_golden(out, g, rec):
    sA, sB, sC = out["solver"]
    assert biteq(sA, g["out_solver_a-sA.bin"])
    assert biteq(sB, g["out_solver_b-B.bin"])
    assert biteq(sC, g["out_solver_c-sC.bin"])
    variants = [v for v in ("BAROTROPIC", "BAROCLINIC") if v in out]
    assert len(variants) == len(rec["relax"])
    for v, (iters, r1s, r2s) in zip(variants, rec["relax"]):
        res = out[v]
        assert res["iters"] == iters
        # the program prints 7-8 significant digits: pins the scalars to ~1 ulp
        assert float(res["r1"]) == pytest.approx(float(r1s), rel=2e-7)
        assert float(res["r2"]) == pytest.approx(float(r2s), rel=2e-7)
        for key, fn in (("eta", f"eta-[{v}]-A.bin"), ("psi", f"rchi-[{v}]-O.bin"), ("psi", f"rpsi-[{v}]-O.bin"),
                        ("u", f"u-[{v}]-C.bin"), ("w", f"w-[{v}]-A.bin")):
            if "out_" + fn in g.files:
                assert biteq(res[key], g["out_" + fn]), (v, fn)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_outputs(name, golden_hashes):
    g = np.load(H.GOLDEN / f"{name}.npz")
    rec = golden_hashes[name]
    check_against_golden(run_oracle_case(g, rec), g, rec)


def test_reference_order_mode_is_bit_identical(golden_hashes):
    """The i-outer/j-inner 'port' walk used for CPU timing gives the same bits."""
    g = np.load(H.GOLDEN / "s64x48_baroall.npz")
    rec = golden_hashes["s64x48_baroall"]
    check_against_golden(run_oracle_case(g, rec, reforder=True), g, rec)


def test_test1_fixture_md5(golden_hashes):
    """The shipped fixture: 200x200, 17,900 iterations, md5 of rchi/eta as in BASELINE.md section 4."""
    A, B, Cc, bc = H.test1_inputs(200, 200)
    assert H.md5(B) == "1add572fa31053a2c4c057f94030c8de"      # test/test1/B.bin
    out = H.diagnose_case(A, B, Cc, bc, r1=5e-3, r2=5e-3, max_iter=100000, alpha=1.0)["BAROTROPIC"]
    assert out["iters"] == 17900 and out["err"] == 0
    assert H.md5(out["psi"]) == "3495ddd52bcb05c39a5c6685f8f1f039"
    assert H.md5(out["eta"]) == "a671b036d5e6f27fa1f9893076ae6e8a"
    rec = golden_hashes["t200x200_test1"]
    assert H.md5(out["psi"]) == rec["md5"]["rchi-[BAROTROPIC]-O.bin"]


def test_baroclinic_300_md5(golden_hashes):
    A, B, Cc, bc = H.test1_inputs(200, 200)
    out = H.diagnose_case(A, B, Cc, bc, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC"),
                          r1=1e-8, r2=-1.0, max_iter=300, alpha=0.9)["BAROCLINIC"]
    assert H.md5(out["psi"]) == "d652e646617a0f156b7bcf0757b3c6fd"
    assert H.md5(out["eta"]) == "87384dc2f2331c91b90a90bdc2e341d7"


def test_m_grid_500_md5(golden_hashes):
    """1024x512, 500 fixed iterations (BASELINE.md section 4 anchor)."""
    A, B, Cc, bc = H.test1_inputs(1024, 512)
    assert H.md5(B) == "986b0fd66de089d7e32d3afc08bb53d0"
    out = H.diagnose_case(A, B, Cc, bc, r1=1e-8, r2=-1.0, max_iter=500, alpha=1.0)["BAROTROPIC"]
    assert out["iters"] == 500 and out["err"] == 1
    assert H.md5(out["psi"]) == "96ac915d5789a73fdd61a47447916235"
    assert H.md5(out["eta"]) == "826fefae4b9f730278ab0cfbefcf4df9"


def test_both_criteria_off_is_rejected():
    A, B, Cc, bc = H.rich_inputs(16, 12)
    coe = np.zeros((12, 16, 9), np.float32)
    *_, status = H.solve(bc, coe, B, 10, -1.0, 0.0, 1.0)
    assert status == 1


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (reference sources absent)")
def test_oracle_vs_reference_library_direct():
    """Direct call into the reference's own elliptic_tools (oracle/_ref/libref_elliptic.so):
    cal_coe, solve_elliptic incl. the in/out scalars, bit for bit."""
    for nr, nz, seed in ((37, 29, 3), (64, 48, 4)):
        A, B, Cc, bc = H.rich_inputs(nr, nz, seed)
        prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
        sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
        c1 = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
        c2 = H.ref_cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
        assert biteq(c1[1:-1, 1:-1], c2[1:-1, 1:-1])
        f = (-B).astype(np.float32)
        for r1, r2, mi, al in ((1e-8, -1.0, 300, 0.9), (5e-3, 5e-3, 5000, 1.0), (-1.0, 1e-3, 3000, 0.8), (1e-8, -1.0, 150, 1.0)):
            o = H.solve(bc, c1, f, mi, r1, r2, al)
            r = H.ref_solve(bc, c2, f, mi, r1, r2, al)
            assert biteq(o[0], r[0])
            assert o[1] == r[1] and o[2] == r[2]
            assert o[3].tobytes() == r[3].tobytes() and o[4].tobytes() == r[4].tobytes()
