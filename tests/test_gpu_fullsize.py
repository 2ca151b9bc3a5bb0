"""Parity at BASELINE.json's full size (1024x512), against the REFERENCE PROGRAM's own outputs
(tests/golden/golden_hashes.json, made by tests/golden/make_golden_deep.py running oracle/_ref/diagnose, the
unmodified sources of src/diagnose + xtt-lib-fortran compiled with amdflang -O2):

  * the protocol of SURVEY.md 8(d) config 2 at its full depth, 100,000 iterations, both operator variants
    (src/diagnose/diagnose.f90:15-21 and :42-48) -- the work bench.py times;
  * a full-B (BAROCLINIC) anchor at 2,000 iterations;
  * the 32-member batch bench.py runs, noisy members (m > 0) included: members the reference was run on are
    compared with its md5s, every other member with its own solo run (600 iterations each).
"""
import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
BT = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC")
BC = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROCLINIC")


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _inputs(g):
    kind, m = g["inputs"]
    nr, nz = g["shape"]
    ins = H.test1_inputs(nr, nz) if kind == "test1" else H.member_inputs(nr, nz, m)
    for name, arr in zip(("A", "B", "C", "bc_init"), ins):
        assert H.md5(arr) == g["input_md5"][name], "the synthetic inputs are not the ones the reference was run on"
    return ins


@pytest.mark.parametrize("key", ["m1024x512_100000", "m1024x512_baroclinic_100000", "m1024x512_baroclinic_2000",
                                 "bench_m0_100000", "bench_m0_baroclinic_100000", "bench_m0_4096x2048_2000"])
def test_full_depth_anchor(key, golden_hashes):
    if key not in golden_hashes:
        pytest.skip(f"{key} not generated yet (tests/golden/make_golden_deep.py)")
    g = golden_hashes[key]
    A, B, Cc, bc = _inputs(g)
    mode = tuple(g["cfg"]["mode"])
    v = mode[3]
    iters = g["cfg"]["max_iter"]
    out = xee.diagnose_case(A, B, Cc, bc, mode=mode, r1=g["cfg"]["r1"], r2=g["cfg"]["r2"], max_iter=iters, alpha=g["cfg"]["alpha"])[v]
    assert int(out["iters"][0]) == iters == g["relax"][0][0] and int(out["err"][0]) == 1
    assert H.md5(out["psi"][0]) == g["md5"][f"rchi-[{v}]-O.bin"]
    assert H.md5(out["eta"][0]) == g["md5"][f"eta-[{v}]-A.bin"]
    # the "Relaxation uses N steps. Final residue is r1 , r2" line, to the digits the reference prints
    assert np.isclose(float(out["r1"][0]), float(g["relax"][0][1]), rtol=2e-7, atol=0)
    assert np.isclose(float(out["r2"][0]), float(g["relax"][0][2]), rtol=2e-7, atol=0)


@pytest.mark.parametrize("variant,M,ref_members", [("BAROTROPIC", 32, (1, 2, 3, 7)), ("BAROCLINIC", 16, (1, 6))])
def test_full_size_batch_members(variant, M, ref_members, golden_hashes):
    """The batch shape of the bench (32 x 1024x512, noisy members): each member of the batched solve against the
    reference's md5 where the reference was run on it, else against the same member solved alone."""
    mode = BT if variant == "BAROTROPIC" else BC
    ins = [H.member_inputs(1024, 512, m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    cfg = dict(mode=mode, r1=1e-8, r2=-1.0, max_iter=600, alpha=1.0)
    out = xee.diagnose_case(A, B, Cc, bc, **cfg)[variant]
    assert np.all(out["iters"] == 600)
    # (both batches stream: k_sweep_z<star> in two halves of 16, k_sweep_z<nine> for the 16 members that share AP and CQ)
    assert out["timing"]["z_launches"] == out["timing"]["sweep_launches"] > 0, out["timing"]
    checked = 0
    for m in range(M):
        key = f"member{m}_1024x512_600" + ("_baroclinic" if variant == "BAROCLINIC" else "")
        if m in ref_members and key in golden_hashes:
            g = golden_hashes[key]
            assert H.md5(B[m]) == g["input_md5"]["B"]
            assert H.md5(out["psi"][m]) == g["md5"][f"rchi-[{variant}]-O.bin"], m
            assert H.md5(out["eta"][m]) == g["md5"][f"eta-[{variant}]-A.bin"], m
            checked += 1
        else:
            solo = xee.diagnose_case(A[m], B[m], Cc[m], bc[m], **cfg)[variant]
            assert biteq(out["psi"][m], solo["psi"][0]) and biteq(out["eta"][m], solo["eta"][0]), m
            assert out["r1"][m].tobytes() == solo["r1"][0].tobytes()
    assert checked == len(ref_members), "reference anchors for the noisy members are missing from golden_hashes.json"
    # one member against the oracle as well (ties the solo runs to the CPU restatement at this size)
    m = M - 1
    want = H.diagnose_case(A[m], B[m], Cc[m], bc[m], **cfg)[variant]
    assert biteq(out["psi"][m], want["psi"]) and out["r1"][m].tobytes() == want["r1"].tobytes()


def test_the_256_member_ensemble_of_configs2_in_one_batch(golden_hashes):
    """BASELINE.json configs[2] whole: 256 members x 1024x512 resident in ONE device ensemble (6.7 GB of planes), 200
    iterations.  Members the reference program was run on against its md5s, every other member against the same member
    solved alone."""
    M, nr, nz, iters = 256, 1024, 512, 200
    prof = xee.profiles(nr, nz)
    refs = [m for m in (1, 2, 3, 7, 255) if f"member{m}_1024x512_{iters}" in golden_hashes]
    assert len(refs) == 5, "reference anchors missing: tests/golden/make_golden_deep.py --only member1_1024x512_200,..."
    with xee.Ensemble(nr, nz, M) as big, xee.Ensemble(nr, nz, 1) as one:
        big.set_profiles(prof); one.set_profiles(prof)
        for m in range(M):
            A, B, Cc, bc = H.member_inputs(nr, nz, m)
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                big.upload(name, arr, first=m)
        big.forcing_from_B(); big.assemble(xee.BAROTROPIC)
        rec = big.solve(iters, 1e-8, -1.0, 1.0)
        big.cal_eta()
        assert np.all(rec["iters"] == iters) and big.sweep_info()["kernel"] == "k_sweep_z"      # (every launch streamed, the residual-writing ones too)
        for m in range(M):
            psi, eta = big.download("PSI", m, 1)[0], big.download("ETA", m, 1)[0]
            if m in refs:
                g = golden_hashes[f"member{m}_1024x512_{iters}"]
                assert H.md5(psi) == g["md5"]["rchi-[BAROTROPIC]-O.bin"], m
                assert H.md5(eta) == g["md5"]["eta-[BAROTROPIC]-A.bin"], m
                assert np.isclose(float(rec["err_now"][m]), float(g["relax"][0][1]), rtol=2e-7, atol=0)
            else:
                A, B, Cc, bc = H.member_inputs(nr, nz, m)
                for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
                    one.upload(name, arr)
                one.forcing_from_B(); one.assemble(xee.BAROTROPIC)
                r1 = one.solve(iters, 1e-8, -1.0, 1.0)
                one.cal_eta()
                assert biteq(psi, one.download("PSI")[0]) and biteq(eta, one.download("ETA")[0]), m
                assert rec["err_now"][m].tobytes() == r1["err_now"][0].tobytes(), m
