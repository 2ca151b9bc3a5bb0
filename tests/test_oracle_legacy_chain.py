"""The oracle's restatement of the LEGACY program (tests/legacy_chain.py: source-term assembly A11, Stage I -> II -> III,
integral check, exchange conversion, efficiency.txt) pinned against the reference's own legacy binary
(tests/golden/legacy_64x48_*.npz, written by oracle/_ref/old-diagnose = src/old-diagnose compiled unmodified):

  * EVERY field file the program writes is reproduced BITWISE -- J-B, RHS_rpsi_thm/mom-O, rpsi_before-O, w_before-A,
    u_before-C, dtheta_dt-B, theta_after-B, RHS_rchi-O, all eta-/rchi-[..] pairs, rpsi/w/u_after, wtheta_JF_after, and
    (case B) the bndconv files with the program's INTEGER metric factors;
  * every solve's iteration count, and its final residue and every scalar of efficiency.txt to the 8 digits printed.

The program's undefined behaviour is inert on these cases (see tests/golden/make_golden_legacy.py); the one thing
emulated is the sheared Q / F the program really holds (legacy_chain.sheared)."""
import json

import numpy as np
import pytest

import legacy_chain as LC
import orchelp as H

NR, NZ = 64, 48


def load(name):
    g = np.load(H.GOLDEN / f"{name}.npz")
    return g, json.loads(str(g["meta"]))


def run_chain(g, meta, **kw):
    op = meta["mode"].split("-")[3]
    return LC.run(g["in_A"], g["in_B"], g["in_C"], LC.sheared(g["in_Q"], NR, NZ), LC.sheared(g["in_F"], NR, NZ),
                  domain=tuple(meta["domain"]), testing_dt=meta["testing_dt"], operator=op,
                  psi_solver=tuple(meta["psi_solver"]), chi_solver=tuple(meta["chi_solver"]),
                  psi_bc=g["in_psi_bc"] if "in_psi_bc" in g.files else None,
                  chi_bc=g["in_chi_bc"] if "in_chi_bc" in g.files else None, **kw)


def eff_values(text):
    out = []
    for line in text.splitlines():
        if ":" in line and not line.lstrip().startswith("#") and "Time elapsed" not in line:
            label, rest = line.split(":", 1)
            out.append((label.strip(), [float(x) for x in rest.split(",")]))
    return out


@pytest.mark.parametrize("name", ["legacy_64x48_tendency", "legacy_64x48_baroall_bc"])
def test_chain_reproduces_the_legacy_binary(name):
    g, meta = load(name)
    got = run_chain(g, meta)
    # 1. every file, bit for bit
    wrote = [k[4:] for k in g.files if k.startswith("out_")]
    assert len(wrote) >= 19
    for fn in wrote:
        want = g["out_" + fn]
        assert fn in got["files"], f"the chain does not produce {fn}"
        have = got["files"][fn]
        assert have.shape == want.shape, fn
        same = np.array_equal(have.view(np.uint32), want.view(np.uint32))
        assert same, (fn, int(np.sum(have.view(np.uint32) != want.view(np.uint32))), float(np.nanmax(np.abs(have - want))))
    # 2. "Relaxation uses N steps. Final residue is r" of every solve, in program order
    assert [it for it, _ in got["relax"]] == [it for it, _ in meta["relax"]]
    for (_, r), (_, want) in zip(got["relax"], meta["relax"]):
        assert float(r) == pytest.approx(float(want), rel=2e-7)
    # 3. efficiency.txt: same lines in the same order, values to the printed digits
    want_lines = eff_values(meta["efficiency_txt"])
    assert [lab for lab, _, _ in got["lines"]] == [lab for lab, _ in want_lines]
    for (lab, v, ratio), (_, nums) in zip(got["lines"], want_lines):
        assert float(v) == pytest.approx(nums[0], rel=2e-7, abs=0), lab
        if ratio is not None:
            assert float(ratio) == pytest.approx(nums[1], rel=4e-7, abs=0), lab


def test_real_metric_factors_differ_from_the_legacy_integers():
    """The documented fix (real r, dr, dz in the exchange conversion) is a different number on this grid: the pin above
    really exercises the truncation."""
    g, meta = load("legacy_64x48_baroall_bc")
    a = run_chain(g, meta, legacy_int=True)["scalars"]["sum_bndconv_0"]
    b = run_chain(g, meta, legacy_int=False)["scalars"]["sum_bndconv_0"]
    assert abs(float(a) - float(b)) > 1e-4 * abs(float(a))
