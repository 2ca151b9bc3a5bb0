"""Edge sizes through the whole GPU path (blocked kernel tiling, exact norm chunking, ragged pitch):
the smallest legal grids, grids smaller than one tile, exact tile multiples +-1, long thin and tall
thin domains, and max_iter around the check / launch boundaries."""
import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


SIZES = [(3, 3), (4, 3), (3, 7), (5, 4), (9, 65), (65, 9), (64, 64), (63, 65), (66, 53), (53, 54), (105, 53),
         (106, 106), (129, 40), (257, 5), (6, 513), (1030, 7)]


@pytest.mark.parametrize("nr,nz", SIZES)
@pytest.mark.parametrize("variant", ["BAROTROPIC", "BAROCLINIC"])
def test_edge_sizes_whole_case(nr, nz, variant):
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=nr * 7 + nz)
    cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", variant),
               domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=213, alpha=0.9)
    with np.errstate(all="ignore"):
        got = xee.diagnose_case(A, B, Cc, bc, **cfg)[variant]
        want = H.diagnose_case(A, B, Cc, bc, **cfg)[variant]
    assert int(got["iters"][0]) == want["iters"] and int(got["err"][0]) == want["err"]
    assert biteq(got["psi"][0], want["psi"]), (nr, nz)
    assert biteq(got["eta"][0], want["eta"])
    assert got["r1"][0].tobytes() == want["r1"].tobytes() and got["r2"][0].tobytes() == want["r2"].tobytes()


@pytest.mark.parametrize("max_iter", [0, 1, 5, 6, 7, 99, 100, 101, 106, 200, 206])
def test_iteration_count_boundaries(max_iter):
    nr, nz = 70, 45
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=99)
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, sB, sC, prof["dr"], prof["dz"], nr, nz)
    f = (-B).astype(np.float32)
    want = H.solve(bc, coe, f, max_iter, 1e-8, -1.0, 1.0)
    got = xee.solve_elliptic(bc, coe, f, max_iter, 1e-8, -1.0, 1.0)
    assert biteq(got[0], want[0])
    if max_iter > 0:
        assert got[1] == want[1] and got[2] == want[2]
        assert got[3].tobytes() == want[3].tobytes() and got[4].tobytes() == want[4].tobytes()


@pytest.mark.parametrize("tile,T", [(808, 3), (808, 11), (806, 8), (806, 13), (804, 2), (804, 9), (804, 16)])
@pytest.mark.parametrize("variant", ["BAROTROPIC", "BAROCLINIC"])
@pytest.mark.parametrize("alpha", [1.0, 0.8])
def test_sweep_options_members_and_short_check_steps(tile, T, variant, alpha):
    """Every tile shape of k_sweep_q x launch lengths x both operators x the alpha == 1 instance and the general one, on a
    three-member batch with a check every 7 iterations (so residual-writing launches of every length occur, members stop at
    different checks and the snapshot the residual-writing launch writes is what a stopped member's result is taken from)."""
    nr, nz, M = 131, 77, 3
    ins = [H.rich_inputs(nr, nz, seed=300 + m) for m in range(M)]
    A, B, Cc, bc = (np.stack([x[k] for x in ins]) for k in range(4))
    prof = H.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4))
    with xee.Ensemble(nr, nz, M) as e:
        e.set_option("tile", tile); e.set_option("tb_T", T)
        e.set_profiles(xee.profiles(nr, nz, 0, 0, (0.0, 2.0e5, 0.0, 1.5e4)))
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc)):
            e.upload(name, arr)
        e.forcing_from_B()
        e.assemble(xee.BAROCLINIC if variant == "BAROCLINIC" else xee.BAROTROPIC)
        rec = e.solve(3000, 1e-7, 4e-2, alpha, check_step=7, converge_time=3, lost_rate=2)
        psi = e.download("PSI")
        info = e.sweep_info()
    assert info["kernel"] == "k_sweep_q"
    iters = set()
    for m in range(M):
        sA, sB, sC = H.solver_coefficients(A[m], B[m], Cc[m], prof["rcuva"], prof["rho"])
        coe = H.cal_coe(sA, sB if variant == "BAROCLINIC" else np.zeros_like(sB), sC, prof["dr"], prof["dz"], nr, nz)
        want = H.solve(bc[m], coe, (-B[m]).astype(np.float32), 3000, 1e-7, 4e-2, alpha, check_step=7, converge_time=3, lost_rate=2)
        assert int(rec["iters"][m]) == want[1] and int(rec["err_bits"][m]) == want[2], (m, rec["iters"][m], want[1])
        assert biteq(psi[m], want[0]), m
        assert rec["err_now"][m].tobytes() == want[3].tobytes() and rec["ratio"][m].tobytes() == want[4].tobytes()
        iters.add(want[1])
    assert len(iters) > 1 or 3000 in iters
