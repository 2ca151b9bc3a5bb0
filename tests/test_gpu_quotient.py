"""The blocked sweep replaces the IEEE division of the update (elliptic_tools.f90:238) by the loop-invariant
reciprocal refinement plus five operations per iteration (csrc/xee_sweepq.hip).  These tests check, on the GPU,
that (1) the identity holds bit for bit over the whole operand range the kernel admits, and (2) the kernel with
the optimistic body gives the same bits as the plain IEEE-division kernel, including the cases its validation
must reject (zero residuals, exploding iterates, exotic coefficient magnitudes)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("d_lo,d_hi", [(-120, 119), (-50, -20), (-3, 3)])
def test_quotient_identity(d_lo, d_hi):
    # 4096 blocks x 256 threads x 256 pairs = 2.7e8 pairs per range (8e8 in all)
    bad, first = xee.quotient_selfcheck(seed=20260101 + d_lo, blocks=4096, per_thread=256, d_exp_lo=d_lo, d_exp_hi=d_hi)
    assert bad == 0, f"{bad} mismatches, e.g. a={first[0]!r} d={first[1]!r}"


def _run(env_extra, code):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


_CASE = r"""
import sys, hashlib
sys.path.insert(0, 'tests')
import numpy as np, orchelp as H, xee_pkg
xee = xee_pkg.load()
kind = sys.argv[1] if len(sys.argv) > 1 else ''
"""


def _digest(kind, env_extra):
    """psi/residual/record digest of one case family, computed in a child process (the kernel choice is read
    from the environment once per process)."""
    code = _CASE + f"kind = {kind!r}\n" + r"""
rng = np.random.default_rng(11)
nr, nz = (1024, 512) if 'big' in kind else (200, 150)
variant = 'BAROCLINIC' if 'clinic' in kind else 'BAROTROPIC'
A, B, Cc, bc = H.rich_inputs(nr, nz, seed=3)
if 'tropic' in kind:
    B = np.zeros_like(B)
forcing = None
alpha, max_iter = 1.0, (1300 if 'big' in kind else 700)
if 'zero' in kind:                    # psi = 0, forcing = 0 over most of the plane: zero residuals, signed zeros
    bc = np.zeros_like(bc); bc[:, :3] = -0.0
    forcing = np.zeros((nz, nr), np.float32); forcing[60:70, 80:120] = 1e-9
if 'explode' in kind:                 # over-relaxed: the iterate blows up to inf/NaN
    alpha, max_iter = 2.6, 900
if 'tiny' in kind:                    # coefficients scaled far outside the admitted range of |coe5|
    A = A * np.float32(2.0 ** -90); Cc = Cc * np.float32(2.0 ** -90); B = B * np.float32(2.0 ** -90)
if 'huge' in kind:
    A = A * np.float32(2.0 ** 60); Cc = Cc * np.float32(2.0 ** 60); B = B * np.float32(2.0 ** 60)
if 'negalpha' in kind:
    alpha, max_iter = -0.05, 300
out = xee.diagnose_case(A, B, Cc, bc, forcing=forcing, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", variant),
                        domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=0.0, max_iter=max_iter, alpha=alpha, flags=0)
h = hashlib.md5()
res = out[variant]
for k in ('psi', 'eta', 'iters', 'err', 'r1', 'r2'):
    v = np.ascontiguousarray(res[k])
    if v.dtype == np.float32:         # the sign/payload of a NaN is not defined by IEEE 754 (nor by the reference): canonicalise
        v = np.where(np.isnan(v), np.float32(np.nan), v).astype(np.float32)
    h.update(k.encode()); h.update(v.tobytes())
print('DIGEST', h.hexdigest(), res['iters'].tolist(), res['err'].tolist(), [str(x) for x in res['r1'].tolist()])
"""
    out = _run(env_extra, code)
    line = [ln for ln in out.splitlines() if ln.startswith("DIGEST")]
    assert line, out[-2000:]
    return line[0]


@pytest.mark.parametrize("kind", ["tropic", "clinic", "tropic-zero", "clinic-zero", "tropic-explode", "clinic-explode",
                                  "tropic-tiny", "tropic-huge", "tropic-negalpha", "tropic-big", "clinic-big"])
def test_fast_body_matches_exact_kernel(kind):
    fast = _digest(kind, {})
    exact = _digest(kind, {"XEE_NO_FASTQ": "1"})
    forced = _digest(kind, {"XEE_Q_EXACT_ONLY": "1"})
    assert fast == exact == forced


@pytest.mark.parametrize("rows", ["4", "6", "8"])
@pytest.mark.parametrize("kind", ["tropic", "tropic-zero", "tropic-explode", "tropic-tiny"])
def test_slim_star_kernel_matches_exact_kernel(kind, rows):
    """The opt-in slim form of the star body (weights shared between neighbours, f and y1 in LDS, psi updated in
    place, its own exact redo): same bits as the plain IEEE-division kernel."""
    assert _digest(kind, {"XEE_SLIM_ROWS": rows}) == _digest(kind, {"XEE_NO_FASTQ": "1"})
