"""Input families that drive the blocked sweep's optimistic body into (or around) its validation limits:
zero residuals and signed zeros, iterates that explode to Inf/NaN, coefficient magnitudes far outside the range
the hoisted-reciprocal update admits, negative damping.  Shared by the oracle-vs-reference pin (CPU) and the
GPU-vs-oracle parity test."""
import hashlib

import numpy as np

import orchelp as H

KINDS = ["tropic", "clinic", "tropic-zero", "clinic-zero", "tropic-explode", "clinic-explode",
         "tropic-tiny", "tropic-huge", "tropic-negalpha", "tropic-converged", "clinic-converged", "tropic-converged-negzero"]
BIG_KINDS = ["tropic-big", "clinic-big"]


def build(kind):
    """-> dict(A, B, Cc, bc, forcing, cfg) for xee.diagnose_case / H.diagnose_case."""
    nr, nz = (1024, 512) if "big" in kind else (200, 150)
    variant = "BAROCLINIC" if "clinic" in kind else "BAROTROPIC"
    if "converged" in kind:               # a small smooth case iterated until whole regions are bitwise stationary: residuals
        nr, nz = 72, 56                   # that are EXACTLY zero next to a non-zero f (the zero-tolerant validation)
        A, B, Cc, bc = H.member_inputs(nr, nz, 2, noise=False)
        if "negzero" in kind:             # ... and a first guess holding -0 in the interior, which forbids that tolerance
            bc = bc.copy(); bc[10:30, 20:50] = -0.0
        cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", variant), domain=(0.0, 1.0, 0.0, 1.0),
                   r1=1e-12, r2=0.0, max_iter=60000, alpha=1.0)
        return dict(A=A, B=B, Cc=Cc, bc=bc, forcing=None, cfg=cfg, variant=variant)
    A, B, Cc, bc = H.rich_inputs(nr, nz, seed=3)
    if "tropic" in kind:
        B = np.zeros_like(B)
    forcing = None
    alpha, max_iter = 1.0, (1300 if "big" in kind else 700)
    if "zero" in kind:                    # psi = 0, forcing = 0 over most of the plane: zero residuals, signed zeros
        bc = np.zeros_like(bc)
        bc[:, :3] = -0.0
        forcing = np.zeros((nz, nr), np.float32)
        forcing[60:70, 80:120] = 1e-9
    if "explode" in kind:                 # over-relaxed: the iterate blows up to inf/NaN
        alpha, max_iter = 2.6, 900
    if "tiny" in kind:                    # coefficients scaled far outside the admitted range of |coe5|
        A, Cc, B = (x * np.float32(2.0 ** -90) for x in (A, Cc, B))
    if "huge" in kind:
        A, Cc, B = (x * np.float32(2.0 ** 60) for x in (A, Cc, B))
    if "negalpha" in kind:
        alpha, max_iter = -0.05, 300
    mode = ("SECONDARY_CIRCULATION" if forcing is not None else "DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", variant)
    cfg = dict(mode=mode, domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=0.0, max_iter=max_iter, alpha=alpha)
    return dict(A=A, B=B, Cc=Cc, bc=bc, forcing=forcing, cfg=cfg, variant=variant)


def canon(v):
    """the sign/payload of a NaN is not defined by IEEE 754 (nor by the reference): canonicalise before comparing"""
    v = np.ascontiguousarray(v)
    if v.dtype == np.float32:
        v = np.where(np.isnan(v), np.float32(np.nan), v).astype(np.float32)
    return v


def digest(res, keys=("psi", "iters", "err", "r1", "r2")):
    """md5 over the named results of one operator variant (GPU dicts hold a leading member axis of 1)."""
    h = hashlib.md5()
    for k in keys:
        v = np.asarray(res[k])
        if k in ("psi", "eta", "u", "w") and v.ndim == 3:
            v = v[0]
        v = canon(np.asarray(v).reshape(-1) if v.ndim <= 1 else v)
        if k in ("iters", "err"):
            v = np.asarray(v, np.int64).reshape(-1)
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()
