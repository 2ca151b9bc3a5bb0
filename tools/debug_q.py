"""developer aid: compare the fast-quotient kernel with the IEEE-division kernel on one case, iteration count by iteration count"""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import sys
sys.path.insert(0, 'tests')
import numpy as np, orchelp as H, xee_pkg
xee = xee_pkg.load()
nr, nz = 200, 150
A, B, Cc, bc = H.rich_inputs(nr, nz, seed=3)
B = np.zeros_like(B)
for it in [int(x) for x in sys.argv[2].split(',')]:
    out = xee.diagnose_case(A, B, Cc, bc, mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC"),
                            domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=0.0, max_iter=it, alpha=2.6, flags=0)
    np.save(f"{sys.argv[1]}_{it}.npy", out["BAROTROPIC"]["psi"])
"""
its = sys.argv[1] if len(sys.argv) > 1 else "6,12,18,24,30,36,42,48,60,72,84,96,100,106,150,200,300"
for tag, env in (("fast", {}), ("exact", {"XEE_NO_FASTQ": "1"})):
    subprocess.run([sys.executable, "-c", CHILD, f"/tmp/dq_{tag}", its], cwd=ROOT, env=dict(os.environ, **env), check=True)
for it in its.split(","):
    a = np.load(f"/tmp/dq_fast_{it}.npy").view(np.uint32); b = np.load(f"/tmp/dq_exact_{it}.npy").view(np.uint32)
    d = np.argwhere(a != b)
    fa = a.view(np.float32); fb = b.view(np.float32)
    msg = f"it {it}: {len(d)} differing words; finite {np.isfinite(fb).mean():.3f}"
    if len(d):
        i = tuple(d[0]); msg += f" first at {i}: fast {fa[i]!r} ({a[i]:08x}) exact {fb[i]!r} ({b[i]:08x}); max|exact| {np.nanmax(np.abs(fb[np.isfinite(fb)])) if np.isfinite(fb).any() else 'n/a'}"
        ys = sorted(set(int(x[1]) for x in d))[:10]; xs = sorted(set(int(x[2]) for x in d))[:10]
        msg += f" rows {ys} cols {xs}"
    print(msg, flush=True)
