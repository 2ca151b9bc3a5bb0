"""bench.py keeps its contract: exactly ONE JSON line on stdout with the required keys, incl. the
`roofline` and `cpu_baseline` objects (run here on a tiny workload so that it takes seconds)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def test_bench_prints_one_json_line_with_required_keys():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--members", "2", "--nr", "128", "--nz", "64",
                        "--max-iter", "300", "--steps", "1", "--warmup", "0", "--cpu-sample-iters", "5"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "cases/s" and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["all_cases_ran_to_max_iter"] is True and d["value"] > 0


def test_bench_two_ranks_rehearsal_over_gloo():
    """The multi-rank path of bench.py (one process per GPU, barrier + max over ranks, record gather) rehearsed with two
    ranks on ONE card: XEE_BENCH_BACKEND=gloo / XEE_BENCH_DEVICE=0 replace RCCL and the per-rank device, nothing else."""
    import os
    port = 29600 + os.getpid() % 300
    env = dict(os.environ, XEE_BENCH_BACKEND="gloo", XEE_BENCH_DEVICE="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2",
                        "--members", "3", "--nr", "128", "--nz", "64", "--max-iter", "300", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["members_per_gpu"] == 3 and d["all_cases_ran_to_max_iter"] is True
    assert d["value"] > 0 and d["scaling"] == "weak" and d["config"]["parallelism"] == "ensemble-dp2"
