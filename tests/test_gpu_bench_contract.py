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
    # the blocked sweep is VALU-issue bound: priced in flop/s against the unfused (packed) fp32 issue peak; the byte view
    # (algorithmic, compulsory and measured-traffic rates against the 8 TB/s HBM peak) sits beside it
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6 and r["peak_spec"] == 157.3
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    ts = r["traffic_source"]              # (no PMC entry exists for this toy size: nothing may be quoted)
    assert ts["entry_found"] is False and ts["matches_build"] is False and r["traffic"] is None and len(ts["this_build_hash"]) == 64
    h = r["hbm"]
    # (the bench's members share their operator -- A = C = 1 -- so AP and CQ count once per launch of 2 members, not per member:
    # 4 B psi + 4 B f in, 4 B psi out, and 2 x 4 B / 2)
    assert h["generating_planes_shared_by_all_members"] == 2 and h["algorithmic_bytes_per_point_iter"] == 12.0 + 8.0 / 2
    assert h["peak_GBps"] == 8000.0 and h["compulsory_GBps"] > 0
    assert h["real_frac_of_peak"] is None or 0 < h["real_frac_of_peak"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    ca = d["cpu_baseline_all_cores"]
    assert ca["cores"] >= 1 and ca["value"] >= 0.5 * c["value"] and ca["kind"] == c["kind"] and ca["host_nproc"] >= ca["cores"]
    assert d["all_cases_ran_to_max_iter"] is True and d["value"] > 0
    k = d["check"]
    assert len(k["member0_psi_md5"]) == 32 and len(k["member0_eta_md5"]) == 32 and k["matches_golden"] is None   # (no golden at 128x64)


def test_committed_counters_belong_to_this_build():
    """roofline.traffic / valu_busy_measured come from the rocprofv3 PMC passes committed in profiles/traffic.json; bench.py
    quotes them only when the sources they were measured on hash like the tree it runs from (tools/build_hash.py) -- and
    at HEAD they must: a kernel or driver change without a fresh PMC pass would otherwise publish a line without counters."""
    sys.path.insert(0, str(ROOT / "tools"))
    import build_hash
    tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    here = build_hash.build_hash(ROOT)
    assert tj.get("build_hash") == here, "re-run tools/profile_round.sh + tools/collect_profiles.py: the kernel changed since the PMC passes"
    assert tj["entries"] and all(e.get("build_hash") == here for e in tj["entries"])
    sys.path.insert(0, str(ROOT))
    import bench
    # (32 cases = two halves of 16 side by side, their plain launches streamed by k_sweep_z; the members share AP and CQ:
    # 16 x (psi in, f in, psi out) + AP + CQ = 104.2 MB compulsory per launch)
    ent = bench.traffic_entry("k_sweep_z", 1024, 512, 16, "BAROTROPIC")
    assert ent and 1.04e8 < ent["bytes_per_launch"] < 2.0e8 and ent["SQ_ACTIVE_INST_VALU"] > 0
    # (the BAROCLINIC batch streams too -- its members share AP and CQ --, four iterations per launch: 16 x (psi, f, BX, BY in, psi out)
    # + AP + CQ = 172 MB compulsory per launch)
    ent = bench.traffic_entry("k_sweep_z", 1024, 512, 16, "BAROCLINIC")
    assert ent and 1.7e8 < ent["bytes_per_launch"] < 2.6e8 and ent["SQ_ACTIVE_INST_VALU"] > 0


def test_bench_gpus2_as_a_plain_command_self_launches():
    """`python bench.py --gpus 2` with no launcher around it starts its own ranks as a child of torch.distributed.run
    (before touching the GPU) and relays the one JSON line; rehearsed on one card over gloo."""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(XEE_BENCH_BACKEND="gloo", XEE_BENCH_DEVICE="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--members", "2", "--nr", "128", "--nz", "64",
                        "--max-iter", "200", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["all_cases_ran_to_max_iter"] is True and d["config"]["parallelism"] == "ensemble-dp2"


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


def test_bench_sweep_workload_two_ranks_rehearsal():
    """bench.py --workload sweep (configs[4]) with two ranks on one card over gloo: ranks own blocks of timesteps, the case
    records of both arrive at rank 0, one JSON line with the per-stage figures"""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(XEE_BENCH_BACKEND="gloo", XEE_BENCH_DEVICE="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "sweep", "--members", "2", "--nr", "128",
                        "--nz", "64", "--max-iter", "300", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["cases_per_gpu"] == 12 and d["all_cases_ran_to_max_iter"] is True
    assert d["value"] > 0 and set(d["per_stage"]) == {"BAROTROPIC/all three kinds", "BAROCLINIC/all three kinds"}
    assert "gloo" in d["config"]["record_gather"]


def test_cabi_record_gather_on_one_rank():
    """xee_comm_* / xee_gather_scalars (RCCL behind the C-ABI) on the one GPU of this box: a world of one rank goes through
    ncclCommInitRank and ncclAllGather for real (RCCL refuses two ranks on one device, so the N > 1 exchange itself is
    covered by the gloo tests and exercised by the driver's multi-GPU run)."""
    import numpy as np
    sys.path.insert(0, str(ROOT))
    import xee_pkg
    xee = xee_pkg.load()
    uid = xee.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = xee.Comm(uid, 0, 1)
    rec = np.zeros(5, xee.RECORD_DTYPE)
    rec["case_id"] = np.arange(5) + 10
    rec["efficiency"] = np.float32(0.25) * np.arange(5, dtype=np.float32)
    rec["scalars"][:, 3] = 7.0
    out = comm.gather(rec, 8)
    assert len(out) == 8 and out[:5].tobytes() == rec.tobytes() and np.all(out["case_id"][5:] == -1)
    comm.close()


def test_comm_create_gives_up_when_a_rank_never_arrives():
    """xee_comm_create is collective; a rank that never arrives must not hang the others for ever: rank 0 of a world of two,
    alone, gets XEE_ERR_TIMEOUT after xee_comm_set_timeout's seconds (in a child process: the helper thread stays inside RCCL)."""
    code = ("import os, sys, time; sys.path.insert(0, %r); import xee_pkg; xee = xee_pkg.load();\n"
            "ok, why = xee.comm_available(); assert ok, why\n"
            "print('LIB', xee.comm_library()); xee.comm_set_timeout(4.0); t0 = time.time()\n"
            "try:\n"
            "    xee.Comm(xee.comm_unique_id(), 0, 2)\n"
            "    print('JOINED?')\n"
            "except xee.XeeError as e:\n"
            "    print('STATUS', e.status, round(time.time() - t0)); sys.stdout.flush()\n"
            "os._exit(0)\n" % str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "STATUS 6" in p.stdout, p.stdout + p.stderr[-2000:]
    assert "librccl" in p.stdout
