#!/usr/bin/env python3
"""bench.py -- diagnosed cases/s of the elliptic-solve + EE-diagnostic hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one pass of the hot path over this rank's batch of synthetic cases, inputs already
resident in HBM:   f = -B  ->  solver coefficients (K1)  ->  cal_coe (K2)  ->  solve_elliptic
(K3+K4, fixed-work protocol of SURVEY.md 8d: r1=1e-8, r2 off, alpha=1, max_iter=100,000, check
every 100)  ->  cal_eta (K5)  ->  EE scalars (K8)  ->  gather of the 128-byte case records over
RCCL when N>1.  Cases are independent, so ranks share nothing but that gather (weak scaling:
--members cases per GPU; 32 x 8 GPUs is the 256-member ensemble of BASELINE.json configs[2]).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the Jacobi sweep) with the
ALGORITHMIC byte count of SURVEY.md 8(d) -- 48 B per interior point per iteration -- over the device
time of the sweep launches, measured with HIP events on the stream they run on (inside libxee's
solve driver, xee_timing_t.sweep_ms).  `cpu_baseline` times the reference's own solve_elliptic
(oracle/_ref/libref_elliptic.so, built from the reference sources with amdflang -O2) -- or, if that
file is absent, the oracle's reference-loop-order port -- on ONE host core for a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_POINT_ITER = 48.0  # 9 coe + f + psi read, psi' written


def member_fields(nr, nz, m):
    """SURVEY.md 8d synthetic member m (float64 math, cast to float32)."""
    r = np.linspace(0.0, 1.0, nr)
    z = np.linspace(0.0, 1.0, nz)
    B = 1e-2 * np.sin(2.0 * np.pi * (1 + m % 4) * r)[None, :] * np.sin(np.pi * (1 + m % 5) * z)[:, None]
    if m > 0:
        B = B + 1e-4 * np.random.default_rng(20140911 + m).uniform(-1.0, 1.0, size=(nz, nr))
    one = np.ones((nz, nr), np.float32)
    rb = 0.5 * (r[1:] + r[:-1])[None, :]
    zb = 0.5 * (z[1:] + z[:-1])[:, None]
    Q = np.exp(-((rb - 0.3) / 0.1) ** 2 - ((zb - 0.5) / 0.2) ** 2) * (1.0 + 0.01 * (m % 7))
    return one, B.astype(np.float32), one, np.zeros((nz, nr), np.float32), Q.astype(np.float32)


def traffic_per_launch(kernel, nr, nz, members, T, operator="BAROTROPIC"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, corrections documented in profiles/r01/pmc/README.md); None unless an
    entry was measured for exactly this workload."""
    try:
        ent = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(kernel)
    except (OSError, ValueError):
        return None
    if ent and (ent["nr"], ent["nz"], ent["members"], ent["T"], ent.get("operator", "BAROTROPIC")) == (nr, nz, members, T, operator):
        return ent["bytes_per_launch"]
    return None


def cpu_baseline(nr, nz, max_iter, sample_iters):
    """Reference CPU path on one host core: solve_elliptic for `sample_iters` iterations of one case
    of the same workload, scaled linearly to max_iter (per-iteration cost is constant)."""
    import orchelp as H
    A, B, Cc, bc = H.test1_inputs(nr, nz)
    prof = H.profiles(nr, nz)
    sA, sB, sC = H.solver_coefficients(A, B, Cc, prof["rcuva"], prof["rho"])
    coe = H.cal_coe(sA, np.zeros_like(sB), sC, prof["dr"], prof["dz"], nr, nz)
    f = (-B).astype(np.float32)
    if H.have_ref():
        kind = "reference"
        t0 = time.perf_counter()
        H.ref_solve(bc, coe, f, sample_iters, 1e-8, -1.0, 1.0)
        dt = time.perf_counter() - t0
    else:
        kind = "port"
        t0 = time.perf_counter()
        H.solve(bc, coe, f, sample_iters, 1e-8, -1.0, 1.0, reforder=True)
        dt = time.perf_counter() - t0
    per_iter = dt / sample_iters
    return {"value": 1.0 / (per_iter * max_iter), "unit": "cases/s", "cores": 1, "kind": kind,
            "sample": f"solve_elliptic on one {nr}x{nz} case, {sample_iters} iterations in {dt:.2f} s "
                      f"({per_iter * 1e3:.2f} ms/iteration), scaled linearly to {max_iter} iterations",
            "ms_per_iteration": per_iter * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=32, help="cases per GPU (weak scaling)")
    ap.add_argument("--nr", type=int, default=1024)
    ap.add_argument("--nz", type=int, default=512)
    ap.add_argument("--max-iter", type=int, default=100000)
    ap.add_argument("--cpu-sample-iters", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="ensemble", choices=["ensemble", "single", "highres", "baroclinic"],
                    help="ensemble (default: the bench line) | single (configs[1]: one case) | highres (configs[3]: one "
                         "4096x2048 case, 2,000 iterations) | baroclinic (the ensemble with the full-B operator)")
    args = ap.parse_args()
    if args.workload == "single":
        args.members = 1
    elif args.workload == "highres":
        args.members, args.nr, args.nz, args.max_iter = 1, 4096, 2048, 2000
        args.cpu_sample_iters = min(args.cpu_sample_iters, 20)

    import torch
    import torch.distributed as dist
    import xee_pkg
    xee = xee_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (never what the driver runs):
    # XEE_BENCH_BACKEND=gloo XEE_BENCH_DEVICE=0 puts every rank on one card and gathers the records over gloo
    backend = os.environ.get("XEE_BENCH_BACKEND", "nccl")
    if "XEE_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["XEE_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available() or xee.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libxee has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    nr, nz, M = args.nr, args.nz, args.members
    prof = xee.profiles(nr, nz, 0, 0, (0.0, 1.0, 0.0, 1.0))
    ens = xee.Ensemble(nr, nz, M, device=local_rank)
    ens.set_profiles(prof)
    ens.set_case_ids(rank * M)
    for m in range(M):                         # inputs resident in HBM before the timed region
        A, B, Cc, bc, Q = member_fields(nr, nz, rank * M + m)
        for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
            ens.upload(name, arr, first=m)

    from xlab_ee_fortran_amd import ensemble_dp as dp

    def step():
        ens.forcing_from_B()                   # A7
        ens.assemble(xee.BAROCLINIC if args.workload == "baroclinic" else xee.BAROTROPIC)   # K1 + K2
        ens.solve(args.max_iter, 1e-8, -1.0, 1.0, check_step=100, converge_time=10, lost_rate=5)   # K3 + K4
        ens.cal_eta()                          # K5
        rec = ens.ee_scalars()                 # K8
        if world > 1:                          # the one collective of the path: scalar records over RCCL
            rec = dp.gather_records(rec, world * M, world, rank, device=coll_dev)
        return rec

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    sweep_ms = 0.0
    iters = 0
    launches = 0
    for _ in range(args.steps):
        rec = step()
        t = ens.timing()
        sweep_ms += t["sweep_ms"]
        iters += t["iterations"]
        launches += t["sweep_launches"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        cases = world * M * args.steps
        interior = (nr - 2) * (nz - 2)
        alg_bytes = BYTES_PER_POINT_ITER * interior * M * iters        # this rank, all timed steps
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        tb_T = int(os.environ.get("XEE_TB_T", "6"))
        kernel = ("k_sweep_tb" if os.environ.get("XEE_NO_BASE4") else
                  "k_sweep_tb4" if os.environ.get("XEE_NO_FASTQ") else "k_sweep_q") if tb_T > 1 else "k_sweep_stream"
        out = {
            "metric": "diagnosed cases/s (1024x512 r-z grid, SOR to 1e-8)",
            "value": cases / dt, "unit": "cases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"[{args.workload}] {M} independent {nr}x{nz} case(s) per GPU"
                                   + (" (x8 GPUs = the 256-member ensemble of configs[2])" if args.workload == "ensemble" else "")
                                   + f"; each case = configs[1]: K1+K2+solve_elliptic(r1=1e-8, r2 off, alpha=1, "
                                   f"max_iter={args.max_iter}, check every 100)+cal_eta+EE scalars; operator "
                                   + ("BAROCLINIC (full B)" if args.workload == "baroclinic" else "BAROTROPIC (B=0)"),
                       "nr": nr, "nz": nz, "members_per_gpu": M, "max_iter": args.max_iter,
                       "solver": "damped Jacobi, bit-identical to the reference (SURVEY.md finding 1)",
                       "parallelism": f"ensemble-dp{world}"},
            "iterations_per_s": world * M * iters / dt,
            "point_updates_per_s": world * M * iters * interior / dt,
            "all_cases_ran_to_max_iter": bool(len(rec) == world * M and np.all(rec["iters"] == args.max_iter)),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_per_launch(kernel, nr, nz, M, tb_T,
                                                                                "BAROCLINIC" if args.workload == "baroclinic" else "BAROTROPIC"),
                         "kernel": f"{kernel} (damped-Jacobi sweep, {tb_T} iteration(s) per launch; temporal blocking "
                                   "makes the effective rate exceed the HBM peak: traffic << algorithmic bytes)",
                         "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1),
                         "avg_launch_us": sweep_ms * 1e3 / max(launches, 1), "launches": launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            # in a child process: the reference prints "Elliptic Tools: ..." through the Fortran runtime's own
            # buffer, which is flushed at exit -- it must not land on this process's stdout (ONE JSON line)
            import subprocess
            code = ("import json, sys; sys.path.insert(0, %r); import bench; "
                    "r = bench.cpu_baseline(%d, %d, %d, %d); sys.stdout.flush(); "
                    "sys.stderr.write('XEE_CPU_BASELINE ' + json.dumps(r) + '\\n')"
                    % (str(ROOT), nr, nz, args.max_iter, args.cpu_sample_iters))
            p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            for line in p.stderr.splitlines():
                if line.startswith("XEE_CPU_BASELINE "):
                    out["cpu_baseline"] = json.loads(line[len("XEE_CPU_BASELINE "):])
            if "cpu_baseline" not in out:
                raise SystemExit("cpu_baseline child failed:\n" + p.stderr[-2000:])
        print(json.dumps(out), flush=True)
    ens.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
