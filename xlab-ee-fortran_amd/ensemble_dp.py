"""Ensemble data-parallelism over the GPUs of one node: one process per GPU,
independent cases, a static block partition of the case list, and ONE collective -- an all-gather
of the fixed-size 128-byte case records (xee_case_record_t) over torch.distributed (backend "nccl"
= RCCL over xGMI on the GPU node; "gloo" in the CPU tests).  Fields (psi, eta) are never gathered;
each rank keeps / writes its own.

The gather is latency-bound (256 cases = 32 KiB), so there is nothing to overlap or bucket.
"""
import numpy as np

RECORD_BYTES = 128


def partition(n_cases, world, rank):
    """Static block partition: rank g owns [g*ceil(N/G), min(N, (g+1)*ceil(N/G)))."""
    per = -(-n_cases // world)
    lo = min(n_cases, rank * per)
    return lo, min(n_cases, lo + per)


def gather_records(records, n_cases, world, rank, device=None):
    """records: this rank's numpy structured array (RECORD_DTYPE, 128 B each, possibly fewer than
    ceil(N/G) on the last rank).  Returns all n_cases records ordered by case (every rank gets them).
    world == 1 needs no process group."""
    import torch
    per = -(-n_cases // world)
    dtype = records.dtype
    assert dtype.itemsize == RECORD_BYTES
    if world == 1:
        return records.copy()
    import torch.distributed as dist
    buf = np.zeros((per, RECORD_BYTES), np.uint8)              # pad the tail rank
    buf[: len(records)] = records.view(np.uint8).reshape(len(records), RECORD_BYTES)
    send = torch.from_numpy(buf)
    if device is not None:
        send = send.to(device)
    recv = torch.empty((world * per, RECORD_BYTES), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    out = recv.cpu().numpy().reshape(-1)[: n_cases * RECORD_BYTES].copy()
    return out.view(dtype)
