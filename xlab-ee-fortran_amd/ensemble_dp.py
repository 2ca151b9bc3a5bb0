"""Ensemble data-parallelism over the GPUs of one node: one process per GPU,
independent cases, a static block partition of the case list, and ONE collective -- an all-gather
of the fixed-size 128-byte case records (xee_case_record_t).  On GPUs the gather is libxee's own
xee_gather_scalars (RCCL ncclAllGather behind the C-ABI, pass `comm`); without a communicator it
falls to torch.distributed ("gloo" in the CPU tests).  Fields (psi, eta) are never gathered; each
rank keeps / writes its own.

The gather is latency-bound (256 cases = 32 KiB), so there is nothing to overlap or bucket.
"""
import numpy as np

RECORD_BYTES = 128


def partition(n_cases, world, rank):
    """Static block partition: rank g owns [g*ceil(N/G), min(N, (g+1)*ceil(N/G)))."""
    per = -(-n_cases // world)
    lo = min(n_cases, rank * per)
    return lo, min(n_cases, lo + per)


def make_comm(xee, world, rank, device, broadcast_bytes):
    """The C-ABI communicator: rank 0 makes the rendezvous id, `broadcast_bytes(b or None) -> bytes` carries it to the
    other ranks over whatever channel the host has (bench.py: a torch.distributed broadcast)."""
    uid = broadcast_bytes(xee.comm_unique_id() if rank == 0 else None)
    return xee.Comm(uid, rank, world, device)


def gather_records(records, n_cases, world, rank, device=None, per_rank=None, comm=None):
    """records: this rank's numpy structured array (RECORD_DTYPE, 128 B each, case_id >= 0), in
    ascending case order; ranks hold consecutive blocks.  Every rank gets all n_cases records ordered
    by rank block.  `per_rank` = the largest number of records any rank holds (default
    ceil(n_cases/world)); shorter ranks are padded with case_id = -1 records, dropped after the gather.
    world == 1 needs no process group."""
    dtype = records.dtype
    assert dtype.itemsize == RECORD_BYTES
    if world == 1:
        return records.copy()
    per = per_rank if per_rank is not None else -(-n_cases // world)
    assert len(records) <= per
    if comm is not None:                      # libxee's RCCL gather (include/xee.h: xee_gather_scalars)
        out = comm.gather(records, per)
        out = out[out["case_id"] >= 0].copy()
        assert len(out) == n_cases, (len(out), n_cases)
        return out
    import torch
    import torch.distributed as dist
    pad = np.zeros(per, dtype)
    pad["case_id"] = -1
    pad[: len(records)] = records
    send = torch.from_numpy(pad.view(np.uint8).reshape(per, RECORD_BYTES))
    if device is not None:
        send = send.to(device)
    recv = torch.empty((world * per, RECORD_BYTES), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    out = recv.cpu().numpy().reshape(-1).view(dtype)
    out = out[out["case_id"] >= 0].copy()
    assert len(out) == n_cases, (len(out), n_cases)
    return out
