"""The N>1 path on CPU: world_size-2 `gloo` processes, block partition of the case list and the
all-gather of the 128-byte case records (the one collective of the path)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_records(dtype, lo, hi):
    rec = np.zeros(hi - lo, dtype)
    rec["case_id"] = np.arange(lo, hi)
    rec["iters"] = 100000 - np.arange(lo, hi)
    rec["err_now"] = np.float32(1e-3) * (1 + np.arange(lo, hi, dtype=np.float32))
    rec["efficiency"] = np.float32(0.125) * np.arange(lo, hi, dtype=np.float32)
    return rec


def _worker(rank, world, port, n_cases, q):
    import torch.distributed as dist
    import xee_pkg
    xee = xee_pkg.load()
    from xlab_ee_fortran_amd import ensemble_dp as dp
    from xlab_ee_fortran_amd.xee import RECORD_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = dp.partition(n_cases, world, rank)
    allrec = dp.gather_records(_fake_records(RECORD_DTYPE, lo, hi), n_cases, world, rank)
    want = _fake_records(RECORD_DTYPE, 0, n_cases)
    q.put((rank, bool(allrec.tobytes() == want.tobytes()), len(allrec)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_cases", [8, 7, 1])
def test_gloo_world2_gather(n_cases):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_cases, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok and n == n_cases for _, ok, n in res)


def test_partition_covers_every_case_once():
    import xee_pkg
    xee_pkg.load()
    from xlab_ee_fortran_amd import ensemble_dp as dp
    for n in (1, 7, 8, 256, 384):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = dp.partition(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
