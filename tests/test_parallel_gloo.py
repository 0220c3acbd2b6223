"""CPU, world_size 2 over gloo: the N > 1 layout of the hot path -- contiguous sharding of frame pairs and
the single flat all-gather of the per-pair records -- without a GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vo_single_camera_sos_amd.parallel import RECORD_WIDTH, gather_records, max_over_ranks, shard_range, shard_rows_max


def test_shard_range_is_a_contiguous_partition():
    for n, w in [(512, 8), (10, 3), (7, 8), (0, 4), (64, 1)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 3, 8) == (192, 256)  # C4: 512 pairs over 8 GPUs, blocks of 64
    assert shard_rows_max(10, 3) == 4 and shard_rows_max(512, 8) == 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # the record of global pair g carries g in every column (what a rank's pipeline.results() would hold)
    rec = (torch.arange(lo, hi, dtype=torch.float64)[:, None].repeat(1, RECORD_WIDTH)
           + torch.arange(RECORD_WIDTH, dtype=torch.float64) * 1e-3)
    full = gather_records(rec) if n_total % world == 0 else gather_records(rec, n_total=n_total)
    if n_total % world != 0:
        # shards whose sizes differ by one must be refused by the equal-count form, on every rank, before the
        # collective that would hang on RCCL
        try:
            gather_records(rec)
            full = None
        except ValueError:
            pass
    slowest = max_over_ranks(1.0 + rank, torch.device("cpu"))
    dist.barrier()
    q.put((rank, None if full is None else full.numpy(), slowest))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,n_total", [(2, 12), (2, 7), (3, 10)])
def test_gather_records_in_global_pair_order(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(n_total, dtype=np.float64)[:, None] + np.arange(RECORD_WIDTH) * 1e-3
    for rank, full, slowest in results:
        assert full is not None, "the equal-count gather accepted uneven shards"
        assert full.shape == (n_total, RECORD_WIDTH)
        assert np.array_equal(full, want)      # rank order == global pair order
        assert slowest == float(world)        # max over ranks
