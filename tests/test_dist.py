"""CPU, world_size 2, gloo: the multi-GPU layout (stream sharding + the single all-gather of
probabilities) without GPUs.  Per-rank compute is a stand-in; what is tested is the N>1 plumbing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO
from mycroft_precise_amd.dist import shard_bounds, gather_probabilities, env_world


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 16, 4096, 32768, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            widths = [hi - lo for lo, hi in spans]
            assert max(widths) - min(widths) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_env_world_defaults(monkeypatch):
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    assert env_world() == (0, 0, 1)


def test_single_rank_gather_is_identity():
    x = torch.arange(12.0).view(3, 4)
    assert gather_probabilities(x, 4) is x


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_streams, steps, ret):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_streams, rank, world)
        # stand-in for the per-rank engine: "probability" of global stream s at step k
        sid = torch.arange(lo, hi, dtype=torch.float32)
        local = torch.stack([torch.sin(sid * 0.01 + k) * 0.5 + 0.5 for k in range(steps)])
        full = gather_probabilities(local, n_streams)
        want = torch.stack([torch.sin(torch.arange(n_streams, dtype=torch.float32) * 0.01 + k) * 0.5 + 0.5
                            for k in range(steps)])
        ok = bool(torch.equal(full, want))
        # rank-0 gather variant (what bench.py times)
        on0 = gather_probabilities(local, n_streams, dst=0)
        ok = ok and ((rank == 0 and torch.equal(on0, want)) or (rank != 0 and on0 is None))
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ret[rank] = (ok, float(t.item()), tuple(full.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_streams', [64, 37])          # even and ragged shards
def test_two_rank_gather_restores_global_stream_order(n_streams):
    world, steps = 2, 3
    port = _free_port()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, steps, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        ok, tmax, shape = ret[r]
        assert ok and tmax == float(world) and shape == (steps, n_streams)
