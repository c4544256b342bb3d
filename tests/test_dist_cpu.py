"""world_size-2 gloo test of the clip-axis sharding and the query-embedding all-gather (the N > 1
bench path), on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mraudio_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        full = torch.arange(n_total * 6, dtype=torch.float32).view(n_total, 2, 3)
        lo, hi = parallel.shard_range(n_total, rank, ws)
        got = parallel.all_gather_rows(full[lo:hi].clone(), n_total)
        q.put((rank, bool(torch.equal(got, full)), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


def _run(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_all_gather_equal_shards():
    res = _run(8)
    assert res == [(0, True, (8, 2, 3)), (1, True, (8, 2, 3))]


def test_all_gather_ragged_shards():
    res = _run(7)
    assert res == [(0, True, (7, 2, 3)), (1, True, (7, 2, 3))]


def _worker_packed(rank, ws, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        z = torch.arange(n_total * 4 * 3, dtype=torch.float32).view(n_total, 4, 3)
        c = -torch.arange(n_total * 5, dtype=torch.float32).view(n_total, 5)
        lo, hi = parallel.shard_range(n_total, rank, ws)
        gz, gc = parallel.all_gather_packed([z[lo:hi].clone(), c[lo:hi].clone()], n_total)
        q.put((rank, bool(torch.equal(gz, z)) and bool(torch.equal(gc, c)), tuple(gz.shape), tuple(gc.shape)))
    finally:
        dist.destroy_process_group()


def test_all_gather_packed_is_one_collective_with_the_same_result():
    for n_total in (8, 7):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, n_total, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res == [(0, True, (n_total, 4, 3), (n_total, 5)), (1, True, (n_total, 4, 3), (n_total, 5))]
    # single process: pass-through
    a, b = torch.ones(3, 2), torch.zeros(3)
    assert parallel.all_gather_packed([a, b], 3)[0] is a
