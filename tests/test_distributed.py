"""N > 1 path on CPU: two gloo ranks shard "videos" and all-gather the per-video counts."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_videos, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from swiftwatcher_amd import distributed as d
    r, w, _ = d.init("gloo")
    assert (r, w) == (rank, world)
    calls = []

    def process(i):
        calls.append(i)
        return (100 + i, i % 3, 21 * (i + 1))
    table = d.run_sharded(n_videos, process)
    t = d.max_over_ranks(1.0 + rank)
    d.barrier()
    q.put((rank, calls, table.tolist(), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_videos", [5, 2, 1])
def test_two_ranks_shard_and_gather(n_videos):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_videos, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    expect = [[100 + i, i % 3, 21 * (i + 1)] for i in range(n_videos)]
    for rank, calls, table, t in out:
        assert calls == list(range(rank, n_videos, world))      # round-robin shard, no overlap
        assert table == expect                                    # identical full table on every rank
        assert t == 2.0                                           # max over ranks


def test_single_process_without_group():
    from swiftwatcher_amd import distributed as d
    assert d.shard(7, 1, 3) == [1, 4]
    table = d.run_sharded(3, lambda i: (i, 0, 1))
    assert table.tolist() == [[0, 0, 1], [1, 0, 1], [2, 0, 1]]
    assert d.max_over_ranks(3.5) == 3.5
