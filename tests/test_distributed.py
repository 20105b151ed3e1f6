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


@pytest.mark.parametrize("world,n_videos", [(2, 5), (2, 2), (2, 1), (4, 4), (4, 6)])
def test_ranks_shard_and_gather(world, n_videos):
    """Videos round-robin over the ranks, one gather of the per-video counts (BASELINE config 4 is 4 videos on 4 GPUs; config 5's
    8-GPU leg the same with more ranks): gloo on the CPU, the same code path RCCL takes."""
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
        assert t == float(world)                                  # max over ranks


def test_single_process_without_group():
    from swiftwatcher_amd import distributed as d
    assert d.shard(7, 1, 3) == [1, 4]
    table = d.run_sharded(3, lambda i: (i, 0, 1))
    assert table.tolist() == [[0, 0, 1], [1, 0, 1], [2, 0, 1]]
    assert d.max_over_ranks(3.5) == 3.5


_ONE_RANK = r"""
import os, sys
sys.path.insert(0, {root!r})
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
import torch
from swiftwatcher_amd import distributed as d
r, w, local = d.init({backend!r}, force=True)
assert (r, w) == (0, 1) and torch.distributed.is_initialized() and torch.distributed.get_backend() == {backend!r}
table = d.run_sharded(3, lambda i: (10 + i, i, 21))           # all_gather on the backend, one rank
assert table.tolist() == [[10, 0, 21], [11, 1, 21], [12, 2, 21]], table
assert d.max_over_ranks(2.5) == 2.5                             # all_reduce(MAX)
d.barrier()
torch.distributed.destroy_process_group()
print("one-rank group ok on", {backend!r})
"""


def _one_rank_group(backend):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _ONE_RANK.format(root=root, port=_free_port(), backend=backend)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "one-rank group ok" in p.stdout


def test_one_rank_group_runs_the_collectives_gloo():
    _one_rank_group("gloo")


@pytest.mark.gpu
def test_one_rank_group_runs_the_collectives_rccl():
    """The per-video count gather, the bench clock's all_reduce(MAX) and the barrier on backend "nccl" (= RCCL on ROCm)
    with a one-rank group on the GPU: the collective code path of distributed.py executes on hardware (a child
    process, so the test session itself never joins a process group)."""
    _one_rank_group("nccl")


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun: the parent (which never touches a GPU) starts two fresh rank processes and
    relays rank 0's line.  --rehearse-launch runs the whole multi-rank protocol (process group, barrier, max-over-ranks clock,
    gather of the per-rank counts) around empty steps, so this runs on a CPU-only host with gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SWK_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch", "--steps", "2", "--windows", "4"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["launched_by"] == "bench.py" and line["backend"] == "gloo"
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["frames"] == 4 * 64 * 2 for r in line["per_rank"])


def test_bench_launcher_ends_the_other_ranks_when_one_dies():
    """A rank that dies before it joins the process group would leave the others in init for ever: the launcher watches
    every child, ends the rest, returns non-zero and keeps each rank's stderr tail."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SWK_DIST_BACKEND="gloo", SWK_REHEARSE_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch", "--steps", "1", "--windows", "2"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert time.time() - t0 < 120                                   # not the process group's own timeout (minutes)
    assert "rank 1 exited with code 3" in out.stderr and "fails on purpose" in out.stderr, out.stderr[-1500:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


_VIDEO_RANK = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
from swiftwatcher_amd import distributed as d, pipeline, synthetic
from swiftwatcher_amd import event_classification as ec
r, w, local = d.init("gloo")                      # two ranks share the one GPU of the box: RCCL refuses that, gloo carries the gather
crop_region = [(30, 20), (30 + 96, 20 + 64)]
roi_mask = np.zeros((64, 96), np.uint8)
roi_mask[28:, :] = 255

def video(i):                                     # "one video per GPU" (BASELINE configs 4 / 5): clip i, counted end to end
    clip = synthetic.full_frames(500 + i, 48 + 5 * i, crop_region, frame_hw=(110, 160), birds=4, bird_len=(8, 12), bird_wid=(3, 5))[::-1].copy()
    count, events = pipeline.count_swifts(list(clip), crop_region, roi_mask, device=0)
    return (count, len(events) - count, len(clip))

table = d.run_sharded({n_videos}, video)
d.barrier()
if r == 0:
    print("TABLE " + json.dumps(table.tolist()))
torch.distributed.destroy_process_group()
"""


@pytest.mark.gpu
def test_videos_sharded_over_two_ranks_on_the_gpu():
    """BASELINE configs 4 / 5 in miniature, on the one GPU a test box has: two ranks (fresh processes), each counts its videos
    round-robin through the whole path (segment on the GPU -> track -> events -> count), one all_gather of the per-video counts;
    the table equals the one a single process computes.  (gloo: two ranks on one device; the RCCL path is covered by the one-rank
    test above and by bench.py on the driver's multi-GPU node.)"""
    import json
    import subprocess
    import sys
    import numpy as np
    from swiftwatcher_amd import pipeline, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_videos, port = 3, _free_port()
    code = _VIDEO_RANK.format(root=root, n_videos=n_videos)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    table = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("TABLE ")][0][6:])
    crop_region = [(30, 20), (30 + 96, 20 + 64)]
    roi_mask = np.zeros((64, 96), np.uint8)
    roi_mask[28:, :] = 255
    expect = []
    for i in range(n_videos):
        clip = synthetic.full_frames(500 + i, 48 + 5 * i, crop_region, frame_hw=(110, 160), birds=4, bird_len=(8, 12), bird_wid=(3, 5))[::-1].copy()
        count, events = pipeline.count_swifts(list(clip), crop_region, roi_mask)
        expect.append([count, len(events) - count, len(clip)])
    assert table == expect
    assert sum(r[0] + r[1] for r in table) >= 1


@pytest.mark.gpu
def test_bench_video_sharded_leg_with_two_ranks_on_one_gpu():
    """The line the driver runs on a multi-GPU node carries BASELINE configs 4 / 5 as worded: `bench.py --gpus 2` (self-launched ranks;
    gloo, because the test box's two ranks share one GPU) -- every rank counts its own video end to end (reader that segments ahead ->
    FrameQueue -> classifier -> tracker -> events -> count), ONE all_gather delivers the per-video (predicted, rejected, frames) table,
    and rank 0's single-process recount of every video gives the same numbers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SWK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--windows", "4",
                          "--no-cpu-baseline", "--no-drop-in", "--video-windows", "4", "--verify-all-videos"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    vs = line["video_sharded"]
    assert line["n_gpus"] == 2 and vs["videos"] == 2 and vs["frames_per_video"] == 84, vs
    assert vs["counts_equal_single_rank"] is True and vs["verified_videos"] == [0, 1]
    assert [row[2] for row in vs["per_video_counts"]] == [84, 84] and vs["value"] > 0
    # the trained weights keep a share of the clip's small faint birds: the gathered table is not all zeros
    assert sum(row[0] + row[1] for row in vs["per_video_counts"]) >= 1, vs
