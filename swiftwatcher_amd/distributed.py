"""One process per GPU; videos (or batches of RPCA windows) shard across ranks with NO collective in
the data path -- every window is independent (the reference loops over videos
sequentially, __main__.py:21).  The single exchange is an all-gather of a few int64 per video at the
end (predicted / rejected / frames, io_data.py:113), over RCCL when the ranks own GPUs (backend
"nccl" is RCCL on ROCm) or gloo on CPU-only hosts (tests)."""
import os

import torch
import torch.distributed as dist

COUNT_FIELDS = 3            # per video: predicted swifts, rejected events, frames processed


def init(backend=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun sets them).
    Returns (rank, world, local_rank).  With WORLD_SIZE unset or 1 nothing is initialised unless force=True
    (a one-rank group: the collectives below then really run on the backend, which is how the single-GPU test
    exercises the RCCL path)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard(n_items, rank, world):
    """Indices of the videos rank `rank` owns: r, r + world, ... (configs 4/5: one video per GPU)."""
    return list(range(rank, n_items, world))


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_counts(local_counts, n_items):
    """local_counts: {video index: (predicted, rejected, frames)} for the videos this rank processed.
    Returns an (n_items, 3) int64 CPU tensor, identical on every rank, row i = video i."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    per_rank = (n_items + world - 1) // world
    dev = _comm_device()
    mine = torch.full((per_rank, COUNT_FIELDS + 1), -1, dtype=torch.int64, device=dev)
    for slot, idx in enumerate(shard(n_items, rank, world)):
        vals = local_counts[idx]
        mine[slot, 0] = idx
        mine[slot, 1:] = torch.tensor([int(v) for v in vals], dtype=torch.int64)
    if dist.is_initialized():
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    table = torch.zeros((n_items, COUNT_FIELDS), dtype=torch.int64)
    seen = torch.zeros(n_items, dtype=torch.bool)
    for p in parts:
        p = p.cpu()
        for row in p:
            i = int(row[0])
            if i >= 0:
                table[i] = row[1:]
                seen[i] = True
    if not bool(seen.all()):
        raise RuntimeError("some videos were not reported by any rank")
    return table


def max_over_ranks(value):
    """The benchmark's clock: the slowest rank's elapsed time."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def barrier():
    if dist.is_initialized():
        dist.barrier()


def run_sharded(n_videos, process_video):
    """process_video(index) -> (predicted, rejected, frames).  Every rank processes its shard and gets
    the full per-video table back."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    local = {i: process_video(i) for i in shard(n_videos, rank, world)}
    return gather_counts(local, n_videos)
