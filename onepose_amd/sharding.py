"""Frame sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

Query frames are independent units: they share only read-only weights and the
per-object 3D database.  Frames are dealt round-robin, there is NO collective on the data path; the
only collectives are a barrier around the timed region and one tiny all_gather of per-rank metrics
(the reference's only collective is likewise a metrics gather, src/utils/comm.py:177-215).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    rank, local_rank, world = dist_env()
    # under torchrun (RANK/WORLD_SIZE exported) the group is initialised even for world == 1, so the
    # single-GPU box exercises the same RCCL path the 8-GPU run uses
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def frames_for_rank(n_frames, rank, world):
    """Indices of the frames rank `rank` processes: frame i -> rank i mod world."""
    return list(range(rank, n_frames, world))


def backend_name():
    """"nccl" (= RCCL on ROCm) / "gloo" when a process group is up (any launch under torchrun, also world 1); None for a bare run."""
    return dist.get_backend() if dist.is_initialized() else None


def barrier():
    if dist.is_initialized():
        dist.barrier()


def gather_metrics(vec, device=None):
    """all_gather of a small per-rank fp32 vector -> [world, len(vec)] tensor (on every rank)."""
    t = torch.as_tensor(vec, dtype=torch.float32, device=device)
    if not dist.is_initialized():
        return t[None].clone()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out)


def aggregate_throughput(per_rank):
    """per_rank [world, >=2]: columns (units processed, seconds).  Whole-job throughput = all units
    divided by the slowest rank's time (max over ranks)."""
    units = float(per_rank[:, 0].sum())
    seconds = float(per_rank[:, 1].max())
    return units / seconds, seconds
