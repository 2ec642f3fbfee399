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
    """all_gather of a small per-rank float64 vector -> [world, len(vec)] tensor (on every rank).  float64 so that the
    device-identity keys of `device_identity` (integers below 2**53) ride in the same, single collective as the timings."""
    t = torch.as_tensor(vec, dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return t[None].clone()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out)


def device_identity(device=None):
    """(key, description) of the accelerator this rank drives.  key: an integer < 2**53 that is equal on two ranks only if they
    drive the same physical device -- PCI domain / bus / device where the runtime exposes them, plus 24 bits of the device
    UUID -- so that the rank-0 line of an N-rank job can say how many DISTINCT devices the ranks saw (the reference has no such
    check: its gathers are src/utils/comm.py:141-215).  CPU stand-in ranks (bench.py --dry-run) are keyed by process id."""
    if device is None or not torch.cuda.is_available():
        return float(os.getpid() & ((1 << 40) - 1)), f"cpu:pid{os.getpid()}"
    idx = device.index if isinstance(device, torch.device) else int(device)
    p = torch.cuda.get_device_properties(idx)
    dom, bus, dev = (int(getattr(p, a, -1)) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    uuid = getattr(p, "uuid", None)
    ub = getattr(uuid, "bytes", None) if uuid is not None else None
    u24 = int.from_bytes(ub[-3:], "big") if ub else 0
    if min(dom, bus, dev) < 0 and not ub:      # nothing physical to key on: the visible ordinal is all there is
        return float((1 << 52) + idx), f"ordinal:{idx} ({p.name})"
    key = ((dom & 0xFFFF) << 37) | ((bus & 0xFF) << 29) | ((dev & 0x1F) << 24) | u24
    return float(key), f"pci {dom & 0xFFFF:04x}:{bus & 0xFF:02x}:{dev & 0x1F:02x} uuid..{u24:06x} ({p.name})"


def pin_launch_thread(device=None):
    """Pin this rank's (single) launch thread to the cores of the GPU's NUMA node when the kernel exposes it
    (/sys/bus/pci/devices/<bdf>/numa_node), else leave the affinity alone.  Eight Python launch loops at ~60 k launches/s each
    are host-sensitive; torch's intra-op pool is cut to one thread for the same reason.  Returns (description, previous
    affinity or None)."""
    torch.set_num_threads(1)
    if device is None or not torch.cuda.is_available() or not hasattr(os, "sched_setaffinity"):
        return "unpinned (no GPU / no sched_setaffinity)", None
    try:
        p = torch.cuda.get_device_properties(device.index if isinstance(device, torch.device) else int(device))
        bdf = f"{int(p.pci_domain_id) & 0xFFFF:04x}:{int(p.pci_bus_id) & 0xFF:02x}:{int(p.pci_device_id) & 0x1F:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read())
        if node < 0:
            return f"unpinned ({bdf}: no NUMA node reported)", None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        prev = os.sched_getaffinity(0)
        cpus &= prev
        if not cpus:
            return f"unpinned (NUMA node {node} has no allowed core)", None
        os.sched_setaffinity(0, cpus)
        return f"NUMA node {node} of {bdf}: {len(cpus)} cores", prev
    except (OSError, ValueError, AttributeError) as e:
        return f"unpinned ({type(e).__name__})", None


def aggregate_throughput(per_rank):
    """per_rank [world, >=2]: columns (units processed, seconds).  Whole-job throughput = all units
    divided by the slowest rank's time (max over ranks)."""
    units = float(per_rank[:, 0].sum())
    seconds = float(per_rank[:, 1].max())
    return units / seconds, seconds
