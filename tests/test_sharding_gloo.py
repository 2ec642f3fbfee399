"""N>1 path on CPU: two processes over gloo (127.0.0.1) -- frame dealing, the barrier around the timed
region, the single metrics all_gather, and the max-over-ranks throughput aggregation bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from onepose_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_frames, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = sharding.init_process_group(backend="gloo")
    assert (r, lr, w) == (rank, rank, world)
    mine = sharding.frames_for_rank(n_frames, rank, world)
    sharding.barrier()
    # pretend rank 1 is the slow one: whole-job time must be the max over ranks
    seconds = 2.0 if rank == 1 else 1.0
    per_rank = sharding.gather_metrics([len(mine), seconds])
    fps, t = sharding.aggregate_throughput(per_rank)
    q.put((rank, mine, per_rank.tolist(), fps, t))
    sharding.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_frame_sharding_and_metrics_gather():
    world, n_frames = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = sorted(res[0][1] + res[1][1])
    assert frames == list(range(n_frames)), "every frame processed exactly once"
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7]
    for _, _, per_rank, fps, t in res:  # identical on every rank
        assert per_rank == [[5.0, 1.0], [4.0, 2.0]]
        assert t == 2.0 and fps == pytest.approx(9 / 2.0)


def test_single_process_paths():
    assert sharding.frames_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    per = sharding.gather_metrics([10.0, 0.5])
    assert per.shape == (1, 2)
    fps, t = sharding.aggregate_throughput(per)
    assert fps == pytest.approx(20.0) and t == 0.5
    sharding.barrier()  # no-op without a process group
