"""CPU, world_size 2 over gloo: the multi-GPU path is "independent replicas + a
barrier + max-over-ranks" (bench.py); sessions are partitioned with no data-path
collective."""
import os
import socket

import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sessions, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from livetalking_amd.sharding import shard_for_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_for_rank(n_sessions, world, rank)
    # every rank "renders" its own sessions; only bookkeeping crosses ranks
    frames = torch.tensor([len(mine) * 16], dtype=torch.int64)
    elapsed = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(frames, op=dist.ReduceOp.SUM)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        q.put((int(frames.item()), float(elapsed.item()), gathered))
    dist.destroy_process_group()


def test_two_rank_session_sharding():
    world, n_sessions = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_sessions, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames, elapsed, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert frames == n_sessions * 16
    assert elapsed == pytest.approx(0.75)
    assert sorted(sum(gathered, [])) == list(range(n_sessions))
    assert abs(len(gathered[0]) - len(gathered[1])) <= 1
