"""CPU suite, part 3: the multi-rank bookkeeping of bench.py (replicas-only sharding, MAX-over-ranks timing,
whole-job aggregation) under torch.distributed gloo, world_size 2."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = bench.rank_plan(world, rank)
    elapsed = bench.reduce_elapsed(1.0 + rank, dist, "cpu")          # rank 1 is the slow one
    seeds = [None] * world
    dist.all_gather_object(seeds, plan["latent_seed"])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, elapsed, seeds, plan["total_videos_per_step"], bench.job_fps(world, 3, 97, elapsed)))


def test_two_rank_replica_plan_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, elapsed, seeds, total, fps in res:
        assert elapsed == 2.0                      # MAX over ranks
        assert seeds == [42, 43] and total == 2    # disjoint work, no overlap
        assert abs(fps - 2 * 3 * 97 / 2.0) < 1e-9  # whole-job aggregate
