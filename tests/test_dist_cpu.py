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


def test_rank_plans_for_sharded_workloads():
    """bench.py --config c3 (teams of three guidance-branch ranks) and c4 (the world as one tile team): who owns which
    video, how many videos a step makes, which ranks idle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    p = [bench.rank_plan(8, r, "branches") for r in range(8)]
    assert [x["team"] for x in p[:6]] == [0, 0, 0, 1, 1, 1] and all(x["team_size"] == 3 and x["total_videos_per_step"] == 2 for x in p)
    assert [x["idle"] for x in p] == [False] * 6 + [True] * 2 and p[0]["latent_seed"] != p[3]["latent_seed"] and p[0]["latent_seed"] == p[2]["latent_seed"]
    assert p[0]["dit_weight_seed"] == p[5]["dit_weight_seed"]                     # a team shares one model
    one = bench.rank_plan(1, 0, "branches")
    assert one["team_size"] == 1 and one["total_videos_per_step"] == 1 and not one["idle"]
    t = [bench.rank_plan(4, r, "tiles") for r in range(4)]
    assert all(x["team_size"] == 4 and x["total_videos_per_step"] == 1 and x["latent_seed"] == 42 and not x["idle"] for x in t)
    assert set(bench.CONFIGS) == {"c1", "c2", "c3", "c4", "c5"} and bench.CONFIGS["c2"]["num_frames"] == 97
    # algorithmic FLOPs of SURVEY 8d: 22.35 TFLOP per C2 forward, 48.4 per decode, 204.8 per video
    assert abs(bench.dit_flops(4992) / 1e12 - 22.35) < 0.05 and abs(bench.vae_flops(13, 16, 24) / 1e12 - 48.4) < 0.1
