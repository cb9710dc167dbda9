"""CPU suite, part 4: the multi-GPU forms of the path under torch.distributed gloo.

`ltxhip.sharded` is pure orchestration over injected callables; here they are bound to the CPU oracle, so the test
pins the collectives' bookkeeping (who runs which guidance branch / which VAE tile, what is gathered, in which order the
tiles are blended) against the oracle's single-process `pipeline_call` / `vae_decode`: results must be bit-identical
on every rank."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "candle-video_amd"))


def _setup(rank, world, port):
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "candle-video_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _dit_problem():
    import ltx_oracle as O
    from tools_cfg import PIPE_DIT_CFG
    cfg = O.DitConfig(**PIPE_DIT_CFG)
    w = O.synth_weights(O.dit_weight_shapes(cfg), seed=11)
    g = torch.Generator().manual_seed(5)
    F, H, W, K = 2, 2, 3, 6
    lat = torch.randn(1, F * H * W, cfg.in_channels, generator=g)
    pe, ne = torch.randn(1, K, cfg.caption_channels, generator=g), torch.randn(1, K, cfg.caption_channels, generator=g)
    pm = torch.ones(1, K); pm[0, 4:] = 0
    nm = torch.ones(1, K); nm[0, 2:] = 0
    return O, cfg, w, (F, H, W), lat, pe, pm, ne, nm


def _branch_worker(rank, world, port, team_size, q):
    _setup(rank, world, port)
    from ltxhip import sharded as S
    O, cfg, w, (F, H, W), lat, pe, pm, ne, nm = _dit_problem()
    gs, gr, stg, skip = 3.0, 0.7, 1.0, [1]
    team = S.make_teams(team_size)
    sched = O.FlowMatchEulerScheduler(O.SchedulerCfg())
    sig = list(O.FlowMatchEulerScheduler._linspace(1.0, 1.0 / 4, 4))
    ts = sched.set_timesteps(sigmas=sig, mu=O.calculate_shift(F * H * W))
    coords = O.build_video_coords(1, F, H, W, 25, 8, 32)
    calls = []

    def fwd(name, latents, t):
        calls.append(name)
        emb, mask = (ne, nm) if name == S.BRANCH_UNCOND else (pe, pm)
        slm = None
        if name == S.BRANCH_PERTURBED:
            slm = torch.zeros(cfg.num_layers, 1); slm[skip[0]] = 1.0
        return O.dit_forward(w, cfg, latents, emb, torch.full((1,), float(t)), mask, F, H, W, None, coords, slm, (), torch.float32)

    def gstep(preds, latents, dt):
        return latents + O.guidance_combine(preds[S.BRANCH_TEXT], preds.get(S.BRANCH_UNCOND), preds.get(S.BRANCH_PERTURBED), gs, gr, stg) * dt

    out = S.denoise_branch_sharded(fwd, gstep, lat.clone(), S.guidance_branches(gs, stg), list(sched.sigmas), ts, team)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out.numpy(), sorted(set(calls)), len(calls), team.index, team.rank, team.size))


def _spawn(target, world, extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 21000 + (os.getpid() * 7 + world * 131 + hash(extra) % 97) % 8000
    procs = [ctx.Process(target=target, args=(r, world, port) + extra + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    # tensors travel as numpy arrays (pickled by value): a torch tensor in a Queue is an fd the exited worker no longer serves
    res = sorted(((lambda t: (t[0], torch.from_numpy(t[1])) + tuple(t[2:]))(q.get(timeout=300)) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,team_size", [(2, 2), (3, 3)])
def test_guidance_branches_sharded_match_single_process_trajectory(world, team_size):
    O, cfg, w, (F, H, W), lat, pe, pm, ne, nm = _dit_problem()
    from tools_cfg import VAE_CFG
    args = O.PipelineArgs(height=H * 32, width=W * 32, num_frames=(F - 1) * 8 + 1, num_inference_steps=4, guidance_scale=3.0,
                          guidance_rescale=0.7, stg_scale=1.0, skip_block_list=[1], output_latent=True)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)            # same thread count as the workers (bit-exact comparison)
    try:
        want = O.pipeline_call(w, cfg, None, O.VaeConfig(**VAE_CFG), None, None, args, lat, pe, pm, ne, nm)
    finally:
        torch.set_num_threads(nthreads)
    res = _spawn(_branch_worker, world, (team_size,))
    per_rank_calls = []
    for rank, out, names, ncalls, tidx, trank, tsize in res:
        assert torch.equal(out, want), f"rank {rank}"          # replicated latents, identical to the serial loop
        assert tsize == team_size and tidx == 0 and trank == rank
        per_rank_calls.append((names, ncalls))
    if world == 3:      # one branch per rank, 4 steps each
        assert sorted(n[0][0] for n in per_rank_calls) == ["perturbed", "text", "uncond"] and all(n[1] == 4 for n in per_rank_calls)
    else:               # 3 branches over 2 ranks: rank 0 runs two per step, rank 1 one (its second slot is padding)
        assert per_rank_calls[0] == (["text", "uncond"], 8) and per_rank_calls[1] == (["perturbed"], 4)


def _vae_problem():
    import ltx_oracle as O
    from tools_cfg import VAE_CFG
    cfg = O.VaeConfig(**VAE_CFG)
    cfg.tile_sample_min_height = cfg.tile_sample_min_width = 64
    cfg.tile_sample_stride_height = cfg.tile_sample_stride_width = 32
    cfg.tile_sample_min_num_frames, cfg.tile_sample_stride_num_frames = 16, 8
    w = O.synth_weights(O.vae_decoder_weight_shapes(cfg), seed=7)
    z = torch.randn(1, cfg.latent_channels, 4, 3, 3, generator=torch.Generator().manual_seed(9))
    return O, cfg, w, z, torch.full((1,), 0.05)


def _tile_worker(rank, world, port, framewise, q):
    _setup(rank, world, port)
    from ltxhip import sharded as S
    O, cfg, w, z, temb = _vae_problem()
    team = S.make_teams(world)
    tl = S.Tiling(True, bool(framewise), 64, 64, 16, 32, 32, 8, 32, 8)
    n = []

    def dec(zc):
        n.append(tuple(zc.shape))
        return O.decoder_forward(w, cfg, zc, temb, torch.float32)

    out = S.decode_tile_sharded(dec, O._blend, z, tl, team)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out.numpy(), len(n), len(S.leaf_crops(tl, *z.shape[2:])), [tuple(r) for r in S.temporal_tile_owner(len(S._temporal_ranges(tl, z.shape[2])), world)]))


@pytest.mark.parametrize("world,framewise", [(2, 0), (2, 1), (3, 1), (5, 1)])
def test_vae_tiles_sharded_match_single_process_tiled_decode(world, framewise):
    O, cfg, w, z, temb = _vae_problem()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)            # torch's CPU conv sums in a thread-count-dependent order; the workers use 2
    try:
        want = O.vae_decode(w, cfg, z, temb, torch.float32, use_tiling=True, use_framewise_decoding=bool(framewise))
    finally:
        torch.set_num_threads(nthreads)
    res = _spawn(_tile_worker, world, (framewise,))
    total = 0
    for rank, out, ndec, nleaf, owner in res:
        assert out.shape == want.shape and torch.equal(out, want), f"rank {rank}"
        if framewise:            # whole temporal tiles per rank (contiguous ranges), 9 spatial leaves each here
            assert ndec == 9 * len(owner[rank]) and max(len(o) for o in owner) - min(len(o) for o in owner) <= 1
        else:
            assert ndec in (nleaf // world, nleaf // world + 1)
        total += ndec
    assert total == res[0][3]                                   # every leaf tile decoded exactly once across the team


def test_leaf_crops_cover_reference_tile_loops():
    from ltxhip import sharded as S
    tl = S.Tiling()                                             # reference defaults: 512/384 px, 16/8 frames
    # headline latent grid 13x16x24: spatial-only -> 2x2 tiles; framewise -> 13 temporal windows x 4
    assert len(S.leaf_crops(tl, 13, 16, 24)) == 4
    tl.use_framewise_decoding = True
    crops = S.leaf_crops(tl, 13, 16, 24)
    assert len(crops) == 52 and crops[0] == (0, 3, 0, 16, 0, 16) and crops[-1][0:2] == (12, 13)
    tl2 = S.Tiling(use_tiling=False)
    assert S.leaf_crops(tl2, 13, 16, 24) == [(0, 13, 0, 16, 0, 24)]
    assert S.guidance_branches(1.0, 0.0) == ["text"] and S.guidance_branches(3.0, 1.0) == ["uncond", "text", "perturbed"]
    assert S.branch_owner(3, 2) == (2, [[0, 1], [2]]) and S.branch_owner(2, 3) == (1, [[0], [1], []])
    assert [list(r) for r in S.temporal_tile_owner(13, 8)] == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10], [11], [12]]
    assert [list(r) for r in S.temporal_tile_owner(2, 3)] == [[0], [1], []]
