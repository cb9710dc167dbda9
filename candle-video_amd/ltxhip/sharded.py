"""Multi-GPU forms of the path: one process per GPU, `torch.distributed` over RCCL/xGMI
(backend "nccl" IS RCCL on ROCm; the CPU tests run the same code over "gloo").

The reference has no collective at all (single device, `examples/ltx-video/main.rs:210-214`), so only the two
places where `LtxPipeline::call` naturally partitions get one:

  * guidance branches (`t2v_pipeline.rs:878-939`): with CFG and/or STG a denoise step runs 2-3 INDEPENDENT
    transformer forwards on identical latents.  A team of ranks splits the branches, all-gathers the f32 noise
    predictions ([B,S,128] f32 = 2.56 MB per rank at 2B/4992 tokens, latency-bound next to a ~25 ms forward) and
    every rank applies the fused guidance + Euler update redundantly, so latents stay replicated without a broadcast.
  * VAE tiles (`vae.rs:2225-2290, 2358-2434`): the reference's tiled decode is a serial loop over independent
    `decoder.forward` calls.  With framewise decoding whole TEMPORAL tiles go to ranks in contiguous ranges: spatial
    blends stay local, one strip of <= 8 frames goes to the next rank for the temporal blend, and only finished
    frames are all-gathered (each frame travels once).  Spatial-only tiling (four leaves at C2) deals the leaves
    round-robin and gathers them whole.  Either way blends run in exactly the reference order: numerics = the tiled path.

Guidance-free presets (the distilled headline config) have ONE forward per step and do not shard: those run as
independent replicas, one video per GPU (`bench.py --gpus N`), with no data-path collective.

Everything here is orchestration: the transformer forward, guidance/Euler kernel, tile decode and blend are
injected callables (`HipOps` binds them to libltxhip; the gloo tests bind them to the CPU oracle).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

try:                                   # torch.distributed is plumbing; single-process use needs none of it
    import torch.distributed as dist
except Exception:                      # pragma: no cover
    dist = None

BRANCH_UNCOND, BRANCH_TEXT, BRANCH_PERTURBED = "uncond", "text", "perturbed"


def guidance_branches(guidance_scale: float, stg_scale: float) -> List[str]:
    """Order of `t2v_pipeline.rs:878-939`: uncond (iff guidance_scale > 1), text, perturbed (iff stg_scale > 0)."""
    out = []
    if guidance_scale > 1.0:
        out.append(BRANCH_UNCOND)
    out.append(BRANCH_TEXT)
    if stg_scale > 0.0:
        out.append(BRANCH_PERTURBED)
    return out


@dataclass
class Team:
    """Ranks that cooperate on ONE video.  `group` None = single process (size 1)."""
    group: object = None
    rank: int = 0          # rank inside the team
    size: int = 1
    index: int = 0         # which team (= which video of the job) this process belongs to
    count: int = 1         # teams in the job


def make_teams(team_size: int) -> Team:
    """Split the world into consecutive teams of `team_size` ranks (leftover ranks form a last, smaller team).
    Collective: every rank must call it (dist.new_group semantics)."""
    if dist is None or not dist.is_initialized():
        return Team()
    world, rank = dist.get_world_size(), dist.get_rank()
    team_size = max(1, min(team_size, world))
    count = math.ceil(world / team_size)
    mine = None
    for t in range(count):
        ranks = list(range(t * team_size, min((t + 1) * team_size, world)))
        g = dist.new_group(ranks)
        if rank in ranks:
            mine = Team(g, ranks.index(rank), len(ranks), t, count)
    return mine


def branch_owner(n_branches: int, team_size: int) -> Tuple[int, List[List[int]]]:
    """Contiguous deal of branch indices to team ranks: returns (slots per rank, owned indices per rank)."""
    per = math.ceil(n_branches / team_size)
    owned = [[b for b in range(r * per, min((r + 1) * per, n_branches))] for r in range(team_size)]
    return per, owned


def _all_gather(x: torch.Tensor, team: Team) -> List[torch.Tensor]:
    if team.size == 1:
        return [x]
    outs = [torch.empty_like(x) for _ in range(team.size)]
    dist.all_gather(outs, x.contiguous(), group=team.group)
    return outs


def denoise_branch_sharded(forward_branch: Callable[[str, torch.Tensor, int], torch.Tensor],
                           guidance_step: Callable[[dict, torch.Tensor, float], torch.Tensor],
                           latents: torch.Tensor, branches: Sequence[str], sigmas: Sequence[float],
                           timesteps: Sequence[int], team: Team) -> torch.Tensor:
    """The denoising loop of `LtxPipeline::call` (t2v_pipeline.rs:860-994) with the per-step forwards split over a team.

    forward_branch(name, latents, timestep) -> f32 prediction [B,S,C] of that guidance branch;
    guidance_step(preds, latents, dt) -> latents after CFG/STG mix + rescale + Euler step (dt = sigma_next - sigma).
    Every rank returns the same latents (bit-identical: all ranks apply the same kernel to the same gathered bytes)."""
    nb = len(branches)
    per, owned = branch_owner(nb, team.size)
    mine = owned[team.rank]
    for i, t in enumerate(timesteps):
        slots = []
        for k in range(per):
            if k < len(mine):
                slots.append(forward_branch(branches[mine[k]], latents, int(t)).float())
            else:                                        # this rank has no branch for the slot: pad the collective
                slots.append(torch.zeros_like(latents, dtype=torch.float32))
        gathered = _all_gather(torch.stack(slots, 0), team)      # team.size x [per, B, S, C]
        preds = {}
        for r in range(team.size):
            for k, b in enumerate(owned[r]):
                preds[branches[b]] = gathered[r][k]
        # f32 subtraction of the two f32 sigmas, as `scheduler.step` does (scheduler.rs:544-581)
        dt = float(torch.tensor(sigmas[i + 1], dtype=torch.float32) - torch.tensor(sigmas[i], dtype=torch.float32))
        latents = guidance_step(preds, latents, dt)
    return latents


# ------------------------------------------------------------------ tile-sharded VAE decode
@dataclass
class Tiling:
    """Tiling fields of AutoencoderKLLtxVideo (vae.rs:1744-1758), sample-space units."""
    use_tiling: bool = True
    use_framewise_decoding: bool = False
    tile_sample_min_height: int = 512
    tile_sample_min_width: int = 512
    tile_sample_min_num_frames: int = 16
    tile_sample_stride_height: int = 384
    tile_sample_stride_width: int = 384
    tile_sample_stride_num_frames: int = 8
    spatial_compression_ratio: int = 32
    temporal_compression_ratio: int = 8


Crop = Tuple[int, int, int, int, int, int]          # latent-space (t0, t1, h0, h1, w0, w1)


def _spatial_crops(tl: Tiling, t0: int, t1: int, height: int, width: int) -> List[List[Crop]]:
    r = tl.spatial_compression_ratio
    tmin_h, tmin_w = tl.tile_sample_min_height // r, tl.tile_sample_min_width // r
    ts_h, ts_w = tl.tile_sample_stride_height // r, tl.tile_sample_stride_width // r
    return [[(t0, t1, i, min(i + tmin_h, height), j, min(j + tmin_w, width)) for j in range(0, width, ts_w)]
            for i in range(0, height, ts_h)]


def _mode(tl: Tiling, F: int, H: int, W: int) -> str:
    """Dispatch of decode_z (vae.rs:2037-2066)."""
    r, tr = tl.spatial_compression_ratio, tl.temporal_compression_ratio
    if tl.use_framewise_decoding and F > tl.tile_sample_min_num_frames // tr:
        return "temporal"
    if tl.use_tiling and (W > tl.tile_sample_min_width // r or H > tl.tile_sample_min_height // r):
        return "spatial"
    return "direct"


def _temporal_ranges(tl: Tiling, F: int) -> List[Tuple[int, int]]:
    tr = tl.temporal_compression_ratio
    tmin_t, tstride_t = tl.tile_sample_min_num_frames // tr, tl.tile_sample_stride_num_frames // tr
    return [(i, min(i + tmin_t + 1, F)) for i in range(0, F, tstride_t)]


def _temporal_is_tiled(tl: Tiling, H: int, W: int) -> bool:
    r = tl.spatial_compression_ratio
    return tl.use_tiling and (H > tl.tile_sample_min_height // r or W > tl.tile_sample_min_width // r)


def leaf_crops(tl: Tiling, F: int, H: int, W: int) -> List[Crop]:
    """Every independent `decoder.forward` call of the reference's tiled decode, in its loop order."""
    mode = _mode(tl, F, H, W)
    if mode == "direct":
        return [(0, F, 0, H, 0, W)]
    if mode == "spatial":
        return [c for row in _spatial_crops(tl, 0, F, H, W) for c in row]
    out: List[Crop] = []
    for t0, t1 in _temporal_ranges(tl, F):
        if _temporal_is_tiled(tl, H, W):
            out += [c for row in _spatial_crops(tl, t0, t1, H, W) for c in row]
        else:
            out.append((t0, t1, 0, H, 0, W))
    return out


def _assemble_spatial(tl: Tiling, rows: List[List[torch.Tensor]], height: int, width: int, blend) -> torch.Tensor:
    """Blend/crop/concat of tiled_decode (vae.rs:2259-2290)."""
    r = tl.spatial_compression_ratio
    blend_h = max(tl.tile_sample_min_height - tl.tile_sample_stride_height, 0)
    blend_w = max(tl.tile_sample_min_width - tl.tile_sample_stride_width, 0)
    prev: List[torch.Tensor] = []
    result_rows = []
    for ri, row in enumerate(rows):
        cur: List[torch.Tensor] = []
        out_row = []
        for cj, tile in enumerate(row):
            if ri > 0:
                tile = blend(prev[cj], tile, blend_h, 3)
            if cj > 0:
                tile = blend(cur[cj - 1], tile, blend_w, 4)
            cur.append(tile)
            hs = min(tl.tile_sample_stride_height, tile.shape[3])
            ws = min(tl.tile_sample_stride_width, tile.shape[4])
            out_row.append(tile[:, :, :, :hs, :ws])
        result_rows.append(torch.cat(out_row, 4))
        prev = cur
    dec = torch.cat(result_rows, 3)
    return dec[:, :, :, :height * r, :width * r]


def assemble(tl: Tiling, F: int, H: int, W: int, leaves: Sequence[torch.Tensor], blend) -> torch.Tensor:
    """Rebuild the video from decoded leaf tiles (same order as `leaf_crops`) exactly as the reference does."""
    mode = _mode(tl, F, H, W)
    it = iter(leaves)
    if mode == "direct":
        return next(it)
    if mode == "spatial":
        grid = _spatial_crops(tl, 0, F, H, W)
        return _assemble_spatial(tl, [[next(it) for _ in row] for row in grid], H, W, blend)
    tr = tl.temporal_compression_ratio
    num_sample_frames = (F - 1) * tr + 1
    blend_t = max(tl.tile_sample_min_num_frames - tl.tile_sample_stride_num_frames, 0)
    row = []
    for li, (t0, t1) in enumerate(_temporal_ranges(tl, F)):           # vae.rs:2382-2408
        if _temporal_is_tiled(tl, H, W):
            grid = _spatial_crops(tl, t0, t1, H, W)
            dec = _assemble_spatial(tl, [[next(it) for _ in r_] for r_ in grid], H, W, blend)
        else:
            dec = next(it)
        if li > 0 and dec.shape[2] > 1:
            dec = dec[:, :, :-1]
        row.append(dec)
    out = []
    for idx, tile in enumerate(row):                                   # vae.rs:2410-2434
        if idx > 0:
            bl = blend(row[idx - 1], tile, blend_t, 2)
            out.append(bl[:, :, :min(tl.tile_sample_stride_num_frames, bl.shape[2])])
        else:
            out.append(tile[:, :, :min(tl.tile_sample_stride_num_frames + 1, tile.shape[2])])
    return torch.cat(out, 2)[:, :, :num_sample_frames]


def temporal_tile_owner(n_tiles: int, team_size: int) -> List[range]:
    """Contiguous, balanced deal of temporal tiles to team ranks (the first n_tiles % team_size ranks get one more)."""
    base, extra = divmod(n_tiles, team_size)
    out, start = [], 0
    for r in range(team_size):
        n = base + (1 if r < extra else 0)
        out.append(range(start, start + n))
        start += n
    return out


def _decode_temporal_sharded(decode_tile, blend, z: torch.Tensor, tl: Tiling, team: Team) -> torch.Tensor:
    """Framewise (temporal) tiled decode, vae.rs:2358-2434, with TEMPORAL tiles as the unit of distribution.

    A temporal tile (its spatial leaves, their spatial blend, vae.rs:2225-2290) is decoded and assembled on ONE rank.  The
    only value that crosses tiles is `blend_t(row[i-1], row[i])`, which reads the last `blend` frames of the previous
    tile as decoded (not as blended - no chain): ranks own CONTIGUOUS ranges of temporal tiles, so a rank receives one
    strip of <= blend frames from its predecessor (point to point) and finishes its kept frames locally.  What is
    gathered is the finished video, each frame once: 458 MB of f32 at C2 instead of the 2.8 GB of whole padded tiles
    the round-robin leaf deal moved (SURVEY 8e "crop before send").  Bit-identical to the serial loop."""
    B, _, F, H, W = z.shape
    tr = tl.temporal_compression_ratio
    ranges = _temporal_ranges(tl, F)
    nT = len(ranges)
    owner = temporal_tile_owner(nT, team.size)
    mine = owner[team.rank]
    blend_t = max(tl.tile_sample_min_num_frames - tl.tile_sample_stride_num_frames, 0)
    stride_t = tl.tile_sample_stride_num_frames
    num_sample_frames = (F - 1) * tr + 1
    tiled = _temporal_is_tiled(tl, H, W)
    row = {}
    # every leaf this rank owns, decoded up front: a `decode_tile` that carries a `.many(list of crops)` attribute (HipOps: leaves
    # of one shape stacked along the batch axis, as the single-GPU engine does) gets them all at once; results per leaf are the same
    grids = {li: (_spatial_crops(tl, ranges[li][0], ranges[li][1], H, W) if tiled else [[(ranges[li][0], ranges[li][1], 0, H, 0, W)]]) for li in mine}
    flat = [(li, ri, ci, c) for li in mine for ri, r_ in enumerate(grids[li]) for ci, c in enumerate(r_)]
    crops = [z[:, :, c[0]:c[1], c[2]:c[3], c[4]:c[5]].contiguous() for (_, _, _, c) in flat]
    many = getattr(decode_tile, "many", None)
    decs = many(crops) if many is not None else [decode_tile(c) for c in crops]
    leaf = {(li, ri, ci): d.float() for (li, ri, ci, _), d in zip(flat, decs)}
    for li in mine:                                                   # vae.rs:2382-2408
        if tiled:
            dec = _assemble_spatial(tl, [[leaf[(li, ri, ci)] for ci in range(len(r_))] for ri, r_ in enumerate(grids[li])], H, W, blend)
        else:
            dec = leaf[(li, 0, 0)]
        if li > 0 and dec.shape[2] > 1:
            dec = dec[:, :, :-1]
        row[li] = dec
    # strip exchange: the tail of my last tile goes to the rank that owns the next tile
    def frames_of(li):                                                # decoded length of temporal tile li (after the drop)
        n = (ranges[li][1] - ranges[li][0] - 1) * tr + 1
        return n - 1 if (li > 0 and n > 1) else n
    prev_tail = None
    if team.size > 1:
        ops_ = []
        first, last = (mine[0], mine[-1]) if len(mine) else (None, None)
        send_to = next((r for r in range(team.rank + 1, team.size) if len(owner[r])), None) if len(mine) and last < nT - 1 else None
        recv_from = next((r for r in range(team.rank - 1, -1, -1) if len(owner[r])), None) if len(mine) and first > 0 else None
        Hs, Ws = H * tl.spatial_compression_ratio, W * tl.spatial_compression_ratio
        if send_to is not None:
            e = min(blend_t, row[last].shape[2])
            tail = row[last][:, :, row[last].shape[2] - e:].contiguous()
            ops_.append(dist.P2POp(dist.isend, tail, _global_rank(team, send_to), group=team.group))
        if recv_from is not None:
            e = min(blend_t, frames_of(first - 1))
            prev_tail = torch.empty(B, 3, e, Hs, Ws, dtype=torch.float32, device=z.device)
            ops_.append(dist.P2POp(dist.irecv, prev_tail, _global_rank(team, recv_from), group=team.group))
        if ops_:
            for w_ in dist.batch_isend_irecv(ops_):
                w_.wait()
    out = []
    for li in mine:                                                   # vae.rs:2410-2434
        tile = row[li]
        if li > 0:
            a = row[li - 1] if (li - 1) in row else prev_tail        # the blend reads only a's last min(len, blend) frames
            bl = blend(a, tile, blend_t, 2)
            out.append(bl[:, :, :min(stride_t, bl.shape[2])])
        else:
            out.append(tile[:, :, :min(stride_t + 1, tile.shape[2])])
    Hs, Ws = H * tl.spatial_compression_ratio, W * tl.spatial_compression_ratio
    chunk = torch.cat(out, 2) if out else torch.empty(B, 3, 0, Hs, Ws, dtype=torch.float32, device=z.device)
    if team.size == 1:
        return chunk[:, :, :num_sample_frames]
    # kept frames per rank are known everywhere: gather equal-size (padded) chunks, one frame travels once
    kept = []
    for r in range(team.size):
        n = 0
        for li in owner[r]:
            n += min(stride_t + 1, frames_of(li)) if li == 0 else min(stride_t, frames_of(li))
        kept.append(n)
    mx = max(kept)
    send = torch.zeros(B, 3, mx, Hs, Ws, dtype=torch.float32, device=z.device)
    send[:, :, :chunk.shape[2]] = chunk
    gathered = _all_gather(send, team)
    return torch.cat([gathered[r][:, :, :kept[r]] for r in range(team.size)], 2)[:, :, :num_sample_frames]


def _global_rank(team: Team, team_rank: int) -> int:
    return dist.get_global_rank(team.group, team_rank) if team.group is not None else team_rank


def decode_tile_sharded(decode_tile: Callable[[torch.Tensor], torch.Tensor], blend, z: torch.Tensor, tl: Tiling,
                        team: Team) -> torch.Tensor:
    """Tiled VAE decode split over the team.

    decode_tile(z_crop [B,C,t,h,w]) -> f32 [B,3,8t-7,32h,32w] (one `decoder.forward`);
    blend(a, b, extent, dim) -> b blended against a (blend_t/v/h, vae.rs:1927-2006).
    Every rank returns the full video; the result is bit-identical to the single-rank tiled decode (a noise-injecting decoder
    excepted: each rank's handle continues its own plane stream, ltx_vae_set_noise_seed).
    Framewise decoding (the 52-leaf case at C2) distributes whole temporal tiles and gathers only finished frames
    (_decode_temporal_sharded); the spatial-only mode has four leaves at C2, which are dealt round-robin and gathered whole."""
    _, _, F, H, W = z.shape
    if _mode(tl, F, H, W) == "temporal":
        return _decode_temporal_sharded(decode_tile, blend, z, tl, team)
    crops = leaf_crops(tl, F, H, W)
    r, tr = tl.spatial_compression_ratio, tl.temporal_compression_ratio
    shapes = [((c[1] - c[0] - 1) * tr + 1, (c[3] - c[2]) * r, (c[5] - c[4]) * r) for c in crops]
    mine = [i for i in range(len(crops)) if i % team.size == team.rank]
    dec = {}
    for i in mine:
        t0, t1, h0, h1, w0, w1 = crops[i]
        dec[i] = decode_tile(z[:, :, t0:t1, h0:h1, w0:w1].contiguous()).float()
    if team.size > 1:
        per = math.ceil(len(crops) / team.size)
        mt, mh, mw = (max(s[k] for s in shapes) for k in range(3))
        B = z.shape[0]
        send = torch.zeros(per, B, 3, mt, mh, mw, dtype=torch.float32, device=z.device)
        for k, i in enumerate(mine):
            st, sh, sw = shapes[i]
            send[k, :, :, :st, :sh, :sw] = dec[i]
        gathered = _all_gather(send, team)
        for rk in range(team.size):
            for k, i in enumerate(range(rk, len(crops), team.size)):
                if rk != team.rank:
                    st, sh, sw = shapes[i]
                    dec[i] = gathered[rk][k, :, :, :st, :sh, :sw].contiguous()
    return assemble(tl, F, H, W, [dec[i] for i in range(len(crops))], blend)


# ------------------------------------------------------------------ libltxhip bindings
class HipOps:
    """Binds the orchestration above to libltxhip objects (GPU only; importing this class's methods loads the library)."""

    def __init__(self, transformer, vae=None):
        self.transformer, self.vae = transformer, vae

    def forward_branch_fn(self, prompt_embeds, prompt_mask, neg_embeds, neg_mask, F, H, W, coords, skip_blocks, num_layers):
        def fn(name: str, latents: torch.Tensor, t: int) -> torch.Tensor:
            B = latents.shape[0]
            tv = [float(t)] * B
            if name == BRANCH_UNCOND:
                return self.transformer.forward(latents, neg_embeds, tv, neg_mask, F, H, W, video_coords=coords)
            slm = None
            if name == BRANCH_PERTURBED:                               # t2v_pipeline.rs:911-923
                slm = torch.zeros(num_layers, B)
                for li in skip_blocks or []:
                    if 0 <= li < num_layers:
                        slm[li] = 1.0
            return self.transformer.forward(latents, prompt_embeds, tv, prompt_mask, F, H, W, video_coords=coords, skip_layer_mask=slm)
        return fn

    @staticmethod
    def guidance_step_fn(guidance_scale: float, guidance_rescale: float, stg_scale: float):
        import ctypes as C
        from . import lib, _check, _ptr, _stream, LTX_F32

        def fn(preds: dict, latents: torch.Tensor, dt: float) -> torch.Tensor:
            text, unc, pert = preds[BRANCH_TEXT], preds.get(BRANCH_UNCOND), preds.get(BRANCH_PERTURBED)
            B = latents.shape[0]
            ws = torch.zeros(8 * B, dtype=torch.float64, device=latents.device)
            _check(lib.ltx_guidance_step(_ptr(text), _ptr(unc), _ptr(pert), LTX_F32, _ptr(latents), None, B,
                                         C.c_int64(latents[0].numel()), C.c_float(guidance_scale), C.c_float(guidance_rescale),
                                         C.c_float(stg_scale), C.c_float(dt), _ptr(ws), _stream()))
            return latents
        return fn

    def decode_tile_fn(self, timestep: Optional[float]):
        def fn(zc: torch.Tensor) -> torch.Tensor:
            keep = (self.vae.use_tiling, self.vae.use_framewise_decoding)
            self.vae.use_tiling = self.vae.use_framewise_decoding = False       # a leaf tile is ONE decoder.forward
            try:
                return self.vae.decode(zc, [timestep] * zc.shape[0] if timestep is not None else None)
            finally:
                self.vae.use_tiling, self.vae.use_framewise_decoding = keep

        def many(crops):
            """leaves of one latent shape stacked along the batch axis (one decoder call per group of <= 8: no op of the
            decoder crosses samples, so every leaf gets what it gets alone up to the f32 summation order of split-K convs)"""
            out = [None] * len(crops)
            groups = {}
            for i, c in enumerate(crops):
                groups.setdefault(tuple(c.shape), []).append(i)
            for idxs in groups.values():
                nb = crops[idxs[0]].shape[0]
                step = max(1, 8 // nb)
                for k in range(0, len(idxs), step):
                    part = idxs[k:k + step]
                    dec = fn(torch.cat([crops[i] for i in part], 0))
                    for j, i in enumerate(part):
                        out[i] = dec[j * nb:(j + 1) * nb]
            return out
        fn.many = many
        return fn

    @staticmethod
    def blend(a: torch.Tensor, b: torch.Tensor, extent: int, dim: int) -> torch.Tensor:
        from . import ops
        return ops.blend(a, b, dim, extent)


class ShardedLtxPipeline:
    """`LtxPipeline::call` for a TEAM of GPUs working on one video: guidance branches split across the team during the
    denoise loop, VAE leaf tiles split across the team during decode (when tiling is on).  Mirrors `ltxhip.LtxPipeline.call`
    (same arguments, same result on every rank); with a team of one it reproduces the single-GPU trajectory."""

    def __init__(self, transformer, vae, team: Optional[Team] = None):
        self.transformer, self.vae = transformer, vae
        self.team = team or Team()
        self.ops = HipOps(transformer, vae)

    def call(self, args, latents, prompt_embeds, prompt_attention_mask, negative_prompt_embeds=None,
             negative_prompt_attention_mask=None, decode_noise=None):
        import ltxhip as L
        if args.height % 32 or args.width % 32:
            raise L.LtxError("`height` and `width` must be divisible by 32")
        do_cfg, do_stg = args.guidance_scale > 1.0, args.stg_scale > 0.0
        if do_cfg and (negative_prompt_embeds is None or negative_prompt_attention_mask is None):
            raise L.LtxError("classifier-free guidance needs negative embeddings and mask")
        dev = torch.device("cuda", torch.cuda.current_device())
        f32 = lambda x: None if x is None else x.to(dev, torch.float32).contiguous()
        lat = f32(latents).clone()
        pe, pm, ne, nm = f32(prompt_embeds), f32(prompt_attention_mask), f32(negative_prompt_embeds), f32(negative_prompt_attention_mask)
        cfgd = self.transformer.config
        tsr = self.vae.config.temporal_compression_ratio if self.vae is not None else 8
        spr = self.vae.config.spatial_compression_ratio if self.vae is not None else 32
        F, H, W = (args.num_frames - 1) // tsr + 1, args.height // spr, args.width // spr
        S, B = F * H * W, lat.shape[0]
        K = pe.shape[1]
        L._expect("latents", lat, (B, S, cfgd.in_channels))
        L._expect("prompt_embeds", pe, (B, K, cfgd.caption_channels))
        L._expect("prompt_attention_mask", pm, (B, K))
        if ne is not None: L._expect("negative_prompt_embeds", ne, (B, K, cfgd.caption_channels))
        if nm is not None: L._expect("negative_prompt_attention_mask", nm, (B, K))
        if decode_noise is not None: L._expect("decode_noise", decode_noise, (B, cfgd.in_channels, F, H, W))
        if args.skip_block_list is not None:                           # permanent iff STG is off (:691-697)
            self.transformer.set_skip_block_list([] if do_stg else args.skip_block_list)
        N = args.num_inference_steps
        if args.sigmas is not None:
            sig_in, mu = [float(s) for s in args.sigmas], 0.0
        else:
            one = torch.tensor(1.0, dtype=torch.float32)
            sig_in = [1.0] if N == 1 else [float(one + (one / N - one) * torch.tensor(float(i), dtype=torch.float32) / (N - 1)) for i in range(N)]
            mu = L.calculate_shift(S)
        sched = L.FlowMatchEulerDiscreteScheduler(1.0, args.shift_terminal)
        timesteps = sched.set_timesteps(sig_in, mu)
        coords = L.build_video_coords(F, H, W, args.frame_rate, tsr, spr)[None].repeat(B, 1, 1).to(dev)
        branches = guidance_branches(args.guidance_scale, args.stg_scale)
        fwd = self.ops.forward_branch_fn(pe, pm, ne, nm, F, H, W, coords, args.skip_block_list, cfgd.num_layers)
        gstep = self.ops.guidance_step_fn(args.guidance_scale, args.guidance_rescale, args.stg_scale)
        self.transformer.context_cache(True)
        try:
            lat = denoise_branch_sharded(fwd, gstep, lat, branches, sched.sigmas, timesteps, self.team)
        finally:
            self.transformer.context_cache(False)
        if args.output_latent:
            return lat, None
        tc = self.vae.config.timestep_conditioning
        dn = f32(decode_noise) if (tc and decode_noise is not None) else None
        z = self.vae.prepare_latents(lat, F, H, W, dn, [args.decode_noise_scale] * B if dn is not None else None)
        tl = Tiling(self.vae.use_tiling, self.vae.use_framewise_decoding, self.vae.tile_sample_min_height, self.vae.tile_sample_min_width,
                    self.vae.tile_sample_min_num_frames, self.vae.tile_sample_stride_height, self.vae.tile_sample_stride_width,
                    self.vae.tile_sample_stride_num_frames, spr, tsr)
        video = decode_tile_sharded(self.ops.decode_tile_fn(args.decode_timestep if tc else None), HipOps.blend, z, tl, self.team)
        if args.postprocess:                                           # LtxVideoProcessor::postprocess_video (:146-155)
            video = ((video * 0.5 + 0.5).clamp(0.0, 1.0) * 255.0)
        return lat, video
