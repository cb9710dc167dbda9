// Large-tile bf16 MFMA GEMM / implicit-GEMM conv3d for gfx950 (the production path of the DiT
// Linears and the VAE convs; gemm.hip's 128x128 register-staged kernel covers f32 and small shapes).
//
//   * tile BM x BN (256/192/128 x 256/128), K-step 64 bf16 (128-B LDS rows), 512 threads =
//     8 waves as 2(M) x 4(N), each wave (BM/2) x (BN/4) with v_mfma_f32_16x16x32_bf16;
//   * operands go HBM -> LDS with buffer_load_dwordx4 ... lds (no VGPR round trip): one wave
//     instruction fills 8 rows x 128 B, LDS destination is lane-linear, so the bank swizzle
//     (16-B chunk ^ ((row>>1)&7)) is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 fragment reads (same involution on both sides);
//   * two LDS stages (<= 128 KiB): the loads of K-step t+1 are in flight while step t is
//     multiplied; one vmcnt(0)+barrier per K-step;
//   * conv mode gathers the A rows per tap from the channels-last activation: replicate padding
//     on T is a clamp, zero padding on H/W (and K tails) is an out-of-range buffer offset (reads zeros);
//   * D = Wfrag x Afrag so a lane owns 4 consecutive output columns (shared fused epilogues).
#include <cstring>
#include <map>
#include <mutex>
#include <atomic>
#include "gemm_common.h"
#include "options.h"

#ifndef GEMM_LOADERS
#define GEMM_LOADERS 4   // 8: every wave issues its share of the LDS-DMA pieces; 4: the first wave of each SIMD issues them all
#endif
#ifndef GEMM_PRIO
#define GEMM_PRIO 0      // experiment: static s_setprio 1 for the non-loader waves (1: -2...-6 %) or the loader waves (2: +-1 %)
#endif
#ifndef GEMM_ABL
#define GEMM_ABL 0   // timing ablations (wrong results): bit 0 no operand loads in the K loop, bit 1 no barrier, bit 2 loads never waited for, bit 3 loads all hit one 1-KiB line set, bit 4 no epilogue
#endif
namespace {

constexpr int ROWB = 128;          // bytes per LDS row (64 bf16)

__device__ __forceinline__ int swz_big(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];

template <int BM, int BN, int WGM, int WGN, int EPI, bool CONV, bool PIN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_big_kernel(const GemmArgs g) {
    constexpr int NW = WGM * WGN;                       // waves per block, laid out WGM (M) x WGN (N)
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
    // Loader waves.  All waves of a block queue at the CU's one load pipe (L2 -> LDS, ~110 GB/s per CU) when they issue
    // their LDS-DMA pieces, and an in-order wave issues no MFMA while it waits there.  With GEMM_LOADERS = 4 only the
    // first wave of every SIMD (waves 0..3) issues pieces, twice as many each; its SIMD partner (wave + 4) goes straight
    // to its MFMAs and keeps the matrix pipe busy meanwhile (measured with tools/ubench/dma_vs_mfma.hip: a partner
    // pushing a whole K-step's 64 KiB slows a wave's MFMA stream by 15 %).
    constexpr int NL = (GEMM_LOADERS == 4 && (NW == 8 || NW == 16)) ? 4 : NW;
    constexpr int AI = (BM + 8 * NL - 1) / (8 * NL), BI = BN / (8 * NL);   // glds instructions per loader wave per K-step (A, B)
    constexpr bool A_RAGGED = BM % (8 * NL) != 0;       // e.g. BM = 160: 20 eight-row pieces over 8 waves, the last round half empty
    static_assert(BM % 8 == 0 && BN % (8 * NL) == 0 && WM % 16 == 0 && WN % 16 == 0, "tile/wave layout");
    constexpr int STAGE = (BM + BN) * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lwave = NL == NW ? wave : (wave & (NL - 1));            // piece owner index (non-loader waves compute offsets they never use)
    const bool loader = NL == NW || __builtin_amdgcn_readfirstlane(wave) < NL;
    const int wm = wave / WGN, wn = wave % WGN;
    const int ntn = (g.N + BN - 1) / BN;
    // XCD-aware tile order (speed only, bijective for any grid): blocks b and b+8 share an XCD/L2, so XCD x is
    // given a CONTIGUOUS run of tiles -> neighbouring tiles (shared A rows / conv halos / W columns) hit one L2.
    // Tail split (g.sk_sf > 1): the first g.sk_full blocks each own a whole tile; the tiles of the last, partly filled
    // round are cut into sk_sf K-ranges ("parts") so that the round fills the chip; the parts of a tile meet in an
    // in-launch reduction (below).  The XCD remap covers the whole-tile region only.
    int bid = blockIdx.x;
    int part = 0;
    const bool split = g.sk_sf > 1 && bid >= g.sk_full;
    if (split) {
        const int p = bid - g.sk_full;
        bid = g.sk_full + p / g.sk_sf; part = p - (p / g.sk_sf) * g.sk_sf;
    } else if (g.xcd_remap) {
        const int nblk = g.sk_sf > 1 ? g.sk_full : (int)gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    // Order of the tiles inside a run: row-major, or columns of group_m row-tiles.  The blocks that run at the same time
    // on an XCD (32 CUs x blocks per CU, consecutive tile numbers) then cover a near-square patch of the output, and
    // the A rows + W columns they stream through that XCD's L2 shrink from (1 x C) to (group_m x C/group_m) panels.
    int mt, nt;
    if (g.group_m > 1) {
        const int ntm = (g.M + BM - 1) / BM, gsz = g.group_m * ntn;
        const int grp = bid / gsz, w = bid - grp * gsz, gm0 = grp * g.group_m;
        const int rows = ntm - gm0 < g.group_m ? ntm - gm0 : g.group_m;
        nt = w / rows; mt = gm0 + (w - nt * rows);
    } else { mt = bid / ntn; nt = bid - mt * ntn; }
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
    const int Kdim = g.K;
    const int ktiles = (Kdim + 63) / 64;
    const int ntaps = CONV ? g.ntaps : 1;
    const int nk = ktiles * ntaps;
    // per-lane staging geometry: instruction j of this wave fills LDS rows 8*(j*8+wave) .. +7;
    // lane -> (row in group = lane>>3, physical chunk = lane&7), logical chunk = pc ^ ((row>>1)&7).
    // Addressing is buffer-style (buffer_load_dwordx4 ... lds): a loop-invariant 32-bit byte offset per lane (row start
    // + chunk) plus a wave-uniform scalar offset (K position / tap), out-of-range offsets read as zeros (K tails, conv halo).
    constexpr uint32_t OOB = 0x80000000u;                 // >= num_records; ltx_gemm_big_fits keeps every addressed span < 2 GiB
    // conv mode: the descriptor is based a little before the tile's first voxel (two frames + one row + one voxel: every tap
    // of every row of the tile, replicate-clamped frames included, lies at a non-negative offset), so the 32-bit offsets span
    // the tile's own neighbourhood and the activation tensor may be any size (the 13B decode's last stages are 9 and 35 GB)
    const int64_t m_base = CONV ? (m0 > 2 * g.H * g.Wd + g.Wd + 1 ? (int64_t)m0 - (2 * g.H * g.Wd + g.Wd + 1) : 0) : 0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (CONV ? m_base * g.Cin : 0)), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)OOB, 0x00020000);
    const int lr = lane >> 3, pc = lane & 7;
    int a_chunk[AI], b_chunk[BI];
    uint32_t a_off[AI];                // byte offset of the row's chunk (conv: of the centre voxel)
    int ct[AI], vmask[AI];             // conv: frame index; 6 validity bits (h-1,h,h+1 | w-1,w,w+1 inside the image)
    uint32_t b_off[BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 8 * (j * NL + lwave) + lr;
        a_chunk[j] = pc ^ ((row >> 1) & 7);
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
        if constexpr (CONV) {
            const int w = m % g.Wd; const int t1 = m / g.Wd;
            const int h = t1 % g.H; const int t2 = t1 / g.H;
            ct[j] = t2 % g.T;
            vmask[j] = (h > 0 ? 1 : 0) | 2 | (h < g.H - 1 ? 4 : 0) | (w > 0 ? 8 : 0) | 16 | (w < g.Wd - 1 ? 32 : 0);
            a_off[j] = ((uint32_t)(m - m_base) * (uint32_t)g.Cin + a_chunk[j] * 8) * 2u;
        } else {
            a_off[j] = ((uint32_t)m * (uint32_t)g.lda + a_chunk[j] * 8) * 2u;
        }
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (j * NL + lwave) + lr;
        b_chunk[j] = pc ^ ((row >> 1) & 7);
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
        b_off[j] = ((uint32_t)n * (uint32_t)Kdim + b_chunk[j] * 8) * 2u;
    }
    const uint32_t frame_bytes = CONV ? (uint32_t)g.H * g.Wd * g.Cin * 2u : 0u;
    auto dma = [&](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, unsigned char* lds) {
#if GEMM_ABL & 8
        voff = (threadIdx.x & 63) * 16; soff = 0;       // every piece re-reads the same 1 KiB (always a cache hit)
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
    };

    // One K step's operand staging = AI + BI LDS-DMA pieces per wave.  stage_begin resolves the wave-uniform part (tap, K
    // offset), stage_piece issues piece idx; stage() issues them all back to back.
    struct StageCtx { unsigned char* As; unsigned char* Bs; int kk, dt, vbit; uint32_t a_soff, b_soff; bool more; };
    auto stage_begin = [&](int kt, int buf, bool more) {
        StageCtx c;
        c.As = big_smem + buf * STAGE; c.Bs = c.As + BM * ROWB; c.more = more;
        // K-step order of a conv (shared by every bf16 conv kernel, so plans stay bit-identical): frame tap, then the
        // 64-channel slice, then the kh x kw in-plane taps - consecutive steps re-read the same voxels shifted by one
        int tap = 0; c.kk = kt;
        int it = 0, hw = 0;
        if constexpr (CONV) {
            const int khw = g.kh * g.kw, per_it = ktiles * khw;
            it = kt / per_it; const int r2 = kt - it * per_it;
            c.kk = r2 / khw; hw = r2 - c.kk * khw;
            tap = it * khw + hw;
        }
        c.dt = 0; c.vbit = 0; c.a_soff = (uint32_t)c.kk * 128u;         // all wave-uniform (scalar) per tap
        if constexpr (CONV) {
            const int rem = hw;
            const int ih = rem / g.kw; const int iw = rem - ih * g.kw;
            const int dh = ih - g.kh / 2, dw = iw - g.kw / 2;
            c.dt = it - g.pad_t;
            c.vbit = (1 << (dh + 1)) | (8 << (dw + 1));
            c.a_soff += (uint32_t)((dh * g.Wd + dw) * g.Cin * 2);         // may be "negative": added modulo 2^32 to the lane offset
        }
        c.b_soff = ((uint32_t)tap * (uint32_t)g.N * (uint32_t)Kdim + (uint32_t)c.kk * 64u) * 2u;
        return c;
    };
    auto stage_piece = [&](const StageCtx& c, int idx) {
        if (idx < AI) {
            const int j = idx;
            bool ok = c.more && c.kk * 64 + a_chunk[j] * 8 < Kdim;
            if (A_RAGGED && (j * NL + lwave) * 8 >= BM) return;
            if constexpr (CONV) {
                int tt = ct[j] + c.dt; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);    // replicate pad on T (vae.rs:374-413)
                ok = ok && (vmask[j] & c.vbit) == c.vbit;                                    // zero pad on H/W (vae.rs:337-349)
                const uint32_t voff = a_off[j] + (uint32_t)(tt - ct[j]) * frame_bytes + c.a_soff;
                dma(ra, ok ? voff : OOB, 0u, c.As + (j * NL + lwave) * 1024);
            } else {
                dma(ra, ok ? a_off[j] : OOB, c.a_soff, c.As + (j * NL + lwave) * 1024);
            }
        } else {
            const int j = idx - AI;
            const bool ok = c.more && c.kk * 64 + b_chunk[j] * 8 < Kdim;
            dma(rw, ok ? b_off[j] : OOB, c.b_soff, c.Bs + (j * NL + lwave) * 1024);
        }
    };
    auto stage = [&](int kt, int buf) {
        if (!loader) return;
        const StageCtx c = stage_begin(kt, buf, true);
#pragma unroll
        for (int i = 0; i < AI + BI; ++i) stage_piece(c, i);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kt0 = split ? (int)((int64_t)part * nk / g.sk_sf) : 0;
    const int kt1 = split ? (int)((int64_t)(part + 1) * nk / g.sk_sf) : nk;
    stage(kt0, 0);
    __syncthreads();                                   // emits vmcnt(0) for the outstanding LDS-DMA

    const int frow = lane & 15, fq = lane >> 4;
#if GEMM_PRIO == 1
    if (!loader) __builtin_amdgcn_s_setprio(1);         // measured: -2...-6 %
#elif GEMM_PRIO == 2
    if (loader && NL != NW) __builtin_amdgcn_s_setprio(1);
#endif
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
#if !(GEMM_ABL & 1)
        if (kt + 1 < kt1) stage(kt + 1, buf ^ 1);
#endif
        const unsigned char* As = big_smem + buf * STAGE;
        const unsigned char* Bs = As + BM * ROWB;
        // Fragment stream: W fragments of the k-block stay in registers, A fragments are read ONE AHEAD of the
        // MFMAs that consume them (the pinned order below keeps hipcc from collapsing it back into
        // "read 2, drain lgkmcnt(0), 4 MFMAs", which exposed the LDS latency 6 times per K-step).
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Chunk16 af[FM], wf[FN];
#pragma unroll
            for (int f = 0; f < FN; ++f) wf[f].u = *reinterpret_cast<const u32x4*>(Bs + swz_big(wn * WN + f * 16 + frow, kb * 4 + fq));
            af[0].u = *reinterpret_cast<const u32x4*>(As + swz_big(wm * WM + frow, kb * 4 + fq));
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                if (fm + 1 < FM) af[fm + 1].u = *reinterpret_cast<const u32x4*>(As + swz_big(wm * WM + (fm + 1) * 16 + frow, kb * 4 + fq));
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = Mma<bf16_t>::run(wf[fn], af[fm], acc[fm][fn]);
            }
        }
        if constexpr (PIN) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                __builtin_amdgcn_sched_group_barrier(0x100, FN + 1, 0);            // DS read: W frags + first A frag
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    if (fm + 1 < FM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // next A frag
                    __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);            // MFMAs of the current A frag
                }
            }
        }
#if GEMM_ABL & 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif GEMM_ABL & 4
        __builtin_amdgcn_s_barrier();                  // loads issued but never waited for
#else
        __syncthreads();
#endif
    }

    if (split) {
        // In-launch reduction of a tile's parts: every part stores its f32 accumulators to its slab (fragment-major,
        // 16 B per lane: fully coalesced) and takes a ticket; the part that draws the last one acquires once, adds the other
        // slabs to its registers and runs the epilogue.  Round 4 experiment (LTX_GEMM_SPLIT_SC1=1, off by default): the slab as
        // WRITE-THROUGH (sc1) stores, drained with vmcnt(0) in every storing wave before the workgroup barrier, no agent-scope
        // release on the publishing side (MI355X guide, valid forms; publish-large row), the reader keeps its acquire and loads
        // sc1.  Measured on the small-M shapes (tools/small_m_probe.py, M = 384 / 128): within +-3 % of the fenced protocol on
        // seven shapes, 16 % slower on ff2 at M = 128 - the publish is not what these launches wait for (docs/lab_notes.md R4.8).
        constexpr int SLAB = BM * BN;                                  // floats
        const int tt = bid - g.sk_full;
        float* slab = g.sk_ws + ((int64_t)tt * g.sk_sf + part) * SLAB;
        // the tile's sk_sf slabs through one buffer descriptor (aux 16 = sc1 on gfx950: write-through stores, L1-bypassing loads)
        const __amdgpu_buffer_rsrc_t rslab = __builtin_amdgcn_make_buffer_rsrc(g.sk_ws + (int64_t)tt * g.sk_sf * SLAB, 0, g.sk_sf * SLAB * 4, 0x00020000);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int so = ((fm * FN + fn) * (64 * NW) + tid) * 16;          // byte offset inside the slab
                if (g.sk_plain) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(slab) + so) = acc[fm][fn];
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[fm][fn]), rslab, part * (SLAB * 4) + so, 0, 16 /* sc1: write-through */);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(big_smem);       // all LDS tile reads are behind the barrier above
        if (tid == 0) {
            if (g.sk_plain) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(g.sk_cnt + tt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)g.sk_sf - 1u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // leave the counter at zero for the next launch on this stream (the workspace is zeroed once, at allocation)
                __hip_atomic_store(g.sk_cnt + tt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;
        // canonical order ((s0 + s1) + s2) + ... whichever part happens to be the reducer: results do not depend on timing
        // (its own slab is re-read from memory when it is not part 0; two parts need no re-read, a + b == b + a)
        const float* base = g.sk_ws + (int64_t)tt * g.sk_sf * SLAB;
        const bool reread_own = g.sk_sf > 2 && part != 0;
        auto ld_slab = [&](const float* p) {
            if (g.sk_plain) return *reinterpret_cast<const f32x4*>(p);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rslab, (int)((p - base) * 4), 0, 16 /* sc1 */));
        };
        if (reread_own) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = ld_slab(base + ((fm * FN + fn) * (64 * NW) + tid) * 4);
        }
        for (int p = (reread_own || part == 0) ? 1 : 0; p < g.sk_sf; ++p) {
            if (p == part && !reread_own) continue;
            const float* other = base + (int64_t)p * SLAB;
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const f32x4 o = ld_slab(other + ((fm * FN + fn) * (64 * NW) + tid) * 4);
                    acc[fm][fn] += o;
                }
        }
    }

    // ---- wide epilogue (linear layers, bf16): the MFMA layout gives a lane 4 consecutive columns, i.e. 8-byte stores that
    // touch a 128-byte line from four different instructions (and from two waves where a wave spans 32 columns); the
    // epilogue of a one-round grid is not hidden behind other blocks' K loops, and measured 8-30 % of a DiT GEMM.  Here
    // the result tile goes through LDS (free after the K loop): every lane applies bias / GELU / gate / residual to its
    // own elements exactly as epilogue() does (the residual tile is brought into the same LDS image first, by LDS-DMA, so
    // its reads are row-contiguous too), writes bf16 in place, and the tile leaves as 16-byte row-contiguous stores.
    // 16-byte chunk c of tile row r sits at chunk c ^ (r & 15) (bank spread for the 8-byte fragment accesses; r & 7 for 64-wide tiles).
    // (measured per epilogue on the DiT shapes: bias +3..9 %, gate / residual +1..15 %, GELU -2..+2 % -> GELU keeps the direct stores)
    constexpr bool WIDE = !CONV && (EPI == EPI_BIAS || EPI == EPI_GATE_RESID || EPI == EPI_RESID);
    if constexpr (WIDE) {
        if (g.wide_epi) {
            constexpr int CPR = BN / 8;                                  // 16-byte chunks per tile row
            constexpr int XM = (CPR < 16 ? CPR : 16) - 1;                // chunk XOR mask
            constexpr int NPASS = (BM * BN * 2 + 2 * STAGE - 1) / (2 * STAGE);
            static_assert(NPASS == 1 || (NPASS == 2 && WGM == 2), "result tile must fit LDS in at most two row passes");
            constexpr int BMP = BM / NPASS;
            constexpr bool HAS_R = EPI == EPI_GATE_RESID || EPI == EPI_RESID;
            const int swave = __builtin_amdgcn_readfirstlane(wave);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.resid ? g.resid : g.C), 0, (int)OOB, 0x00020000);
            bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                const int prow0 = p * BMP;
                if constexpr (HAS_R) {
                    constexpr int RPP = 64 / CPR;                        // tile rows per 1-KiB piece
                    for (int pi = swave; pi < BMP / RPP; pi += NW) {
                        const int row = pi * RPP + lane / CPR, pc = lane % CPR;
                        const int lc = pc ^ (row & XM);
                        const int m = m0 + prow0 + row, n = n0 + lc * 8;
                        const bool ok = m < g.M && n < g.N;
                        dma(rr, ok ? (uint32_t)(((int64_t)m * g.ldr + n) * 2) : OOB, 0u, big_smem + pi * 1024);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (NPASS == 1 || wm == p) {
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) {
                        const int row = (NPASS == 1 ? wm * WM : 0) + fm * 16 + frow;
                        const int m = m0 + prow0 + row;
                        const int mc = m < g.M ? m : g.M - 1;
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn) {
                            const int col = wn * WN + fn * 16 + 4 * fq, nb = n0 + col;
                            unsigned char* slot = big_smem + row * (CPR * 16) + (((col >> 3) ^ (row & XM)) << 4) + ((col >> 2) & 1) * 8;
                            float v[4] = {acc[fm][fn][0], acc[fm][fn][1], acc[fm][fn][2], acc[fm][fn][3]};
                            if (nb < g.N) {
                                if (g.bias) {
                                    float b[4];
                                    load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + nb, b);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) v[i] += b[i];
                                }
                                if constexpr (EPI == EPI_GELU) {
                                    gelu_tanh4(v);
                                } else if constexpr (HAS_R) {
                                    float r[4];
                                    load4<bf16_t>(reinterpret_cast<const bf16_t*>(slot), r);
                                    if constexpr (EPI == EPI_GATE_RESID) {
                                        const f32x4 gt = *reinterpret_cast<const f32x4*>(g.gate + (int64_t)(mc / g.rows_per_batch) * g.gate_stride + nb);
#pragma unroll
                                        for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(gt[i], v[i], r[i]);       // ONE rounding, spelled out: every kernel that finishes these rows must agree
                                    } else {
#pragma unroll
                                        for (int i = 0; i < 4; ++i) v[i] += r[i];
                                    }
                                }
                            }
                            store4<bf16_t>(reinterpret_cast<bf16_t*>(slot), v);
                        }
                    }
                }
                __syncthreads();
                for (int id = tid; id < BMP * CPR; id += 64 * NW) {
                    const int row = id / CPR, c = id - row * CPR;
                    const int m = m0 + prow0 + row, n = n0 + c * 8;
                    if (m >= g.M || n >= g.N) continue;
                    const u32x4 d = *reinterpret_cast<const u32x4*>(big_smem + row * (CPR * 16) + ((c ^ (row & XM)) << 4));
                    bf16_t* dst = Cb;
                    int nc = n;
                    if (g.c_seg_shift) { const int sg = n >> g.c_seg_shift; dst += sg * g.c_seg_stride; nc -= sg << g.c_seg_shift; }
                    *reinterpret_cast<u32x4*>(dst + (int64_t)m * g.ldc + nc) = d;
                }
                if (p + 1 < NPASS) __syncthreads();
            }
            return;
        }
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = m0 + wm * WM + fm * 16 + frow;
        if (m >= g.M) continue;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int nb = n0 + wn * WN + fn * 16 + 4 * fq;
            if (nb >= g.N) continue;
            float v[4] = {acc[fm][fn][0], acc[fm][fn][1], acc[fm][fn][2], acc[fm][fn][3]};
#if GEMM_ABL & 16
            if (v[0] == 123.456f)                          // no epilogue (no residual loads, no stores)
#endif
            epilogue<bf16_t, EPI>(g, m, nb, v);
        }
    }
}

// ---- tail split-K planning + workspace ---------------------------------------------------------------------------
// slots = blocks the chip holds at once for this tile (LDS- and thread-limited).  T tiles = `full` whole rounds + a tail of
// `tail` tiles; when the tail would leave most of the chip idle, each tail tile is cut into sf K-ranges (parts).
struct SplitWs { void* slabs = nullptr; size_t slab_bytes = 0; void* cnt = nullptr; size_t cnt_bytes = 0; };
std::map<std::pair<int, hipStream_t>, SplitWs> g_split_ws;     // one workspace per (device, stream): launches on a stream are ordered
std::mutex g_split_mu;

}  // namespace

// The split factor is a function of the PROBLEM SHAPE only (never of the tile or of a measured plan), and every plan a
// split shape may run is a gemm_big tile: the K-ranges [part * nk / sf, (part + 1) * nk / sf) and the canonical sum of the
// parts are then the same for every plan, so results do not depend on which plan a process happened to measure as fastest
// (VERDICT r1 / ADVICE r1: sf used to follow the tile).  Rule: outputs that cannot half-fill the chip with 256 x 256 tiles
// (M*N*2 <= 256 CUs * 256 * 256) are cut into floor(chip / outputs) <= 8 K-ranges of at least 8 K-steps each.
// Measured on MI355X: cutting the thin LAST round of a multi-round grid loses (qkv/ff1 -4..-16 %); a grid that cannot even
// half-fill the chip gains (VAE mid-block conv, 27648-deep K: +11..16 %).
int ltx_gemm_split_factor(const GemmArgs& g) {
    if (!ltx_opt().gemm_splitk) return 1;
    const int nk = (g.K + 63) / 64 * (g.conv ? g.ntaps : 1);
    if (const int sf = ltx_exp("gemm_split_force", 0)) {            // measurement aid (tools/ring_probe.py): this many K-ranges for every linear layer
        if (!g.conv && sf >= 1 && sf <= 8 && nk / sf >= 1) return sf;
    }
    const int max_sf = ltx_exp("gemm_split_max", 8), min_k = ltx_exp("gemm_split_mink", 8);
    // Linear layers of at most 512 rows (round 4, gemm_ring.hip): never split.  On the deep-ring tiles a K-range costs 0.2-0.3 us
    // per step where the two-stage tiles paid 0.5, and a grid of 96 x 32 / 64 x 32 tiles fills the chip without cutting K, while
    // the in-launch reduction costs 5 us for two parts and 8 for four (slab store, agent-scope release + ticket + acquire, the
    // last arriver re-reading the slabs: tools/ring_trace.py).  Measured over 14 shapes with M = 128 .. 512
    // (tools/ring_split_probe.py, profiles/r4_ring_split_probe.jsonl): unsplit wins 12, ties one, loses 12 % on one (T5's wo,
    // K = 10240).  A function of (M, N, K) alone, like the rest of this rule.
    // Round 5: ... except the deepest ones (K >= 8192: the DiT's ff2 at C1's 384 tokens, T5-XXL's wo), cut into four ranges.  The
    // DiT hands the ranges to the row norm that follows the layer (GemmArgs::defer_parts: no in-launch reduction at all, a block
    // runs 32 K-steps instead of 128: ff2 33 -> ~20 us); a stand-alone call pays the in-launch reduction (ff2 +3 us, T5's wo -5 us:
    // profiles/r4_ring_split_probe.jsonl).
    if (!g.conv && g.M <= 512 && g.N >= 32 && g.N % 4 == 0 && g.K % 8 == 0 && ltx_exp("gemm_split_ring", 1)) return nk >= 128 && nk % 4 == 0 ? 4 : 1;
    if (!g.conv && g.M <= 1536 && ltx_exp("gemm_split_smallm", 1)) {
        // Small-M linear layers (C1's 384 tokens, the 128 text rows; round 3, tools/small_m_probe.py): these are latency-bound
        // weight streams - a K-step costs 0.4-0.75 us whatever it computes - and the in-launch reduction grows faster than
        // linearly with the parts (1.4 / 6 / 22 us for 2 / 4 / 8), so split only until ~256 blocks are in flight (512 when K is
        // long), keep >= 16 K-steps per part, and at most 4 parts once there are more than 16 tiles.  The area rule below cut
        // qkv at M = 384 (144 tiles of 128 x 128) into 4 parts: 36 us against 25 unsplit; M = 1152: 55 against 31.
        const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
        const int t0 = ltx_exp("gemm_split_small_target", 256), mk = ltx_exp("gemm_split_small_mink", 16), mx = ltx_exp("gemm_split_small_max", 4);
        const int target = nk >= 96 ? 2 * t0 : t0;
        int sf = 1;
        while (sf * 2 <= 8 && tiles * sf * 2 <= target && nk / (sf * 2) >= mk) sf *= 2;
        if (tiles > 16 && sf > mx) sf = mx;
        return sf;
    }
    const double area = (double)g.M * (double)g.N, chip = 256.0 * 256.0 * 256.0;
    if (area * 2.0 > chip) return 1;
    int sf = (int)(chip / area);
    if (sf > max_sf) sf = max_sf;
    while (sf > 1 && nk / sf < min_k) --sf;            // every part keeps at least 8 K-steps
    return sf < 2 ? 1 : sf;
}

namespace {
int plan_tail_split(GemmArgs* g, int tiles, int bm, int bn, int threads, int smem, hipStream_t s) {
    (void)threads; (void)smem;
    g->sk_sf = 1; g->sk_full = tiles;
    const int sf = ltx_gemm_split_factor(*g);
    if (sf < 2) return LTX_OK;
    const int full = 0, tail = tiles;
    const size_t slab_bytes = (size_t)tail * sf * bm * bn * sizeof(float), cnt_bytes = (size_t)tail * sizeof(unsigned);
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_split_mu);
    SplitWs& w = g_split_ws[std::make_pair(dev, s)];
    if (w.slab_bytes < slab_bytes) { if (w.slabs) (void)hipFree(w.slabs); w.slabs = nullptr; HIP_TRY(hipMalloc(&w.slabs, slab_bytes)); w.slab_bytes = slab_bytes; }
    if (w.cnt_bytes < cnt_bytes) {
        const size_t nb = cnt_bytes < 4096 ? 4096 : cnt_bytes;
        if (w.cnt) (void)hipFree(w.cnt);
        w.cnt = nullptr; HIP_TRY(hipMalloc(&w.cnt, nb)); w.cnt_bytes = nb;
        HIP_TRY(hipMemsetAsync(w.cnt, 0, nb, s));      // once: every reducer hands its counter back at zero
    }
    g->sk_sf = sf; g->sk_full = full; g->sk_ws = reinterpret_cast<float*>(w.slabs); g->sk_cnt = reinterpret_cast<unsigned*>(w.cnt);
    g->sk_plain = !ltx_exp("gemm_split_sc1", 0);          // default: round 3's protocol (see the kernel)
    return LTX_OK;
}
}  // namespace
int ltx_gemm_split_workspace(GemmArgs* g, int tiles, int bm, int bn, hipStream_t s) { return plan_tail_split(g, tiles, bm, bn, 0, 0, s); }
namespace {

template <int BM, int BN, int WGM, int WGN, int EPI, bool CONV>
int launch_one(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * ROWB;
    static std::atomic<unsigned long long> attr_devs{0};
    // PIN = true: the pinned fragment-stream schedule (A/B on MI355X: linear +-1 %, conv +1..4 % over the compiler's order)
    auto kern = gemm_big_kernel<BM, BN, WGM, WGN, EPI, CONV, true>;
    LTX_TRY(ltx_set_max_dyn_smem(attr_devs, reinterpret_cast<const void*>(kern), smem));
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    GemmArgs ga = g;
    LTX_TRY(plan_tail_split(&ga, tiles, BM, BN, 64 * WGM * WGN, smem, s));
    {   // tile order inside an XCD's run: the C blocks one XCD runs at once (32 CUs x blocks per CU) should cover a
        // patch of pm x pn tiles minimising pm*BM + pn*BN, i.e. pm = sqrt(C*BN/BM)
        int per_cu = (160 * 1024) / smem; if (per_cu > 2048 / (64 * WGM * WGN)) per_cu = 2048 / (64 * WGM * WGN); if (per_cu < 1) per_cu = 1;
        const int C = 32 * per_cu;
        int gm = 1; while ((gm + 1) * (gm + 1) * BM <= C * BN) ++gm;
        { const int x_gm = ltx_exp("gemm_group_m", -1); if (x_gm >= 0) gm = x_gm; }   // tuning aid: 0/1 row-major
        const int ntm = cdiv(g.M, BM);
        if (gm > ntm) gm = ntm;
        ga.group_m = (CONV || gm < 2) ? 0 : gm;
    }
    {   // wide epilogue: 16-byte chunks need 8-column granularity and 16-byte aligned rows
        const bool seg_ok = !g.c_seg_shift || ((1 << g.c_seg_shift) % 8 == 0 && g.c_seg_stride % 8 == 0);
        ga.wide_epi = ltx_opt().gemm_wide_epi && !CONV && g.N % 8 == 0 && g.ldc % 8 == 0 && ((uintptr_t)g.C & 15) == 0 && seg_ok &&
                      (!g.resid || (g.ldr % 8 == 0 && ((uintptr_t)g.resid & 15) == 0)) && (double)g.M * g.ldr * 2.0 < 2147483648.0;
    }
    dim3 grid((unsigned)(ga.sk_sf > 1 ? ga.sk_full + (tiles - ga.sk_full) * ga.sk_sf : tiles)), block(64 * WGM * WGN);
    LTX_LAUNCH_TIMED(kern, grid, block, smem, s, ga);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

template <int BM, int BN, int WGM, int WGN, bool CONV>
int launch_epi(const GemmArgs& g, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_one<BM, BN, WGM, WGN, EPI_BIAS, CONV>(g, s);
        case EPI_GELU: return launch_one<BM, BN, WGM, WGN, EPI_GELU, CONV>(g, s);
        case EPI_GATE_RESID: return launch_one<BM, BN, WGM, WGN, EPI_GATE_RESID, CONV>(g, s);
        case EPI_RESID: return launch_one<BM, BN, WGM, WGN, EPI_RESID, CONV>(g, s);
        case EPI_D2S: if constexpr (CONV) return launch_one<BM, BN, WGM, WGN, EPI_D2S, CONV>(g, s); break;
        case EPI_UNPATCH: if constexpr (CONV) return launch_one<BM, BN, WGM, WGN, EPI_UNPATCH, CONV>(g, s); break;
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_big: bad epilogue");
}

// tile ids: see kTiles below
template <bool CONV>
int launch_tile(const GemmArgs& g, int epi, int tile, hipStream_t s) {
    switch (tile) {
        case 0: return launch_epi<256, 256, 2, 4, CONV>(g, epi, s);
        case 1: return launch_epi<192, 256, 2, 4, CONV>(g, epi, s);
        case 2: return launch_epi<128, 256, 2, 4, CONV>(g, epi, s);
        case 3: return launch_epi<256, 128, 2, 4, CONV>(g, epi, s);
        case 4: return launch_epi<192, 128, 2, 4, CONV>(g, epi, s);
        case 5: return launch_epi<128, 128, 2, 4, CONV>(g, epi, s);
        case 6: return launch_epi<160, 128, 2, 4, CONV>(g, epi, s);     // M = 4992 = 31.2 x 160: 32 x (N/128) tiles = whole rounds of 512
        case 7: return launch_epi<192, 64, 4, 2, CONV>(g, epi, s);      // narrow outputs (conv_out N = 48, proj_out): 4(M) x 2(N) waves of 48x32
        // 16-wave blocks = two of the 2-per-CU tiles side by side sharing ONE activation tile in LDS (same wave shapes, same
        // scheduling granularity, 28-30 % fewer operand bytes through the CU's load pipe per MFMA)
        case 8: return launch_epi<160, 256, 2, 8, CONV>(g, epi, s);
        case 9: return launch_epi<192, 256, 2, 8, CONV>(g, epi, s);
        case 10: return launch_epi<320, 256, 2, 8, CONV>(g, epi, s);    // M = 4992 = 15.6 x 320: ff1 (N = 8192) = 512 tiles = two full rounds
        case 11: return launch_epi<256, 256, 4, 4, CONV>(g, epi, s);
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_big: unsupported tile");
}

struct TileInfo { int bm, bn, threads; double rate; int per_cu; const char* name; };
// rate: per-CU sustained TFLOP/s of the tile at full occupancy (tools/microbench.py tiles, MI355X, random data);
// per_cu: blocks that fit one CU (LDS 2*(BM+BN)*128 B of 160 KiB, <= 2048 threads)
const TileInfo kTiles[] = {
    {256, 256, 512, 1160, 1, "256x256"}, {192, 256, 512, 1190, 1, "192x256"}, {128, 256, 512, 1023, 1, "128x256"},
    {256, 128, 512, 989, 1, "256x128"},  {192, 128, 512, 1400, 2, "192x128"}, {128, 128, 512, 1032, 2, "128x128"},
    {160, 128, 512, 1330, 2, "160x128"}, {192, 64, 512, 900, 2, "192x64"},
    {160, 256, 1024, 1300, 1, "160x256w16"}, {192, 256, 1024, 1300, 1, "192x256w16"},
    {320, 256, 1024, 1300, 1, "320x256w16"}, {256, 256, 1024, 1300, 1, "256x256w16"},
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
// a tile is a candidate for an N-wide output: 64-wide tiles only for narrow outputs, 256-wide only beyond 128 columns
bool tile_fits(const TileInfo& t, int N) { return t.rate > 0 && !(t.bn > 128 && N <= 128) && !(t.bn == 64 && N > 64) && !(t.bn > 64 && N <= 64); }

}  // namespace

// Tile choice: minimise (rounds over the CUs) x (tile time); ties go to the larger tile (less operand re-reading).
int ltx_gemm_big_pick_tile(int M, int N) {
    const char* force = ltx_opt().gemm_plan;           // a gemm_big tile named by the gemm_plan option, e.g. "256x128" (tuning / test aid)
    if (force[0]) for (int i = 0; i < kNumTiles; ++i) if (!strcmp(force, kTiles[i].name)) return i;
    double best = 1e30; int bi = 4;
    for (int i = 0; i < kNumTiles; ++i) {
        const TileInfo& t = kTiles[i];
        if (!tile_fits(t, N)) continue;
        const int64_t tiles = (int64_t)cdiv(M, t.bm) * cdiv(N, t.bn);
        const double cost = (double)cdiv64(tiles, 256 * t.per_cu) * (double)(t.bm * t.bn * t.per_cu) / t.rate;
        if (cost < best * 0.999) { best = cost; bi = i; }
    }
    return bi;
}

// 32-bit buffer offsets (0x80000000 = out of range).  Linear layers address whole operands; a conv tile addresses a window of
// the activation around itself (gemm_big) or one frame (conv_halo), so only that window has to stay below 2 GiB.
bool ltx_gemm_big_fits(const GemmArgs& g) {
    const double w_bytes = (double)(g.conv ? g.ntaps : 1) * g.N * g.K * 2.0;
    if (w_bytes >= 2147483648.0) return false;
    if (!g.conv) return (double)g.M * g.lda * 2.0 < 2147483648.0;
    const double window_rows = 3.0 * g.H * g.Wd + 2.0 * g.Wd + 512.0;      // two frames back, one forward, the tile itself
    return window_rows * g.Cin * 2.0 < 2147483648.0 && (double)g.B * g.T * g.H * g.Wd < 2147483648.0;
}

bool ltx_gemm_big_eligible(const GemmArgs& g, int dtype) {
    if (dtype != LTX_DT_BF16) return false;
    if (ltx_opt().gemm_off & LTX_FAM_BIG) return false;
    if (g.conv && (g.kh > 3 || g.kw > 3)) return false;   // the validity mask covers 3x3 (and 1x1) spatial taps
    if (!ltx_gemm_big_fits(g)) return false;              // 32-bit buffer offsets: every addressed span < 2 GiB (else gemm.hip's kernel)
    // Linear layers of any M take the 128-row tiles with the shape-only split-K (small outputs: up to 8 K-ranges per tile, so
    // the weight matrix streams through every CU): context k/v and caption projections (M = 128), the timestep MLPs
    // (M = 1) and the T5 encoder's M = 128 GEMMs run 1.5-2.5x faster than on gemm.hip's 128 x 128 register-staged kernel
    // (T5-XXL at 128 tokens 14.9 -> 10.5 ms).  Convs from 256 output voxels up take them too since round 3 (C1's mid block is
    // 4 x 8 x 12 = 384 voxels of 1024 channels: 11 convs of 57 MB of weights each, 500 us apiece on gemm.hip's kernel, 6.5 of
    // C1's 63 ms; with the split-K tiles the decode went 11.9 -> 6.3 ms); smaller ones stay there.  (experiment builds:
    // x_gemm_big_minm / x_gemm_big_conv_minm override.)
    int min_m = g.conv ? 16 : 1;      // (round 4: the edge tiles of the tiled decode are convs of 48..192 voxels x 1024 channels - 57 MB of weights at 0.11 TB/s on the 128 x 128 kernel, 500 us apiece)
    { const int x = ltx_exp("gemm_big_minm", -1); if (x >= 0) { min_m = x; if (g.conv && min_m < 1024) min_m = 1024; } }
    if (g.conv) { const int x = ltx_exp("gemm_big_conv_minm", -1); if (x >= 0) min_m = x; }
    return g.M >= min_m && g.N >= 32;
}

// ---- plan selection --------------------------------------------------------------------------------------------
// A plan is a gemm_big tile id (0 .. kNumTiles-1) or kPlanP8 + {0: BN=256, 1: BN=128} (gemm_p8.hip).  Every plan
// accumulates K in the same order with the same MFMA, so the choice changes speed only, never a bit of the result.
// Default: measure once per problem shape (first call: each candidate runs into a scratch output, best of the timed
// launches wins, cached for the life of the process).  LTX_GEMM_TUNE=0 falls back to the static cost model;
// LTX_GEMM_TILE / LTX_GEMM_P8 force a plan (tests, A/B runs).
namespace {
constexpr int kPlanP8 = 100;
constexpr int kPlanHalo = 200;           // + {0: BN=128, 1: BN=256} (conv_halo.hip)
constexpr int kPlanAsm16 = 300;          // + {0: 256 x 256, 1: 160 x 256, 2: 320 x 256} (gemm_asm.hip, the 16x16x32 one-wave-per-SIMD kernel; linear layers)
constexpr int kPlanAsm16Conv = 350;      // gemm_asm.hip's loop in conv mode, tile 256 x 256 (3x3x3 convs of many channels: the VAE mid block; split shapes too: it keeps gemm_big's K partition)
constexpr int kPlanRing = 400;           // + tile index of gemm_ring.hip (small-M linear layers; split shapes too: it keeps gemm_big's K partition)
struct PlanKey {
    int M, N, K, conv, ntaps, T, H, W;
    bool operator<(const PlanKey& o) const { return memcmp(this, &o, sizeof(PlanKey)) < 0; }
};
std::map<PlanKey, int> g_plans;
std::mutex g_plan_mu;

// The predicates tune_plan applies before it measures a plan: a cached / loaded plan must pass them again for the shape
// it is used on (a stale or hand-edited plan file, or one saved under other LTX_* settings).
bool plan_shape_ok(int plan, int N, int nk, bool split_shape) {
    if (plan >= kPlanRing) return plan < kPlanRing + ltx_gemm_ring_tiles() && N >= 32 && N % 4 == 0;
    if (plan == kPlanAsm16Conv) return nk >= 2 && N >= 1024 && N % 8 == 0;
    if (plan >= kPlanAsm16) return plan <= kPlanAsm16 + 2 && !split_shape && nk >= 2 && N >= 512 && N % 8 == 0;
    if (plan >= kPlanHalo) return plan <= kPlanHalo + 1 && !split_shape;
    if (plan >= kPlanP8) return plan <= kPlanP8 + 1 && !split_shape && nk >= 2 && N > 64 && !(plan == kPlanP8 && N <= 128);
    return plan >= 0 && plan < kNumTiles && tile_fits(kTiles[plan], N);
}
bool plan_ok(const GemmArgs& g, int epi, int plan) {
    const int nk = (g.K + 63) / 64 * (g.conv ? g.ntaps : 1);
    if (!plan_shape_ok(plan, g.N, nk, ltx_gemm_split_factor(g) > 1)) return false;
    if (plan >= kPlanRing) return ltx_gemm_ring_tile_fits(g, epi, plan - kPlanRing);
    if (plan == kPlanAsm16Conv) return ltx_gemm_asm16_conv_fits(g, epi);
    if (plan >= kPlanAsm16) return ltx_gemm_asm16_fits(g, epi);
    if (plan >= kPlanHalo) return true;                    // run_plan checks the halo kernel's own eligibility (epilogue-dependent)
    if (plan >= kPlanP8) return ltx_gemm_p8_fits(g);
    (void)epi;
    return true;
}

int run_plan(const GemmArgs& g, int epi, int plan, hipStream_t s) {
    if (!plan_ok(g, epi, plan)) plan = ltx_gemm_big_pick_tile(g.M, g.N);
    if (plan >= kPlanRing) return ltx_launch_gemm_ring(g, epi, plan - kPlanRing, s);
    if (plan == kPlanAsm16Conv) return ltx_launch_gemm_asm16_conv(g, epi, s);
    if (plan >= kPlanAsm16) return ltx_launch_gemm_asm16(g, epi, plan - kPlanAsm16, s);
    if (plan >= kPlanHalo) {
        const int bn = plan == kPlanHalo ? 128 : 256;
        if (ltx_conv_halo_eligible(g, epi, bn)) return ltx_launch_conv_halo(g, epi, bn, s);
        plan = ltx_gemm_big_pick_tile(g.M, g.N);           // an epilogue the halo kernel does not carry
    }
    if (plan >= kPlanP8) return ltx_launch_gemm_p8(g, epi, plan == kPlanP8 ? 256 : 128, s);
    ltx_prof_kernel(LTX_PROFK_GEMM_BIG);
    return g.conv ? launch_tile<true>(g, epi, plan, s) : launch_tile<false>(g, epi, plan, s);
}

int tune_plan(const GemmArgs& g_in, int epi, hipStream_t s, int fallback, int* plan_out) {
    *plan_out = fallback;
    // plain bias epilogue into a scratch [M, N] output - except the upsamplers' depth-to-space convs, measured with their own epilogue
    // (its scattered 8-byte stores cost the kernels differently; the output, (2T - 1) x 2H x 2W x N / 8 rows, fits the same scratch)
    const int tepi = (g_in.conv && epi == EPI_D2S) ? EPI_D2S : EPI_BIAS;
    GemmArgs g = g_in;
    void* scratch = nullptr;
    // (a call that leaves its K ranges to the consumer - GemmArgs::defer_parts - is measured in that form, on the ring tiles that serve it)
    const bool defer = g_in.defer_parts != nullptr;
    const size_t scratch_bytes = defer ? (size_t)ltx_gemm_split_factor(g) * g.M * g.N * sizeof(float) : (size_t)g.M * g.N * sizeof(bf16_t);
    if (hipMalloc(&scratch, scratch_bytes) != hipSuccess) { (void)hipGetLastError(); return LTX_OK; }
    g.C = scratch; g.ldc = g.N; g.resid = nullptr; g.gate = nullptr; g.c_seg_shift = 0; g.c_seg_stride = 0; g.rowsq = nullptr;
    g.C2 = nullptr; g.scale2 = nullptr; g.rs_sq = nullptr; g.cvec = nullptr;      // (norm-fold calls share the plain call's plan: same shape, same loop)
    g.defer_parts = defer ? reinterpret_cast<float*>(scratch) : nullptr;
    struct Guard {                                           // events and scratch are released on every return path
        void* scratch; hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Guard() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (scratch) (void)hipFree(scratch); }
    } guard{scratch};
    HIP_TRY(hipEventCreate(&guard.e0)); HIP_TRY(hipEventCreate(&guard.e1));
    const hipEvent_t e0 = guard.e0, e1 = guard.e1;
    const int nk = (g.K + 63) / 64 * (g.conv ? g.ntaps : 1);
    float best = 1e30f;
    const bool p8_off = ltx_opt().gemm_off & LTX_FAM_P8, halo_off = ltx_opt().gemm_off & LTX_FAM_HALO;
    const bool split_shape = ltx_gemm_split_factor(g) > 1;     // split shapes run gemm_big tiles only (same K partition in every plan)
    const bool asm16_off = ltx_opt().gemm_off & LTX_FAM_ASM16;
    for (int plan = defer ? kPlanRing : 0; plan < kPlanRing + ltx_gemm_ring_tiles(); ++plan) {
        if (plan >= kPlanRing) {
            if (!plan_ok(g, tepi, plan)) continue;
            // a ring tile is built for grids of about one round: skip the ones that would need more than three
            if ((int64_t)cdiv(g.M, ltx_gemm_ring_tile_bm(plan - kPlanRing)) * cdiv(g.N, ltx_gemm_ring_tile_bn(plan - kPlanRing)) * ltx_gemm_split_factor(g) > 768) continue;
        } else if (plan > kPlanAsm16Conv) { plan = kPlanRing - 1; continue; }
        else if (plan == kPlanAsm16Conv) {
            // a candidate from 1024 input channels up only: at 512 / 256 it ties with / loses to the halo-staged kernel in the
            // pipeline (1637-1675 vs 1617-1623 us, 3572 vs 3365-3462) while back-to-back repeats of the measurement favour it
            // (profiles/r5zz_bench_c2_kernel_stats.md, launches 22 and 33); gemm_plan=asm16c:256x256 still forces it
            if (asm16_off || g.Cin < 1024 || !plan_ok(g, tepi, plan)) continue;
        } else if (plan >= kPlanAsm16 + 3) { plan = kPlanAsm16Conv - 1; continue; }
        else if (plan >= kPlanAsm16) {
            if (asm16_off || !plan_ok(g, tepi, plan)) continue;
        } else if (plan >= kPlanHalo + 2) { plan = kPlanAsm16 - 1; continue; }
        else if (plan < kPlanP8) {
            if (plan >= kNumTiles) { plan = kPlanP8 - 1; continue; }
            if (!tile_fits(kTiles[plan], g.N)) continue;
        } else if (plan < kPlanHalo) {
            if (plan >= kPlanP8 + 2) { plan = kPlanHalo - 1; continue; }
            if (split_shape || p8_off || nk < 2 || g.N <= 64 || (plan == kPlanP8 && g.N <= 128) || !ltx_gemm_p8_fits(g)) continue;
        } else if (split_shape || halo_off || !ltx_conv_halo_eligible(g, tepi, plan == kPlanHalo ? 128 : 256)) continue;
        // warm launch (code object load, caches), timed on its own to size the measurement: ~1.5 ms of launches,
        // 3..16 of them, best of three rounds
        HIP_TRY(hipEventRecord(e0, s));
        int rc = run_plan(g, tepi, plan, s);
        if (rc != LTX_OK) continue;
        HIP_TRY(hipEventRecord(e1, s));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (ms >= 10.f) {                                    // a launch this long (the 13B decode's last stages) is its own measurement
            if (ms < best) { best = ms; *plan_out = plan; }
            continue;
        }
        int n = ms > 0.f ? (int)(1.5f / ms) : 16;
        n = n < 3 ? 3 : (n > 16 ? 16 : n);
        for (int round = 0; round < 3; ++round) {
            HIP_TRY(hipEventRecord(e0, s));
            bool ok = true;
            for (int i = 0; i < n; ++i) ok = ok && run_plan(g, tepi, plan, s) == LTX_OK;
            HIP_TRY(hipEventRecord(e1, s));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            ms /= (float)n;
            if (ok && ms < best) { best = ms; *plan_out = plan; }     // a plan whose launch failed is never the winner
        }
    }
    return LTX_OK;
}
}  // namespace

namespace {
std::atomic<int> g_autotune{1};      // ltx_set_autotune(0): never measure inside a call (cached / loaded plans, else the static model)

PlanKey plan_key(const GemmArgs& g, int epi) {
    PlanKey key; memset(&key, 0, sizeof(key));
    key.M = g.M; key.N = g.N; key.K = g.K; key.conv = g.conv ? (epi == EPI_D2S ? 2 : 1) : (g.defer_parts ? 3 : 0);      // 2: a depth-to-space conv, 3: a linear layer with deferred K ranges (their own measurements)
    if (g.conv) { key.ntaps = g.ntaps; key.T = g.T; key.H = g.H; key.W = g.Wd; }
    return key;
}
const char* plan_name(int plan) {
    if (plan >= kPlanRing) return ltx_gemm_ring_tile_name(plan - kPlanRing);
    if (plan == kPlanAsm16Conv) return "asm16c:256x256";
    if (plan >= kPlanAsm16) return plan == kPlanAsm16 ? "asm16:256x256" : (plan == kPlanAsm16 + 1 ? "asm16:160x256" : "asm16:320x256");
    if (plan >= kPlanHalo) return plan == kPlanHalo ? "halo:128" : "halo:256";
    if (plan >= kPlanP8) return plan == kPlanP8 ? "p8:256" : "p8:128";
    return plan >= 0 && plan < kNumTiles ? kTiles[plan].name : "";
}
int plan_from_name(const char* n) {
    for (int i = 0; i < ltx_gemm_ring_tiles(); ++i) if (!strcmp(n, ltx_gemm_ring_tile_name(i))) return kPlanRing + i;
    if (!strcmp(n, "asm16c:256x256")) return kPlanAsm16Conv;
    if (!strcmp(n, "asm16:256x256")) return kPlanAsm16; if (!strcmp(n, "asm16:160x256")) return kPlanAsm16 + 1; if (!strcmp(n, "asm16:320x256")) return kPlanAsm16 + 2;
    if (!strcmp(n, "halo:128")) return kPlanHalo; if (!strcmp(n, "halo:256")) return kPlanHalo + 1;
    if (!strcmp(n, "p8:256")) return kPlanP8; if (!strcmp(n, "p8:128")) return kPlanP8 + 1;
    for (int i = 0; i < kNumTiles; ++i) if (!strcmp(n, kTiles[i].name)) return i;
    return -1;
}
// *plan holds the static model's choice on entry; replaced by the cached plan, or by a fresh measurement when allowed
int cached_or_tuned_plan(const GemmArgs& g, int epi, hipStream_t s, int* plan) {
    if (!ltx_opt().gemm_tune) return LTX_OK;
    const PlanKey key = plan_key(g, epi);
    std::lock_guard<std::mutex> lock(g_plan_mu);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        if (!g_autotune.load()) return LTX_OK;
        int tuned = *plan;
        LTX_TRY(tune_plan(g, epi, s, *plan, &tuned));
        it = g_plans.emplace(key, tuned).first;
    }
    *plan = it->second;
    return LTX_OK;
}
}  // namespace

// ---- public plan control (include/ltxhip.h) ---------------------------------------------------------------------------
extern "C" int ltx_set_autotune(int enabled) { g_autotune.store(enabled ? 1 : 0); return LTX_OK; }
extern "C" int ltx_plan_save(const char* path) {
    if (!path) LTX_FAIL(LTX_ERR_ARG, "ltx_plan_save: null path");
    FILE* f = fopen(path, "w");
    if (!f) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_plan_save: cannot write ") + path);
    fprintf(f, "# ltxhip GEMM plans: M N K conv ntaps T H W plan\n");
    std::lock_guard<std::mutex> lock(g_plan_mu);
    for (const auto& kv : g_plans)
        fprintf(f, "%d %d %d %d %d %d %d %d %s\n", kv.first.M, kv.first.N, kv.first.K, kv.first.conv, kv.first.ntaps, kv.first.T, kv.first.H, kv.first.W, plan_name(kv.second));
    fclose(f);
    return LTX_OK;
}
extern "C" int ltx_plan_load(const char* path) {
    if (!path) LTX_FAIL(LTX_ERR_ARG, "ltx_plan_load: null path");
    FILE* f = fopen(path, "r");
    if (!f) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_plan_load: cannot read ") + path);
    char line[256], name[64];
    std::lock_guard<std::mutex> lock(g_plan_mu);
    int n = 0;
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#' || line[0] == '\n') continue;
        PlanKey key; memset(&key, 0, sizeof(key));
        if (sscanf(line, "%d %d %d %d %d %d %d %d %63s", &key.M, &key.N, &key.K, &key.conv, &key.ntaps, &key.T, &key.H, &key.W, name) != 9) { fclose(f); LTX_FAIL(LTX_ERR_ARG, std::string("ltx_plan_load: malformed line: ") + line); }
        const int plan = plan_from_name(name);
        if (plan < 0) { fclose(f); LTX_FAIL(LTX_ERR_ARG, std::string("ltx_plan_load: unknown plan name: ") + name); }
        {   // the shape-level predicates of tune_plan (run_plan re-checks the call's own operands and falls back to the static model)
            const bool kconv = key.conv == 1 || key.conv == 2;                                            // 2: a depth-to-space conv; 3: a linear layer with deferred K ranges
            GemmArgs g; g.M = key.M; g.N = key.N; g.K = key.K; g.conv = kconv ? 1 : 0; g.ntaps = kconv ? key.ntaps : 1;
            const int nk = (key.K + 63) / 64 * g.ntaps;
            const bool dims_ok = key.M > 0 && key.N > 0 && key.K > 0 && key.conv >= 0 && key.conv <= 3 && (!kconv || (key.ntaps > 0 && key.T > 0 && key.H > 0 && key.W > 0)) && (key.conv != 3 || plan >= kPlanRing);
            if (!dims_ok || !plan_shape_ok(plan, key.N, nk, ltx_gemm_split_factor(g) > 1) || (plan >= kPlanRing && ((kconv && (key.ntaps != 27 || key.conv == 2 || ltx_gemm_ring_tile_bn(plan - kPlanRing) < 64)) || key.K % 8 || key.M > 2048)) || (plan == kPlanAsm16Conv && (!kconv || key.ntaps != 27 || key.K % 64)) || (plan >= kPlanAsm16 && plan < kPlanAsm16Conv && (kconv || key.conv == 3 || key.K % 64)) || (plan >= kPlanHalo && plan < kPlanAsm16 && (!kconv || key.ntaps != 27 || key.K % 64 || key.N % (plan == kPlanHalo ? 128 : 256)))) {
                fclose(f); LTX_FAIL(LTX_ERR_ARG, std::string("ltx_plan_load: plan not valid for its shape: ") + line);
            }
        }
        g_plans[key] = plan; ++n;
    }
    fclose(f);
    return LTX_OK;
}

int ltx_launch_gemm_big(const GemmArgs& g_in, int epi, hipStream_t s) {
    GemmArgs g = g_in;
    ltx_prof_kernel(LTX_PROFK_GEMM_BIG);                   // the kernels this dispatcher hands over to overwrite it
    // XCD-contiguous tile order (measured, tools/microbench.py xcd: linear +3..19 %, conv +5 %)
    g.xcd_remap = ltx_exp("xcd_remap", 1);
    if (g.pn_on) return ltx_launch_conv_halo(g, epi, g.N, s);      // fused output norm: only that kernel's wide epilogue carries it
    if (g.defer_parts) {                                   // K-range sums left to the consumer: gemm_ring.hip's tiles only (the caller asked ltx_gemm_defer_ok)
        if (epi != EPI_BIAS || g.bias || g.conv || g.M > 512 || !ltx_gemm_ring_fits(g, EPI_BIAS)) LTX_FAIL(LTX_ERR_ARG, "gemm: defer_parts needs a bare linear layer of at most 512 rows (ltx_gemm_defer_ok)");
        int plan = kPlanRing + ltx_gemm_ring_pick_tile(g);
        if (ltx_opt().gemm_plan[0]) { const int f = plan_from_name(ltx_opt().gemm_plan); if (f >= kPlanRing && plan_ok(g, epi, f)) plan = f; }
        else {
            int cached = plan;
            (void)cached_or_tuned_plan(g, epi, s, &cached);
            if (cached >= kPlanRing && plan_ok(g, epi, cached)) plan = cached;
        }
        return ltx_launch_gemm_ring(g, epi, plan - kPlanRing, s);
    }
    const LtxOptions& o = ltx_opt();
    if (g.C2 || g.rs_sq) {
        // Norm fold (kernels.h): only gemm_asm16's wide epilogue carries the second output / the row scale (the caller asked
        // ltx_gemm_fold_ok).  Tile: the shape's measured plan where that is an asm16 tile (or a forced one), else the family's own choice.
        if (!ltx_gemm_fold_ok(g, epi)) LTX_FAIL(LTX_ERR_ARG, "gemm: norm-fold arguments on a call gemm_asm16 does not serve (ltx_gemm_fold_ok)");
        const int t = ltx_gemm_asm_pick_tile(g.M, g.N);                  // kAsmTiles order: 256 x 256, 320 x 256, 160 x 256
        int plan = kPlanAsm16 + (t == 0 ? 0 : (t == 1 ? 2 : 1));
        if (o.gemm_plan[0]) { const int f = plan_from_name(o.gemm_plan); if (f >= kPlanAsm16 && f < kPlanAsm16 + 3) plan = f; }
        else {
            int cached = plan;
            (void)cached_or_tuned_plan(g, epi, s, &cached);
            if (cached >= kPlanAsm16 && cached < kPlanAsm16 + 3) plan = cached;
        }
        return ltx_launch_gemm_asm16(g, epi, plan - kPlanAsm16, s);
    }
    const bool split_shape = ltx_gemm_split_factor(g) > 1;
    const int nk = (g.K + 63) / 64 * (g.conv ? g.ntaps : 1);
    // Option gemm_plan (tests, A/B): one plan forced wherever the call is eligible for it - a full plan name, or a family name
    // ("asm16", "ring") for that family's own tile choice.  A gemm_big tile name is always honoured (ltx_gemm_big_pick_tile).
    int forced = -1;
    if (o.gemm_plan[0]) {
        if (!strcmp(o.gemm_plan, "asm16")) {
            if (!g.conv && g.M >= 2048 && g.N >= 1024) { const int t = ltx_gemm_asm_pick_tile(g.M, g.N); forced = kPlanAsm16 + (t == 0 ? 0 : (t == 1 ? 2 : 1)); }   // kAsmTiles order: 256 x 256, 320 x 256, 160 x 256
        } else if (!strcmp(o.gemm_plan, "ring")) {
            if (ltx_gemm_ring_fits(g, epi)) forced = kPlanRing + ltx_gemm_ring_pick_tile(g);
        } else forced = plan_from_name(o.gemm_plan);
        if (forced >= kNumTiles) {                          // not a gemm_big tile: the plan's own predicates for THIS call
            bool ok = plan_ok(g, epi, forced);
            if (forced >= kPlanHalo && forced < kPlanAsm16) ok = ok && g.conv && ltx_conv_halo_eligible(g, epi, forced == kPlanHalo ? 128 : 256);
            if (forced >= kPlanP8 && forced < kPlanHalo) ok = ok && nk >= 2;
            if (split_shape && forced < kPlanRing && forced != kPlanAsm16Conv) ok = false;      // split shapes: gemm_big tiles, ring tiles or the conv-mode asm16 tile (one K partition whatever the plan)
            if (!ok) forced = -1;
        }
    }
    const bool big_forced = forced >= 0 && forced < kNumTiles;
    if (forced >= kPlanRing) return ltx_launch_gemm_ring(g, epi, forced - kPlanRing, s);
    if (forced == kPlanAsm16Conv) return ltx_launch_gemm_asm16_conv(g, epi, s);
    if (split_shape) {                                     // gemm_big tiles (or gemm_ring's, which keep their K partition): one partition whatever the plan
        int plan = ltx_gemm_big_pick_tile(g.M, g.N);
        if (!big_forced) {
            // static model: convs over planes of at most 512 voxels (C1's mid block, an edge tile of the tiled decode) on the deep ring
            // (72 -> 41 us at 384 voxels x 1024 channels, profiles/r5l_ring_conv_probe.jsonl)
            if (g.conv && g.M <= 512 && ltx_gemm_ring_fits(g, epi)) plan = kPlanRing + ltx_gemm_ring_pick_tile(g);
            (void)cached_or_tuned_plan(g, epi, s, &plan);
        }
        if (plan >= kPlanRing && plan_ok(g, epi, plan)) return ltx_launch_gemm_ring(g, epi, plan - kPlanRing, s);
        if (plan == kPlanAsm16Conv && plan_ok(g, epi, plan)) return ltx_launch_gemm_asm16_conv(g, epi, s);
        if (plan >= kPlanP8) plan = ltx_gemm_big_pick_tile(g.M, g.N);     // (includes the halo and asm16 families)
        return g.conv ? launch_tile<true>(g, epi, plan, s) : launch_tile<false>(g, epi, plan, s);
    }
    // conv_out (N = 48, unpatchify): the halo-staged kernel on its 64-wide tile - one staging of the 128-channel activation
    // per nine taps instead of one per tap (the per-tap 192 x 64 tile moves 27 x 610 MB through L2 -> LDS at C2).  Same K order:
    // same bits.  gemm_off=halo_out: the per-tap tile (A/B aid).
    if (g.conv && epi == EPI_UNPATCH && !(o.gemm_off & LTX_FAM_HALO_OUT) && ltx_conv_halo_eligible(g, epi, 64)) return ltx_launch_conv_halo(g, epi, 64, s);
    if (forced >= kPlanP8) return run_plan(g, epi, forced, s);
    int plan = ltx_gemm_big_pick_tile(g.M, g.N);
    if (!big_forced) {
        // static model, small M: the deep-ring tile whose grid wastes the least of a round (these shapes are never split, see
        // ltx_gemm_split_factor: a gemm_big tile would leave most CUs idle)
        if (!g.conv && g.M <= 512 && ltx_gemm_ring_fits(g, epi)) plan = kPlanRing + ltx_gemm_ring_pick_tile(g);
        // static model (no measured plan: gemm_tune=0 / ltx_set_autotune(0) without a plan file): the large linear layers take
        // the one-wave-per-SIMD kernel too, tile by whole rounds x tile area (it won every such shape that was measured)
        if (!g.conv && !(o.gemm_off & LTX_FAM_ASM16) && g.M >= 2048 && g.N >= 1024 && ltx_gemm_asm16_fits(g, epi)) {
            const int t = ltx_gemm_asm_pick_tile(g.M, g.N);              // 0: 256 x 256, 1: 320 x 256, 2: 160 x 256
            plan = kPlanAsm16 + (t == 0 ? 0 : (t == 1 ? 2 : 1));
        }
        LTX_TRY(cached_or_tuned_plan(g, epi, s, &plan));
    }
    return run_plan(g, epi, plan, s);
}

// tuning/diagnostic aid: name of the plan cached for a shape ("" if none)
const char* ltx_gemm_plan_name(int M, int N, int K, int conv, int ntaps, int T, int H, int W) {
    if (!conv) {                                                   // shapes the asm family serves (by shape, not by plan)
        GemmArgs g; g.M = M; g.N = N; g.K = K; g.lda = K;
        if (ltx_gemm_asm_eligible(g, LTX_DT_BF16, EPI_BIAS)) return ltx_gemm_asm_tile_name(ltx_gemm_asm_pick_tile(M, N));
    }
    PlanKey key; memset(&key, 0, sizeof(key));
    key.M = M; key.N = N; key.K = K; key.conv = conv;
    if (conv) { key.ntaps = ntaps; key.T = T; key.H = H; key.W = W; }
    std::lock_guard<std::mutex> lock(g_plan_mu);
    auto it = g_plans.find(key);
    return it == g_plans.end() ? "" : plan_name(it->second);
}
