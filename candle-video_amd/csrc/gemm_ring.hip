// Small-M linear layers (C1's 384 video tokens, the 128 text rows of the context projections and of the T5 encoder):
// bf16 GEMM on small output tiles with a DEEP ring of LDS stages (plan family "ring:*" of gemm_big.hip's plan measurement).
//
// Why its own kernel.  At M <= 512 a DiT linear layer has fewer outputs than 256 CUs x one 128 x 128 tile, so every tile shape
// of gemm_big / gemm_asm16 leaves CUs idle or runs a handful of K-steps per block, and their two-stage pipelines (the loads of
// step t+1 in flight while step t is multiplied) then pay one memory latency per K-step: 0.4-0.75 us per step whatever it
// computes (tools/small_m_probe.py; docs/lab_notes.md R4.8 - qkv at M = 384 ran 25 us against a 7 us floor).  What bounds such
// a shape is the per-CU L2 -> LDS path (~64 B/clk): a CU that owns a BM x BN tile moves (BM + BN) * K * 2 bytes through it.
// So: tiles small enough that the grid is ONE round of ~256 blocks (96 x 96 / 96 x 128 / 96 x 64 at M = 384), one block per CU,
// and NS = 5..8 stages of (BM + BN) * 128 B in the 160 KiB of LDS, NS - 1 of them in flight, with a COUNTED vmcnt at the one
// barrier per K-step - the step's loads were issued NS - 1 steps earlier, so the latency is paid once per block, not per step.
//
//   * 256 threads = 4 waves (one per SIMD) as 2 (M) x 2 (N), wave tile (BM/2) x (BN/2) of v_mfma_f32_16x16x32_bf16;
//   * operands HBM/L2 -> LDS by buffer_load ... lds pieces of 8 rows x 128 B, same swizzle as gemm_big (16-byte chunk ^
//     ((row >> 1) & 7) on the source side and on the fragment reads); every wave issues (BM + BN) / 32 pieces per stage, dead
//     stages (past the K range) are out-of-range pieces (zeros, no memory traffic), so the vmcnt arithmetic is static;
//   * fragments are read one k half ahead into a second register set (conv_halo.hip's pipelined form), the barrier sits between
//     the two halves of a step;
//   * K partition and summation order are gemm_big's: the shape-only split factor (ltx_gemm_split_factor), K-ranges
//     [part * nk / sf, (part + 1) * nk / sf), slabs + ticket, canonical ((s0 + s1) + s2) + ... by the last arriver; k ascending
//     inside a range with the same MFMA - so a "ring:*" plan returns the same bits as every other plan of the shape
//     (tests/test_gpu_determinism.py, tests/test_gpu_gemm_ring.py);
//   * block order: XCD x gets a contiguous run of (column tile, part, row tile) triples, row tile fastest - the blocks that share
//     a weight tile or an activation K-range run on one XCD and meet in its L2, the weight matrix leaves HBM once.
#include <atomic>
#include <cstring>
#include "gemm_common.h"
#include "options.h"

#ifndef RING_ABL
#define RING_ABL 0      // timing ablations (wrong results): 1 no MFMAs, 2 no fragment reads, 4 loads all hit one line set
#endif

namespace {

constexpr int ROWB = 128;
constexpr uint32_t OOB = 0x80000000u;
__device__ __forceinline__ int swz_r(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int I, int N, typename F> __device__ __forceinline__ void sfor_r(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor_r<I + 1, N>(f); }
}

extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];
#ifdef RING_TRACE       // per-block phase stamps (100 MHz wall clock) of the last launch: tools/ring_trace.py
__device__ unsigned g_ring_trace[1024 * 8];
#define RSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_ring_trace[blockIdx.x * 8 + (i)] = (unsigned)wall_clock64(); } while (0)
#else
#define RSTAMP(i)
#endif

template <int BM, int BN, int NS, int EPI, bool SPEC, bool CONV = false>
__global__ __launch_bounds__(SPEC ? 512 : 256) void gemm_ring_kernel(const GemmArgs g) {
    static_assert(!CONV || SPEC, "conv mode: the stage bookkeeping lives in the producer waves");
    RSTAMP(0);
    // the kernel arguments of the set-up requested together (the compiler's order: geometry, K partition, then - 400 instructions later,
    // in front of the first LDS-DMA piece - the operand pointers, each batch a scalar-cache miss of a 15-us launch; gemm_asm.hip)
#ifndef RING_NO_KERNARG_BATCH
    asm volatile("" :: "s"(g.A), "s"(g.W), "s"(g.Wp), "s"(g.C), "s"(g.bias), "s"(g.M), "s"(g.N), "s"(g.K), "s"(g.lda), "s"(g.ldc), "s"(g.sk_sf), "s"(g.sk_ws), "s"(g.sk_cnt),
                 "s"(g.defer_parts), "s"(gridDim.x));
#endif
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16, NM = FM * FN;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int PA = BM / 32, PW = BN / 32, P = PA + PW;      // LDS-DMA pieces per issuing wave per stage (activation, weight)
    constexpr int NSLOT = NM > FN + FM ? (NM > P ? NM : P) : (FN + FM > P ? FN + FM : P);      // instruction slots of a k half
    static_assert(BM % 32 == 0 && BN % 32 == 0 && WM % 16 == 0 && WN % 16 == 0, "tile / wave layout");
    static_assert(NS >= 3 && (NS - 1) * P <= 63 && NS * STAGE <= 160 * 1024, "ring depth: vmcnt is 6 bits, LDS is 160 KiB");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SPEC: waves 0..3 multiply (one per SIMD), waves 4..7 (their SIMD partners) do nothing but feed the ring - a wave that
    // stalls at the issue of an LDS-DMA piece (the CU's one load path takes 16 cycles per piece) then stalls no MFMA
    const bool producer = SPEC && wave >= 4;
    const int cw = wave & 3, wm = cw >> 1, wn = cw & 1;
    const int ntm = (g.M + BM - 1) / BM, sf = g.sk_sf;
    // block -> (column tile, part, row tile): XCD-contiguous runs (blocks b and b + 8 share an XCD), row tile fastest
    int bid = blockIdx.x;
    {
        const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int mt = bid % ntm, rest = bid / ntm, part = rest % sf, nt = rest / sf;
    const int m0 = mt * BM, n0 = nt * BN;
    const int ktiles = (g.K + 63) / 64;
    const int nk = ktiles * (CONV ? g.ntaps : 1);        // conv: K-steps in the shared order (frame tap, 64-channel slice, in-plane tap)
    const int kt0 = sf > 1 ? (int)((int64_t)part * nk / sf) : 0, kt1 = sf > 1 ? (int)((int64_t)(part + 1) * nk / sf) : nk;
    const int n = kt1 - kt0;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)OOB, 0x00020000);
    const bool packed = g.Wp != nullptr;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(packed ? reinterpret_cast<const bf16_t*>(g.Wp) : W), 0, (int)OOB, 0x00020000);
    const uint32_t w_step = packed ? 4096u : 128u;                 // bytes from one K-step to the next in the weight operand
    // piece j of issuing wave cw fills rows 8 * (4 j + cw) .. + 7 of the activation (j < PA) or weight image of a stage; a lane
    // writes 16 bytes: row lr = lane >> 3, physical chunk pc = lane & 7 holds logical chunk pc ^ ((row >> 1) & 7)
    const int lr = lane >> 3, pc = lane & 7;
    uint32_t voff[P]; int klim[P];                                  // klim: first K position at which the lane's 16-byte chunk lies outside K
    // conv mode (gemm_big.hip's addressing): an activation row is a voxel of the channels-last tensor, a K-step reads it shifted
    // by a tap; ct = its frame (replicate padding clamps ct + dt, vae.rs:374-413), vmask = which of the 3 x 3 in-plane
    // neighbours exist (zero padding: the others are out-of-range pieces, vae.rs:337-349)
    int ct[CONV ? PA : 1], vmask[CONV ? PA : 1]; (void)ct; (void)vmask;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const bool isa = j < PA;
        const int row = 8 * (4 * (isa ? j : j - PA) + cw) + lr;
        const int chunk = pc ^ ((row >> 1) & 7);
        klim[j] = g.K - chunk * 8;
        if (isa) {
            int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
            if constexpr (CONV) {
                const int w = m % g.Wd, t1 = m / g.Wd, h = t1 % g.H, t2 = t1 / g.H;
                ct[j] = t2 % g.T;
                vmask[j] = (h > 0 ? 1 : 0) | 2 | (h < g.H - 1 ? 4 : 0) | (w > 0 ? 8 : 0) | 16 | (w < g.Wd - 1 ? 32 : 0);
                voff[j] = ((uint32_t)m * (uint32_t)g.Cin + chunk * 8) * 2u;
            } else voff[j] = ((uint32_t)m * (uint32_t)g.lda + chunk * 8) * 2u;
        }
        else {
            int c = n0 + row; if (c > g.N - 1) c = g.N - 1;
            if (packed) { voff[j] = (uint32_t)(c >> 5) * (uint32_t)nk * 4096u + (uint32_t)(c & 31) * 128u + chunk * 16u; klim[j] = 0x3fffffff; }   // (K is zero padded there)
            else voff[j] = ((uint32_t)c * (uint32_t)g.K + chunk * 8) * 2u;
        }
    }
    // what a stage's pieces share (wave-uniform): the K position its chunks are tested against, the scalar offsets of the two
    // operands, and in conv mode the tap (frame delta, the validity bits it needs, its byte distance inside the plane)
    struct Stg { int kpos; uint32_t a_soff, b_soff; int dt, vbit; };
    const uint32_t frame_bytes = CONV ? (uint32_t)g.H * g.Wd * g.Cin * 2u : 0u;
    auto stage_of = [&](int i) {
        Stg c; c.dt = 0; c.vbit = 0;
        if (i >= n) { c.kpos = 0x40000000; c.a_soff = 0u; c.b_soff = 0u; return c; }      // a dead stage lies beyond every klim
        const int kt = kt0 + i;
        if constexpr (CONV) {
            const int per_it = ktiles * 9, it = kt / per_it, r2 = kt - it * per_it, kk = r2 / 9, hw = r2 - kk * 9;
            const int ih = hw / 3, iw = hw - ih * 3, dh = ih - 1, dw = iw - 1;
            c.kpos = kk * 64;
            c.dt = it - g.pad_t;
            c.vbit = (1 << (dh + 1)) | (8 << (dw + 1));
            c.a_soff = (uint32_t)kk * 128u + (uint32_t)((dh * g.Wd + dw) * g.Cin * 2);     // may be "negative": added modulo 2^32 to the lane offset
            c.b_soff = ((uint32_t)(it * 9 + hw) * (uint32_t)g.N * (uint32_t)g.K + (uint32_t)kk * 64u) * 2u;
        } else { c.kpos = kt * 64; c.a_soff = (uint32_t)kt * 128u; c.b_soff = (uint32_t)kt * w_step; }
        return c;
    };
    auto issue_piece = [&](const Stg& c, unsigned char* base, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr bool isa = j < PA;
        uint32_t vo = c.kpos < klim[j] ? voff[j] : OOB;
        uint32_t soff = isa ? c.a_soff : c.b_soff;
        if constexpr (CONV && isa) {
            int tt = ct[j] + c.dt; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);
            vo = (c.kpos < klim[j] && (vmask[j] & c.vbit) == c.vbit) ? voff[j] + (uint32_t)(tt - ct[j]) * frame_bytes + c.a_soff : OOB;
            soff = 0u;
        }
#if RING_ABL & 4
        vo = lane * 16;
#endif
        unsigned char* dst = base + (isa ? 0 : BM * ROWB) + (4 * (isa ? j : j - PA) + cw) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isa ? ra : rw, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, (int)soff, 0, 0);
    };
    auto issue = [&](int i, int slot) {                             // stage i of this block's K range into ring slot `slot`, all pieces
        const Stg c = stage_of(i);
        sfor_r<0, P>([&](auto j) { issue_piece(c, ring_smem + slot * STAGE, j); });
    };

    f32x4 acc[FM][FN];
    const int frow = lane & 15, fq = lane >> 4;
    // epilogue operands of this lane's outputs, fetched BEFORE the first LDS-DMA piece (loads return in order, so the counted
    // vmcnt waits below cover them) instead of after the K loop, where their latency was in nobody's shadow
    constexpr bool HAS_R = EPI == EPI_GATE_RESID || EPI == EPI_RESID;
    constexpr bool HAS_G = EPI == EPI_GATE_RESID;
    u32x2 pbias[FN], presid[HAS_R ? FM : 1][HAS_R ? FN : 1]; f32x4 pgate[HAS_G ? FM : 1][HAS_G ? FN : 1];
    if (!producer && !g.defer_parts) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            int nb = n0 + wn * WN + fn * 16 + 4 * fq; if (nb > g.N - 4) nb = g.N - 4;
            pbias[fn] = g.bias ? *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.bias) + nb) : (u32x2){0u, 0u};
        }
        if constexpr (HAS_R) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                int m = m0 + wm * WM + fm * 16 + frow; if (m > g.M - 1) m = g.M - 1;
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    int nb = n0 + wn * WN + fn * 16 + 4 * fq; if (nb > g.N - 4) nb = g.N - 4;
                    presid[fm][fn] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.resid) + (int64_t)m * g.ldr + nb);
                    if constexpr (HAS_G) pgate[fm][fn] = *reinterpret_cast<const f32x4*>(g.gate + (int64_t)(m / g.rows_per_batch) * g.gate_stride + nb);
                }
            }
        }
        asm volatile("" ::: "memory");           // (!SPEC) these loads stay in front of the ring's pieces: the vmcnt arithmetic counts on it
    }
    (void)pgate;

    Chunk16 w0[FN], a0[FM], w1[FN], a1[FM];
    bool first = true; (void)first;
    auto read_frag = [&](int slot, int kb, auto idx_tag, Chunk16 (&wf)[FN], Chunk16 (&af)[FM]) {
        constexpr int idx = decltype(idx_tag)::value;
#if RING_ABL & 2
        if (!first) return;
#endif
        const unsigned char* As = ring_smem + slot * STAGE;
        if constexpr (idx < FN) wf[idx].u = *reinterpret_cast<const u32x4*>(As + BM * ROWB + swz_r(wn * WN + idx * 16 + frow, kb * 4 + fq));
        else af[idx - FN].u = *reinterpret_cast<const u32x4*>(As + swz_r(wm * WM + (idx - FN) * 16 + frow, kb * 4 + fq));
    };
    auto mfma = [](f32x4& c, const Chunk16& w, const Chunk16& a_) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w.u), "v"(a_.u));
    };
    auto mma = [&](auto m_tag, const Chunk16 (&wf)[FN], const Chunk16 (&af)[FM]) {
        constexpr int mi = decltype(m_tag)::value;
#if !(RING_ABL & 1)
        // in place in the AGPR half of the register file: left to itself hipcc renames the accumulators from MFMA to MFMA and
        // moves them back at the loop edge (conv_halo.hip)
        mfma(acc[mi / FN][mi % FN], wf[mi % FN], af[mi / FN]);
#endif
    };

    // One barrier per K-step, between its two k halves.  At the barrier of step i: stage i + 1 has landed (every issuing wave
    // waited for its own pieces: all but the NS - 2 youngest stages), and every multiplying wave has finished reading slot
    // i % NS (lgkmcnt(0)), which is then refilled with stage i + NS.
    if (producer) {
#pragma unroll
        for (int s = 0; s < NS; ++s) issue(s, s);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NS - 1) * P) : "memory");
        int slot = 0;
        for (int i = 0; i < n; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NS - 2) * P) : "memory");
            issue(i + NS, slot);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the dead stages' zero fills have landed: the ring is quiet
    } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (!SPEC) {
#pragma unroll
            for (int s = 0; s < NS; ++s) issue(s, s);
        }
        RSTAMP(1);
        if constexpr (SPEC) asm volatile("s_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NS - 1) * P) : "memory");      // stage 0 has landed in every wave
        RSTAMP(2);
        sfor_r<0, FN + FM>([&](auto t) { read_frag(0, 0, t, w0, a0); });
        int slot = 0;
        for (int i = 0; i < n; ++i) {
            const int nslot = slot + 1 == NS ? 0 : slot + 1;
            first = false;
            // everything in program order, pinned by sched_barrier: one fragment read (of the NEXT half) in front of each MFMA
            sfor_r<0, NSLOT>([&](auto t) {
                constexpr int k = decltype(t)::value;
                if constexpr (k < FN + FM) read_frag(slot, 1, t, w1, a1);
                if constexpr (k < NM) mma(t, w0, a0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // a raw barrier (__syncthreads() would add a fence, i.e. vmcnt(0))
            if constexpr (SPEC) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((NS - 2) * P) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const Stg nxt = SPEC ? Stg{} : stage_of(i + NS);
            unsigned char* const base = ring_smem + slot * STAGE;
            sfor_r<0, NSLOT>([&](auto t) {                          // (!SPEC) stage i + NS into the slot just freed, its pieces spread over the MFMAs
                constexpr int k = decltype(t)::value;
                if constexpr (k < FN + FM) read_frag(nslot, 0, t, w0, a0);
                if constexpr (!SPEC && k < P) issue_piece(nxt, base, t);
                if constexpr (k < NM) mma(t, w1, a1);
                __builtin_amdgcn_sched_barrier(0);
            });
            slot = nslot;
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results before anything reads the accumulators
        RSTAMP(3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is out of the ring
    }

    if (g.defer_parts) {
        // Deferred reduction: this range's f32 sums, row-major, and nothing else (the consumer adds the ranges in part order)
        if (!producer) {
            float* pp = g.defer_parts + (int64_t)part * g.M * g.N;
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int m = m0 + wm * WM + fm * 16 + frow;
                if (m >= g.M) continue;
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const int nb = n0 + wn * WN + fn * 16 + 4 * fq;
                    if (nb < g.N) *reinterpret_cast<f32x4*>(pp + (int64_t)m * g.N + nb) = acc[fm][fn];
                }
            }
        }
        return;
    }
    if (sf > 1) {
        // In-launch reduction of a tile's parts (gemm_big.hip's protocol and order): f32 slabs, a ticket, the last arriver adds
        // ((s0 + s1) + s2) + ... and runs the epilogue
        constexpr int SLAB = BM * BN;
        const int tt = nt * ntm + mt;
        float* base = g.sk_ws + (int64_t)tt * sf * SLAB;
        float* slab = base + (int64_t)part * SLAB;
        if (!producer) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) *reinterpret_cast<f32x4*>(slab + ((fm * FN + fn) * 256 + tid) * 4) = acc[fm][fn];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RSTAMP(4);
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(ring_smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(g.sk_cnt + tt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)sf - 1u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g.sk_cnt + tt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // back to zero for the next launch
            }
            *flag = last;
        }
        __syncthreads();
        RSTAMP(5);
        if (*flag == 0u || producer) return;
        const bool reread_own = sf > 2 && part != 0;
        if (reread_own) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = *reinterpret_cast<const f32x4*>(base + ((fm * FN + fn) * 256 + tid) * 4);
        }
        for (int p = (reread_own || part == 0) ? 1 : 0; p < sf; ++p) {
            if (p == part && !reread_own) continue;
            const float* other = base + (int64_t)p * SLAB;
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] += *reinterpret_cast<const f32x4*>(other + ((fm * FN + fn) * 256 + tid) * 4);
        }
    }
    if (producer) return;

    RSTAMP(6);
    // D = Wfrag x Afrag: a lane holds 4 consecutive output columns of one row; the arithmetic of gemm_common.h's epilogue<> on
    // the prefetched operands
    bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = m0 + wm * WM + fm * 16 + frow;
        if (m >= g.M) continue;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            int nb = n0 + wn * WN + fn * 16 + 4 * fq;
            if (nb >= g.N) continue;
            float v[4] = {acc[fm][fn][0], acc[fm][fn][1], acc[fm][fn][2], acc[fm][fn][3]};
            if (g.bias) {
                float b[4];
                load4<bf16_t>(reinterpret_cast<const bf16_t*>(&pbias[fn]), b);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += b[i];
            }
            bf16_t* C = Cb;
            if constexpr (EPI == EPI_BIAS) {
                if (g.c_seg_shift) { const int sg = nb >> g.c_seg_shift; C += sg * g.c_seg_stride; nb -= sg << g.c_seg_shift; }
            } else if constexpr (EPI == EPI_GELU) {
                gelu_tanh4(v);
            } else if constexpr (EPI == EPI_GATE_RESID) {
                float r[4];
                load4<bf16_t>(reinterpret_cast<const bf16_t*>(&presid[fm][fn]), r);
                const f32x4 gt = pgate[fm][fn];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(gt[i], v[i], r[i]);       // ONE rounding, spelled out: every kernel that finishes these rows must agree
            } else if constexpr (EPI == EPI_RESID) {
                float r[4];
                load4<bf16_t>(reinterpret_cast<const bf16_t*>(&presid[fm][fn]), r);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += r[i];
            }
            store4<bf16_t>(C + (int64_t)m * g.ldc + nb, v);
        }
    }
#ifdef RING_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSTAMP(7);
#endif
}

struct RingTile { int bm, bn, ns; const char* name; };
// ns: stages that fit 160 KiB (at most 8: beyond that the ring holds more than any latency)
const RingTile kRing[] = {
    {96, 64, 8, "ring:96x64"}, {96, 96, 6, "ring:96x96"}, {96, 128, 5, "ring:96x128"}, {64, 64, 8, "ring:64x64"},
    {64, 128, 6, "ring:64x128"}, {128, 64, 6, "ring:128x64"}, {128, 128, 5, "ring:128x128"}, {128, 96, 5, "ring:128x96"},
    {96, 32, 8, "ring:96x32"}, {128, 32, 8, "ring:128x32"}, {64, 32, 8, "ring:64x32"},
};
constexpr int kNumRing = sizeof(kRing) / sizeof(kRing[0]);

template <int BM, int BN, int NS, int EPI, bool SPEC, bool CONV = false>
int launch_ring(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = NS * (BM + BN) * ROWB;
    static std::atomic<unsigned long long> attr_devs{0};
    auto kern = gemm_ring_kernel<BM, BN, NS, EPI, SPEC, CONV>;
    LTX_TRY(ltx_set_max_dyn_smem(attr_devs, reinterpret_cast<const void*>(kern), smem));
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    GemmArgs ga = g;
    if (g.defer_parts) { ga.sk_sf = ltx_gemm_split_factor(g); ga.sk_full = 0; ga.sk_ws = nullptr; ga.sk_cnt = nullptr; }     // the caller's buffer takes the ranges
    else LTX_TRY(ltx_gemm_split_workspace(&ga, tiles, BM, BN, s));
    ltx_prof_kernel(LTX_PROFK_GEMM_RING);
    LTX_LAUNCH_TIMED(kern, dim3((unsigned)(tiles * ga.sk_sf)), dim3(SPEC ? 512 : 256), smem, s, ga);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
template <int BM, int BN, int NS>
int launch_ring_epi(const GemmArgs& g, int epi, hipStream_t s) {
#ifdef LTX_EXPERIMENTS     // x_gemm_ring_spec=0: every wave loads and multiplies (256 threads), the first form of the kernel (A/B aid)
    if (!g.conv && !ltx_exp("gemm_ring_spec", 1)) switch (epi) {
        case EPI_BIAS: return launch_ring<BM, BN, NS, EPI_BIAS, false>(g, s);
        case EPI_GELU: return launch_ring<BM, BN, NS, EPI_GELU, false>(g, s);
        case EPI_GATE_RESID: return launch_ring<BM, BN, NS, EPI_GATE_RESID, false>(g, s);
        case EPI_RESID: return launch_ring<BM, BN, NS, EPI_RESID, false>(g, s);
    }
#endif
    if constexpr (BN >= 64) {      // conv mode (3 x 3 x 3 convs over planes of a few hundred voxels): the tiles of at least 64 columns, bias / residual epilogues
        if (g.conv) switch (epi) {
            case EPI_BIAS: return launch_ring<BM, BN, NS, EPI_BIAS, true, true>(g, s);
            case EPI_RESID: return launch_ring<BM, BN, NS, EPI_RESID, true, true>(g, s);
            default: LTX_FAIL(LTX_ERR_ARG, "gemm_ring (conv): bad epilogue");
        }
    }
    if (g.conv) LTX_FAIL(LTX_ERR_ARG, "gemm_ring (conv): tile not available");
    switch (epi) {
        case EPI_BIAS: return launch_ring<BM, BN, NS, EPI_BIAS, true>(g, s);
        case EPI_GELU: return launch_ring<BM, BN, NS, EPI_GELU, true>(g, s);
        case EPI_GATE_RESID: return launch_ring<BM, BN, NS, EPI_GATE_RESID, true>(g, s);
        case EPI_RESID: return launch_ring<BM, BN, NS, EPI_RESID, true>(g, s);
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_ring: bad epilogue");
}

}  // namespace

#ifdef RING_TRACE
extern "C" int ltx_dbg_ring_trace(unsigned* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_trace), (size_t)n * 4) == hipSuccess ? 0 : -1; }
#endif
namespace {
// one thread per 16-byte chunk of the packed image: chunk id -> (row group, K block, row in group, chunk in row)
__global__ __launch_bounds__(256) void ring_pack_kernel(const bf16_t* __restrict__ W, int N, int K, int nk, u32x4* __restrict__ out, int64_t nchunks) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nchunks) return;
    const int c8 = (int)(t & 7), r = (int)((t >> 3) & 31);
    const int64_t blk = t >> 8;
    const int kb = (int)(blk % nk); const int64_t grp = blk / nk;
    const int64_t nrow = grp * 32 + r; const int k = kb * 64 + c8 * 8;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (nrow < N && k < K) v = *reinterpret_cast<const u32x4*>(W + nrow * K + k);      // (K % 8 == 0: a chunk is inside K or outside it)
    out[t] = v;
}
}  // namespace
size_t ltx_ring_packed_bytes(int N, int K) { return (size_t)cdiv(N, 32) * cdiv(K, 64) * 4096; }
int ltx_pack_ring_weights(const void* W, int N, int K, void* out, hipStream_t s) {
    if (!W || !out || N < 1 || K < 8 || K % 8) LTX_FAIL(LTX_ERR_ARG, "ring pack: bf16 [N, K] with K a multiple of 8");
    const int64_t nchunks = (int64_t)(ltx_ring_packed_bytes(N, K) / 16);
    hipLaunchKernelGGL(ring_pack_kernel, dim3((unsigned)cdiv64(nchunks, 256)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(W), N, K, cdiv(K, 64), reinterpret_cast<u32x4*>(out), nchunks);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
int ltx_gemm_ring_tiles() { return kNumRing; }
const char* ltx_gemm_ring_tile_name(int i) { return i >= 0 && i < kNumRing ? kRing[i].name : ""; }
int ltx_gemm_ring_tile_bm(int i) { return i >= 0 && i < kNumRing ? kRing[i].bm : 1; }
int ltx_gemm_ring_tile_bn(int i) { return i >= 0 && i < kNumRing ? kRing[i].bn : 1; }

// Linear layers of at most 2048 rows (the plan measurement offers the family up to 512: ltx_gemm_split_factor) whose operands the 32-bit buffer offsets reach (gemm_big's own bound), K in whole 16-byte
// chunks, 4-column output groups inside or outside N as a whole.  Conv mode: 3 x 3 x 3 convs over at most 2048 voxels (the VAE's
// first stages at C1's 4 x 8 x 12 latent, the edge tiles of the tiled decode), bias / residual epilogues, the whole activation
// inside the 32-bit offsets.
bool ltx_gemm_ring_fits(const GemmArgs& g, int epi) {
    if (ltx_opt().gemm_off & LTX_FAM_RING) return false;
    if (g.pn_on || g.M < 1 || g.M > 2048 || g.N < 32 || g.N % 4 || g.K % 8) return false;
    if (g.conv) {
        if (g.ntaps != 27 || g.kh != 3 || g.kw != 3 || g.K != g.Cin || g.N < 64 || g.defer_parts || g.rowsq || g.c_seg_shift) return false;
        if (epi != EPI_BIAS && epi != EPI_RESID) return false;
        if (g.T < 1 || g.H < 1 || g.Wd < 1 || g.M % (g.T * g.H * g.Wd) != 0 || ((uintptr_t)g.A & 15)) return false;
        if ((double)g.M * g.Cin * 2.0 >= 2147483648.0 || g.ldc % 4 != 0) return false;
    } else if (g.lda % 8) return false;
    if (epi != EPI_BIAS && epi != EPI_GELU && epi != EPI_GATE_RESID && epi != EPI_RESID) return false;
    // the epilogue's prefetch loads 8 bytes of the residual row and 16 bytes of the gate row per lane (gemm_asm16's conditions)
    if ((epi == EPI_GATE_RESID || epi == EPI_RESID) && (!g.resid || g.ldr % 4 != 0 || ((uintptr_t)g.resid & 7))) return false;
    if (epi == EPI_GATE_RESID && (!g.gate || ((uintptr_t)g.gate & 15) || g.gate_stride % 4 != 0 || g.rows_per_batch < 1)) return false;
    if (g.bias && ((uintptr_t)g.bias & 7)) return false;
    return ltx_gemm_big_fits(g);
}
bool ltx_gemm_ring_tile_fits(const GemmArgs& g, int epi, int tile) {
    return tile >= 0 && tile < kNumRing && ltx_gemm_ring_fits(g, epi) && (!g.conv || kRing[tile].bn >= 64);
}

// Deferred reduction (GemmArgs::defer_parts) is this kernel's: bf16 linear layers of at most 512 rows that the ring serves, N in
// whole 16-byte f32 groups.  The launch itself carries no epilogue (EPI_BIAS instantiation, no bias): the consumer finishes the rows.
bool ltx_gemm_defer_ok(const GemmArgs& g_in, int epi) {
    if (epi != EPI_GATE_RESID && epi != EPI_RESID) return false;
    GemmArgs g = g_in; g.bias = nullptr; g.resid = nullptr; g.gate = nullptr;
    // the ring is reached through ltx_launch_gemm_big's plan dispatch only: with that family switched off (gemm_off=big) the call
    // would fall to gemm.hip's kernel, which knows nothing of defer_parts
    if (!ltx_gemm_big_eligible(g, LTX_DT_BF16)) return false;
    return g.M <= 512 && g.N % 8 == 0 && !g.rowsq && !g.c_seg_shift && ltx_gemm_ring_fits(g, EPI_BIAS);
}

int ltx_launch_gemm_ring(const GemmArgs& g, int epi, int tile, hipStream_t s) {
    if (!ltx_gemm_ring_tile_fits(g, epi, tile)) LTX_FAIL(LTX_ERR_ARG, "gemm_ring: shape not eligible");
    switch (tile) {
        case 0: return launch_ring_epi<96, 64, 8>(g, epi, s);
        case 1: return launch_ring_epi<96, 96, 6>(g, epi, s);
        case 2: return launch_ring_epi<96, 128, 5>(g, epi, s);
        case 3: return launch_ring_epi<64, 64, 8>(g, epi, s);
        case 4: return launch_ring_epi<64, 128, 6>(g, epi, s);
        case 5: return launch_ring_epi<128, 64, 6>(g, epi, s);
        case 6: return launch_ring_epi<128, 128, 5>(g, epi, s);
        case 7: return launch_ring_epi<128, 96, 5>(g, epi, s);
        case 8: return launch_ring_epi<96, 32, 8>(g, epi, s);
        case 9: return launch_ring_epi<128, 32, 8>(g, epi, s);
        case 10: return launch_ring_epi<64, 32, 8>(g, epi, s);
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_ring: unsupported tile");
}

// Static choice (no measured plan): the tile whose grid wastes the least of whole rounds of 256 blocks, larger tiles on ties.
int ltx_gemm_ring_pick_tile(const GemmArgs& g) {
    const int sf = ltx_gemm_split_factor(g);
    double best = 1e30; int bi = 1;
    for (int i = 0; i < kNumRing; ++i) {
        if (g.conv && kRing[i].bn < 64) continue;
        const int64_t blocks = (int64_t)cdiv(g.M, kRing[i].bm) * cdiv(g.N, kRing[i].bn) * sf;
        // a block's time ~ its bytes through the CU's load path, (bm + bn) per K element
        const double cost = (double)cdiv64(blocks, 256) * (double)(kRing[i].bm + kRing[i].bn);
        if (cost < best * 0.999) { best = cost; bi = i; }
    }
    return bi;
}
