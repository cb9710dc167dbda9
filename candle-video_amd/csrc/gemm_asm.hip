// One-wave-per-SIMD bf16 GEMM for the DiT's linear layers (nn::Linear + the fused epilogues of gemm_common.h):
//   C[M, N] = epi(A[M, K] @ W[N, K]^T + bias)
// The K loop is ONE generated inline-asm statement per tile shape (tools/gen_gemm_asm.py -> gemm_asm_loop.inc): four waves
// of 128 x 128 (or 160 x 64 / 160 x 128) output each, accumulators in the AGPR half of the 512-register file, and every
// LDS-DMA piece and fragment read placed by hand between the MFMAs - the interleave that hipcc cannot be made to emit.
// LDS image, source-side swizzle, buffer addressing with out-of-range = zeros and the epilogue arithmetic are those of
// gemm_big.hip; the results are BIT-IDENTICAL to gemm_big's whatever the MFMA shape: the matrix core accumulates its bf16
// products in ascending k as an f32 chain, so only the k order matters (tests/test_gpu_gemm_asm.py).
//
// Two kernels live here:
//   * gemm_asm16_kernel (round 3; v_mfma_f32_16x16x32_bf16, tiles 256 x 256, 160 x 256, 320 x 256): the plan family "asm16:*"
//     that gemm_big.hip's plan measurement tries beside the gemm_big tiles, and the default on every large DiT shape since
//     the DMA pieces of K-step t + 2 were spread over the whole iteration (s_memtime trace: packed into a third of the
//     K-step they queued on the CU's address path and stalled the only instruction stream of the SIMD) and the epilogue
//     was rebuilt around an f32 pass through LDS (docs/lab_notes.md section 4, "Linear GEMMs, round 3");
//   * gemm_asm_kernel (round 1; v_mfma_f32_32x32x16_bf16): the first attempt at this structure, kept behind LTX_GEMM_ASM=1
//     as a measured reference (4-12 % behind gemm_big on the DiT shapes: its DMA pieces are still issued in a burst).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "gemm_common.h"
#include "options.h"

namespace {

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#ifndef GEMM_ASM_LOOP_INC
#define GEMM_ASM_LOOP_INC "gemm_asm_loop.inc"
#endif
#include GEMM_ASM_LOOP_INC

extern __shared__ __attribute__((aligned(16))) unsigned char asm_smem[];

#ifdef LTX_EXPERIMENTS     // round 1's 32x32x16 kernel: 4-12 % behind gemm_big on the DiT shapes; experiment builds only (x_gemm_asm=1)
template <int BM, int BN, int WGM, int WGN> struct AsmLoop;
template <> struct AsmLoop<256, 256, 2, 2> {
    template <typename... T> static __device__ __forceinline__ void run(f32x32 (&c)[10], T&&... t) { gemm_asm_loop_256_256(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], t...); }
};
template <> struct AsmLoop<320, 256, 2, 2> {
    template <typename... T> static __device__ __forceinline__ void run(f32x32 (&c)[10], T&&... t) { gemm_asm_loop_320_256(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], t...); }
};
template <> struct AsmLoop<160, 256, 1, 4> {
    template <typename... T> static __device__ __forceinline__ void run(f32x32 (&c)[10], T&&... t) { gemm_asm_loop_160_256(c[0], c[1], c[2], c[3], c[4], t...); }
};

template <int BM, int BN, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_asm_kernel(const GemmArgs g) {
    constexpr int WM = BM / WGM, WN = BN / WGN, MB = WM / 32, NB = WN / 32, NT = MB * NB, AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int ntn = (g.N + BN - 1) / BN;
    // tile order: XCD-contiguous runs, columns of group_m row-tiles inside a run (gemm_big.hip)
    int bid = blockIdx.x;
    if (g.xcd_remap) {
        const int nblk = (int)gridDim.x, q = nblk >> 3, rr = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
    }
    int mt, nt;
    if (g.group_m > 1) {
        const int ntm = (g.M + BM - 1) / BM, gsz = g.group_m * ntn;
        const int grp = bid / gsz, w = bid - grp * gsz, gm0 = grp * g.group_m;
        const int rows = ntm - gm0 < g.group_m ? ntm - gm0 : g.group_m;
        nt = w / rows; mt = gm0 + (w - nt * rows);
    } else { mt = bid / ntn; nt = bid - mt * ntn; }
    const int m0 = mt * BM, n0 = nt * BN;

    // DMA source offsets: piece j of this wave fills LDS rows 8 * (4 j + wave) .. + 7; lane -> (row lane >> 3, physical chunk
    // lane & 7), logical chunk = physical ^ ((row >> 1) & 7).  Rows past M / N repeat the last one (never stored).
#ifndef GEMM_ASM_REG          // 1: register-staged operands (buffer loads to VGPRs two K-steps ahead + ds_write_b128), 256 x 256 tile only
#define GEMM_ASM_REG 0
#endif
    constexpr bool REGSTAGE = GEMM_ASM_REG && BM == 256 && BN == 256;
    const int lr = lane >> 3, pc = lane & 7;
    u32x16 dma0; u32x2 dma1 = {0x80000000u, 0x80000000u};
    uint32_t off[AI + BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 8 * (j * 4 + wave) + lr;
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
        off[j] = ((uint32_t)m * (uint32_t)g.lda + (uint32_t)(REGSTAGE ? pc : (pc ^ ((row >> 1) & 7))) * 8u) * 2u;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (j * 4 + wave) + lr;
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
        off[AI + j] = ((uint32_t)n * (uint32_t)g.K + (uint32_t)(REGSTAGE ? pc : (pc ^ ((row >> 1) & 7))) * 8u) * 2u;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) dma0[j] = j < AI + BI ? off[j] : 0x80000000u;
    if constexpr (AI + BI > 16) { dma1[0] = off[16]; if constexpr (AI + BI > 17) dma1[1] = off[17]; }

    // fragment read bases [stage][16-deep step]: lane (r, h) reads row r of a 32-row block, logical chunk 2 ks + h
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)asm_smem;
    u32x16 rbase;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t ch = (uint32_t)(((2 * ks + h) ^ ((r >> 1) & 7)) << 4);
            rbase[st * 4 + ks] = smem_base + st * STAGE + BM * 128 + (wn * WN + r) * 128 + ch;      // W rows (MFMA A operand)
            rbase[8 + st * 4 + ks] = smem_base + st * STAGE + (wm * WM + r) * 128 + ch;             // activation rows (B operand)
        }
    if constexpr (REGSTAGE) {        // LDS write address of this lane's 16 bytes of piece 0 (pieces add 4096 j, W adds BM * 128), stage 0 / 1
        const int row0 = 8 * wave + lr;
        dma1[0] = smem_base + (uint32_t)(row0 * 128 + ((pc ^ ((row0 >> 1) & 7)) << 4));
        dma1[1] = dma1[0] + (uint32_t)STAGE;
    }
    const uint64_t ap = (uint64_t)(uintptr_t)g.A, wp = (uint64_t)(uintptr_t)g.W;
    const u32x4 ra = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ap), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ap >> 32)) & 0xffffu, 0x80000000u, 0x00020000u};
    const u32x4 rw = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wp >> 32)) & 0xffffu, 0x80000000u, 0x00020000u};
    const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 1024u));
    const int nk = __builtin_amdgcn_readfirstlane(g.K / 64);

    f32x32 c[10];
    AsmLoop<BM, BN, WGM, WGN>::run(c, rbase, dma0, dma1, ra, rw, nk, 0u, 0u, ldsw);

    // epilogue: accumulator tile (nb, mb) = registers of D = W_frag x A_frag: lane column = output row m, register i =
    // output column 8 (i >> 2) + 4 h + (i & 3) of the 32-wide block: four consecutive columns per register quad
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nb = t / MB, mb = t - nb * MB;
        const int m = m0 + wm * WM + mb * 32 + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int n = n0 + wn * WN + nb * 32 + 8 * g4 + 4 * h;
            if (n >= g.N) continue;
            const int e = (t & 1) * 16 + 4 * g4;
            float v[4] = {c[t >> 1][e], c[t >> 1][e + 1], c[t >> 1][e + 2], c[t >> 1][e + 3]};
            epilogue<bf16_t, EPI>(g, m, n, v);
        }
    }
}

#endif  // LTX_EXPERIMENTS

// ---- the 16x16x32 kernel (plan family asm16:*, or LTX_GEMM_ASM=16 to force it; tools/gen_gemm_asm.py gen16).  The vendor library's
// kernel for these shapes, disassembled, is this structure - four waves of 128 x 128, LDS-DMA staging - with the 16x16x32 MFMA,
// the shape that holds a higher clock at equal cycles per FLOP.  Fragment = 16 rows x 32 k: lane (rr = lane & 15, q = lane >> 4)
// reads row rr of a 16-row block, logical chunk 4 half + q; accumulator block (nb, mb) = D = W_frag x A_frag: the lane holds
// output row m = mb * 16 + rr and the four consecutive columns nb * 16 + 4 q .. + 3.  Same k order as gemm_big: same bits.
template <int BM, int BN, int WGM, int WGN> struct AsmLoop16;
// run: the K loop; store(p, wm, ...): the wave's accumulator blocks that belong to row pass p of the epilogue -> LDS
template <> struct AsmLoop16<256, 256, 2, 2> {
    template <typename... T> static __device__ __forceinline__ void run(f32x32 (&c)[10], const u32x8& rb, const u32x16& d0, const u32x2&, T&&... t) {
        gemm_asm16_loop_256_256(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], rb, d0, t...); }
    static __device__ __forceinline__ void store(int p, int wm, const f32x32 (&c)[10], const u32x8& ad) {
        if (wm == p) gemm_asm16_store_256_256_p0(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], ad); }
};
template <> struct AsmLoop16<160, 256, 1, 4> {
    template <typename... T> static __device__ __forceinline__ void run(f32x32 (&c)[10], const u32x8& rb, const u32x16& d0, const u32x2&, T&&... t) {
        gemm_asm16_loop_160_256(c[0], c[1], c[2], c[3], c[4], rb, d0, t...); }
    static __device__ __forceinline__ void store(int p, int, const f32x32 (&c)[10], const u32x8& ad) {
        if (p == 0) gemm_asm16_store_160_256_p0(c[0], c[1], c[2], c[3], c[4], ad); else gemm_asm16_store_160_256_p1(c[0], c[1], c[2], c[3], c[4], ad); }
    // the loop that also requests the lane's 20 residual rows, two per K-step over its first ten K-steps (needs >= 12 K-steps)
    template <typename... T> static __device__ __forceinline__ void run_res(f32x32 (&c)[10], u32x16 (&res)[5], const u32x2& rvoff, const u32x4& rres, uint32_t rstride,
                                                                            const u32x8& rb, const u32x16& d0, T&&... t) {
        gemm_asm16_loop_160_256_res(c[0], c[1], c[2], c[3], c[4], rb, d0, t..., res[0], res[1], res[2], res[3], res[4], rvoff, rres, rstride); }
};
template <> struct AsmLoop16<320, 256, 2, 2> {             // 320 accumulator registers: a[0:255] and v[192:255]
    template <typename... T> static __device__ __forceinline__ void run(f32x32 (&c)[10], const u32x8& rb, const u32x16& d0, const u32x2& d1, T&&... t) {
        gemm_asm16_loop_320_256(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], rb, d0, d1, t...); }
    static __device__ __forceinline__ void store(int p, int wm, const f32x32 (&c)[10], const u32x8& ad) {
        if (wm != (p >> 1)) return;
        if ((p & 1) == 0) gemm_asm16_store_320_256_p0(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], ad);
        else gemm_asm16_store_320_256_p1(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], ad); }
};

#ifdef GEMM_ASM_TRACE      // tools/gemm_asm_tune.py trace: per-wave cycle sums of the K-step's segments (loop generated with trace=1)
__device__ uint32_t g_asm16_trace[1024 * 4 * 16];   // per wave: 8 segment sums, then entry / loop start / loop end / exit (100 MHz clock) and HW_ID
#endif

// FOLD (EPI_BIAS / EPI_GELU): the consumer side of the norm fold (GemmArgs::rs_sq / cvec, kernels.h) - out = epi(r_m * acc + cvec[b][n]).
// The producer side (GemmArgs::C2, residual epilogues with RSQ) is a uniform run-time branch of those instantiations.
template <int BM, int BN, int WGM, int WGN, int EPI, bool RSQ = false, bool PF_R = false, bool FOLD = false>
__global__ __launch_bounds__(256, 1) void gemm_asm16_kernel(const GemmArgs g) {
    constexpr int WM = BM / WGM, WN = BN / WGN, MB = WM / 16, NB = WN / 16, NT = MB * NB, AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 128;
#ifdef GEMM_ASM_TRACE
    const uint32_t t_entry = (uint32_t)wall_clock64();
#endif
    // every kernel argument the set-up reads, requested together: left to the compiler they arrive in three dependent batches (one per
    // branch of the tile mapping), each a scalar-cache miss of a block that has nothing else to do yet
#ifndef ASM16_NO_KERNARG_BATCH
    asm volatile("" :: "s"(g.A), "s"(g.W), "s"(g.M), "s"(g.N), "s"(g.K), "s"(g.lda), "s"(g.xcd_remap), "s"(g.group_m), "s"(g.rows_per_batch), "s"(gridDim.x));
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, q = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;
    const int ntn = (g.N + BN - 1) / BN;
    int bid = blockIdx.x;
    if (g.xcd_remap) {
        const int nblk = (int)gridDim.x, qq = nblk >> 3, r8 = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r8 ? x * (qq + 1) : r8 * (qq + 1) + (x - r8) * qq) + i;
    }
    int mt, nt;
    if (g.group_m > 1) {
        const int ntm = (g.M + BM - 1) / BM, gsz = g.group_m * ntn;
        const int grp = bid / gsz, w = bid - grp * gsz, gm0 = grp * g.group_m;
        const int rows = ntm - gm0 < g.group_m ? ntm - gm0 : g.group_m;
        nt = w / rows; mt = gm0 + (w - nt * rows);
    } else { mt = bid / ntn; nt = bid - mt * ntn; }
    const int m0 = mt * BM, n0 = nt * BN;
    // (norm fold: the tile rows' partial sums of squares are requested first thing - see in front of the K loop)
    static_assert(!FOLD || BM <= 256 || BM == 320, "norm fold: 64 extra rows = 256 (row, group) units");
    f32x4 fpt[4], fpx = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FOLD) {
        auto group_of = [&](int row, int q4) {
            const int m = m0 + row < g.M ? m0 + row : g.M - 1;
            const bool in = 4 * q4 < g.rs_n;
            const f32x4 u = *reinterpret_cast<const f32x4*>(g.rs_sq + (int64_t)m * g.rs_n + (in ? 4 * q4 : g.rs_n - 4));
            return in ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        const int row = tid < BM ? tid : BM - 1;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) fpt[q4] = group_of(row, q4);
        if constexpr (BM > 256) fpx = group_of(256 + (tid >> 2), tid & 3);
    }
    const int lr = lane >> 3, pc = lane & 7;
    u32x16 dma0; u32x2 dma1 = {0x80000000u, 0x80000000u};    // source offsets of the wave's AI + BI (<= 18) pieces of a K-step
#pragma unroll
    for (int j = 0; j < 16; ++j) dma0[j] = 0x80000000u;
    auto set_piece = [&](int j, uint32_t v) { if (j < 16) dma0[j] = v; else dma1[j - 16] = v; };
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 8 * (j * 4 + wave) + lr;
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
#ifdef ASM16_NOSWZ        // timing ablation (wrong results): lanes of a piece read ascending addresses
        set_piece(j, ((uint32_t)m * (uint32_t)g.lda + (uint32_t)pc * 8u) * 2u);
#else
        set_piece(j, ((uint32_t)m * (uint32_t)g.lda + (uint32_t)(pc ^ ((row >> 1) & 7)) * 8u) * 2u);
#endif
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (j * 4 + wave) + lr;
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
#ifdef ASM16_NOSWZ
        set_piece(AI + j, ((uint32_t)n * (uint32_t)g.K + (uint32_t)pc * 8u) * 2u);
#else
        set_piece(AI + j, ((uint32_t)n * (uint32_t)g.K + (uint32_t)(pc ^ ((row >> 1) & 7)) * 8u) * 2u);
#endif
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)asm_smem;
    u32x8 rbase;                                               // W [stage][half] then A [stage][half]
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const uint32_t ch = (uint32_t)(((4 * kh + q) ^ ((rr >> 1) & 7)) << 4);
            rbase[st * 2 + kh] = smem_base + st * STAGE + BM * 128 + (wn * WN + rr) * 128 + ch;
            rbase[4 + st * 2 + kh] = smem_base + st * STAGE + (wm * WM + rr) * 128 + ch;
        }
    const uint64_t ap = (uint64_t)(uintptr_t)g.A, wp = (uint64_t)(uintptr_t)g.W;
    const u32x4 ra = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ap), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ap >> 32)) & 0xffffu, 0x80000000u, 0x00020000u};
    const u32x4 rw = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wp >> 32)) & 0xffffu, 0x80000000u, 0x00020000u};
    const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 1024u));
    const int nk = __builtin_amdgcn_readfirstlane(g.K / 64);
    f32x32 c[10];
    // Residual tile requested INSIDE the K loop (PF_R: 160-row tiles, at least 12 K-steps; 80 registers per lane that the loop
    // leaves free).  A one-round grid runs its blocks in lockstep: every block reaches its epilogue together and the 20 MB residual
    // read of the whole launch is a burst that nothing overlaps - in-kernel stamps (tools/gemm2048_trace.py,
    // profiles/r5b_gemm2048_trace.jsonl) put the epilogue's row passes at 2.6 us with a bias only and 5.8 / 6.9 us with a residual /
    // gate + residual, of a 35-40 us block.  Requesting the whole tile in front of the loop only moved that burst to the start
    // (measured: no change); the generated loop spreads it, two rows of every lane per K-step over the first ten K-steps, between
    // the fragment reads and the first barrier.  Loads return in order, so the loop's counted waits are raised by what they skip.
    constexpr bool HAS_R = EPI == EPI_GATE_RESID || EPI == EPI_RESID;
    constexpr int RP = BM == 256 ? 128 : 80;               // tile rows per pass of the wide epilogue: RP KiB of f32 must fit the two stages
    static_assert(!PF_R || (HAS_R && BM == 160 && BN == 256), "residual prefetch: the 160 x 256 tile's residual epilogues");
    u32x16 pf[5];                                          // 20 rows x (first | second) column group x 2 dwords, in that order
    u32x2 pf_voff = {0x80000000u, 0x80000000u}; u32x4 pf_rsrc = {0u, 0u, 0u, 0u}; uint32_t pf_stride = 0u;
    if constexpr (PF_R) {
        const int rows_valid_pf = g.M - m0 < BM ? g.M - m0 : BM;
        const uint64_t rp = (uint64_t)(uintptr_t)(reinterpret_cast<const bf16_t*>(g.resid) + (int64_t)m0 * g.ldr + n0);
        pf_rsrc = (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rp >> 32)) & 0xffffu,
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)(rows_valid_pf - 1) * (uint32_t)g.ldr + 256u) * 2u)), 0x00020000u};
        const int cg_pf = tid & 31;
        const uint32_t row0 = (uint32_t)(tid >> 5) * (uint32_t)g.ldr * 2u;
        pf_voff[0] = n0 + 4 * cg_pf < g.N ? row0 + (uint32_t)(8 * cg_pf) : 0x80000000u;
        pf_voff[1] = n0 + 4 * cg_pf + 128 < g.N ? row0 + (uint32_t)(8 * cg_pf + 256) : 0x80000000u;
        pf_stride = (uint32_t)__builtin_amdgcn_readfirstlane((int)(8u * (uint32_t)g.ldr * 2u));
    }
    // Norm fold, consumer side: the tile rows' partial sums of squares were written by the previous launch on other XCDs (a trip past
    // the L2).  Read after the K loop they cost the block 3 - 5 us of exposed latency per tile (kernel trace: qkv 98.5 -> 109.2 us,
    // ff1 133.7 -> 143.9); requested in front of the loop and summed behind it they held 20 registers across it - three more than the
    // 320-row tile has, spilled and reloaded at the top of the epilogue.  Now: requested first thing in the kernel (above), summed
    // HERE - their latency ran under the set-up - and the rows' 1 / rms parked in the LDS behind the two stages, which the K loop
    // never touches (the epilogue's first barrier publishes it): nothing of the fold lives across the loop.  One row per thread, at
    // most 16 partials per row (D <= 2048), groups past rs_n re-read the last one and are zeroed - no branch around a load; rows
    // 256 .. BM - 1 of the 320-row tile: four lanes per row, one group of four partials each, the ascending sum (the expression of
    // rownorm_presum_kernel) then runs through the four lanes in turn.
    if constexpr (FOLD) {
        // (the compiler otherwise sums right behind the loads and waits for them in front of the set-up)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(fpt[0]), "+v"(fpt[1]), "+v"(fpt[2]), "+v"(fpt[3]), "+v"(fpx), "+v"(dma0), "+v"(dma1), "+v"(rbase));   // (the set-up's results: it stays in front)
        float* rl = reinterpret_cast<float*>(asm_smem + 2 * STAGE);
        if (tid < BM) {
            float ss = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { ss += fpt[q4][0]; ss += fpt[q4][1]; ss += fpt[q4][2]; ss += fpt[q4][3]; }
            rl[tid] = 1.0f / sqrtf(ss * (1.0f / (float)g.rs_D) + g.rs_eps);
        }
        if constexpr (BM > 256) {                          // lanes 4 i .. 4 i + 3 hold groups 0 .. 3 of row 256 + tid / 4: the same ascending chain
            float run = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float before = q4 ? __shfl(run, (lane & ~3) + q4 - 1) : 0.f;
                if ((tid & 3) == q4) { run = before; run += fpx[0]; run += fpx[1]; run += fpx[2]; run += fpx[3]; }
            }
            if ((tid & 3) == 3) rl[256 + (tid >> 2)] = 1.0f / sqrtf(run * (1.0f / (float)g.rs_D) + g.rs_eps);
        }
    }
#ifdef ASM16_STAGGER      // experiment (with a stagger=1 loop): block-dependent start position in K, wrapping at the end
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((unsigned)blockIdx.x % ASM16_STAGGER) * (unsigned)(nk / ASM16_STAGGER) * 128u));
#else
    const uint32_t k0 = 0u;
#endif
#ifdef GEMM_ASM_TRACE
    uint32_t tr[8];
    const uint32_t t_loop0 = (uint32_t)wall_clock64();
    AsmLoop16<BM, BN, WGM, WGN>::run(c, rbase, dma0, dma1, ra, rw, nk, k0, k0, ldsw, (uint32_t)__builtin_amdgcn_readfirstlane(g.K * 2), tr);
    const uint32_t t_loop1 = (uint32_t)wall_clock64();
#else
    if constexpr (PF_R) AsmLoop16<BM, BN, WGM, WGN>::run_res(c, pf, pf_voff, pf_rsrc, pf_stride, rbase, dma0, ra, rw, nk, k0, k0, ldsw, (uint32_t)__builtin_amdgcn_readfirstlane(g.K * 2));
    else AsmLoop16<BM, BN, WGM, WGN>::run(c, rbase, dma0, dma1, ra, rw, nk, k0, k0, ldsw, (uint32_t)__builtin_amdgcn_readfirstlane(g.K * 2));
#endif
    // Wide epilogue.  One wave per SIMD hides no latency and the 256 accumulators leave no registers to unroll into, so the
    // fragment-wise epilogue of gemm_big (serial bias load -> wait -> 8-byte store per block) took 8-10 us of a 60 us block
    // (tools/gemm_asm_tune.py trace).  Here the f32 accumulators go through LDS - free after the K loop - in row passes of
    // RP tile rows: the waves that own rows of the pass write their blocks (16-byte chunk c of row r at chunk c ^ (r & 15):
    // conflict-free for the 8-lane write groups and the 16-lane read groups), then ALL threads walk the pass row-major: thread
    // (row = tid / 32 + 8 k, cg = tid % 32) owns columns 4 cg .. + 3 and 128 + 4 cg .. + 3 of its rows, keeps its bias / gate
    // vectors in registers, reads the residual straight from global memory (8 bytes per lane, 256 contiguous bytes per half row)
    // and applies bias / GELU / gate / residual with the expressions of epilogue(): same bits.
    // (No fragment-wise fallback in this kernel: extracting elements of the accumulator tuples makes the compiler copy all 256
    // to VGPRs and spill; shapes that do not meet the conditions - ltx_gemm_asm16_epilogue_ok - stay on gemm_big.)
    {
        static_assert(BN == 256, "thread -> column map below");
        static_assert(RP * 1024 <= 2 * STAGE && BM % RP == 0 && RP % 16 == 0 && RP % 8 == 0, "pass geometry");
        constexpr bool RSQ_COMPACT = RSQ && 2 * STAGE - RP * 1024 >= RP * 2 * 144;   // GemmArgs::rowsq leaves in the LDS behind a pass
        const int r0 = tid >> 5, cg = tid & 31;
        int nA = n0 + 4 * cg, nB = nA + 128;
        const bool okA = nA < g.N, okB = nB < g.N;         // N % 8 == 0: a 4-column group is inside or outside as a whole
        if (!okA) nA = 0;
        if (!okB) nB = 0;
        float bA[4] = {0.f, 0.f, 0.f, 0.f}, bB[4] = {0.f, 0.f, 0.f, 0.f};
        static_assert(!FOLD || EPI == EPI_BIAS || EPI == EPI_GELU, "norm fold, consumer side: bias / GELU epilogues");
        if constexpr (FOLD) {
            const float* cp = g.cvec + (int64_t)(m0 / g.rows_per_batch) * g.cvec_stride;
            const f32x4 cA4 = *reinterpret_cast<const f32x4*>(cp + nA), cB4 = *reinterpret_cast<const f32x4*>(cp + nB);
#pragma unroll
            for (int i = 0; i < 4; ++i) { bA[i] = cA4[i]; bB[i] = cB4[i]; }
        } else
        if (g.bias) { load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + nA, bA); load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + nB, bB); }
        // Output and residual tiles through buffer descriptors based at the tile's first row: 32-bit offsets, rows beyond M
        // and column groups beyond N fall out of range (stores dropped, loads return 0) - no branch around a memory operation.
        // A tile lies in ONE column segment of a segmented output (segment width % 256 == 0: ltx_gemm_asm_eligible).
        bf16_t* Ct = reinterpret_cast<bf16_t*>(g.C);
        {
            int nc = n0;
            if (g.c_seg_shift) { const int sg = n0 >> g.c_seg_shift; Ct += sg * g.c_seg_stride; nc -= sg << g.c_seg_shift; }
            Ct += (int64_t)m0 * g.ldc + nc;
        }
        const int rows_valid = g.M - m0 < BM ? g.M - m0 : BM;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Ct, 0, (int)(((uint32_t)(rows_valid - 1) * (uint32_t)g.ldc + 256u) * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(g.resid ? g.resid : g.C) + (g.resid ? (int64_t)m0 * g.ldr + n0 : 0)), 0,
            (int)(g.resid ? ((uint32_t)(rows_valid - 1) * (uint32_t)g.ldr + 256u) * 2u : 0u), 0x00020000);
        const uint32_t cA = okA ? (uint32_t)(8 * cg) : 0x80000000u, cB = okB ? (uint32_t)(8 * cg + 256) : 0x80000000u;
        const int m_last = m0 + rows_valid - 1;
        const float fA = okA ? 1.f : 0.f, fB = okB ? 1.f : 0.f;
        const int m_split = (m0 / g.rows_per_batch + 1) * g.rows_per_batch;      // first row of the tile's second batch element
        const bool straddle = m_last >= m_split;                                  // (block-uniform)
        // Gate rows: the tile's first and LAST batch element are loaded here; a tile that straddles one boundary (rows_per_batch >= BM)
        // chooses per row.  (Round 6: the row loop used to re-load the gate behind an `if (!one_batch)` - never taken at one batch
        // element, but a load in the loop makes the compiler drain vmcnt, i.e. every output store of the rows before, at each step:
        // 22 vmcnt(0) per tile in the gate + residual epilogues against 1 - 2 in the others.)  Batch elements shorter than the
        // tallest tile are not this kernel's (ltx_gemm_asm16_fits).
        f32x4 gA = {}, gB = {}, gA1 = {}, gB1 = {};
        if constexpr (EPI == EPI_GATE_RESID) {
            const float* gp = g.gate + (int64_t)(m0 / g.rows_per_batch) * g.gate_stride;
            const float* gp1 = g.gate + (int64_t)(m_last / g.rows_per_batch) * g.gate_stride;
            gA = *reinterpret_cast<const f32x4*>(gp + nA); gB = *reinterpret_cast<const f32x4*>(gp + nB);
            gA1 = *reinterpret_cast<const f32x4*>(gp1 + nA); gB1 = *reinterpret_cast<const f32x4*>(gp1 + nB);
        }
        // Norm fold: the per-batch-row vectors (cvec of the consumer, 1 + scale of the producer) of a tile that straddles a batch
        // boundary (rows_per_batch >= BM: at most one - ltx_gemm_fold_ok) are BOTH loaded here and chosen per row: a load inside
        // the row loop, even behind a branch that is never taken, makes the compiler drain vmcnt - the output stores of the
        // previous rows - before every step (kernel trace: + 9 us per consumer launch, + 4 per producer).
        float bA1[4] = {0.f, 0.f, 0.f, 0.f}, bB1[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FOLD) {
            const float* cp1 = g.cvec + (int64_t)(m_last / g.rows_per_batch) * g.cvec_stride;
            const f32x4 cA4 = *reinterpret_cast<const f32x4*>(cp1 + nA), cB4 = *reinterpret_cast<const f32x4*>(cp1 + nB);
#pragma unroll
            for (int i = 0; i < 4; ++i) { bA1[i] = cA4[i]; bB1[i] = cB4[i]; }
        }
        // producer side: 1 + scale of the NEXT norm for this thread's columns (a second output of the residual epilogues)
        const bool fold_out = HAS_R && RSQ && g.C2 != nullptr;
        float s2A[4] = {1.f, 1.f, 1.f, 1.f}, s2B[4] = {1.f, 1.f, 1.f, 1.f}, s2A1[4] = {1.f, 1.f, 1.f, 1.f}, s2B1[4] = {1.f, 1.f, 1.f, 1.f};
        __amdgpu_buffer_rsrc_t rc2 = rc;
        if constexpr (HAS_R && RSQ) {
            if (fold_out) {
                const float* sp = g.scale2 + (int64_t)(m0 / g.rows_per_batch) * g.scale2_stride;
                const float* sp1 = g.scale2 + (int64_t)(m_last / g.rows_per_batch) * g.scale2_stride;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(sp + nA), b4 = *reinterpret_cast<const f32x4*>(sp + nB);
                const f32x4 a41 = *reinterpret_cast<const f32x4*>(sp1 + nA), b41 = *reinterpret_cast<const f32x4*>(sp1 + nB);
#pragma unroll
                for (int i = 0; i < 4; ++i) { s2A[i] = 1.0f + a4[i]; s2B[i] = 1.0f + b4[i]; s2A1[i] = 1.0f + a41[i]; s2B1[i] = 1.0f + b41[i]; }
                rc2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(g.C2) + (int64_t)m0 * g.ldc + n0, 0, (int)(((uint32_t)(rows_valid - 1) * (uint32_t)g.ldc + 256u) * 2u), 0x00020000);
            }
        }
        // value of one 4-column group: the expressions of epilogue() (gemm_common.h), residual already loaded
        auto finish = [&](auto str_tag, const f32x4& acc, const float* bias4, const float* bias41, const f32x4& gate_one, const f32x4& gate_two, const bf16x4& res, int n, int m, float* v, float rrow) {
            constexpr bool STR = decltype(str_tag)::value;
            if constexpr (FOLD) {
                if constexpr (!STR) {                               // (uniform: the whole tile in one batch element - no per-element selects)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(rrow, acc[i], bias4[i]);
                } else {
                    const bool second = m >= m_split;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(rrow, acc[i], second ? bias41[i] : bias4[i]);
                }
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[i] + bias4[i];
            }
            if constexpr (EPI == EPI_GELU) {
                gelu_tanh4(v);
            } else if constexpr (HAS_R) {
                const float r[4] = {(float)res[0], (float)res[1], (float)res[2], (float)res[3]};
                if constexpr (EPI == EPI_GATE_RESID) {
                    f32x4 gt = gate_one;
                    if constexpr (STR) {
                        const bool second = m >= m_split;
#pragma unroll
                        for (int i = 0; i < 4; ++i) gt[i] = second ? gate_two[i] : gate_one[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(gt[i], v[i], r[i]);       // ONE rounding, spelled out: every kernel that finishes these rows must agree
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += r[i];
                }
            }
        };
        u32x8 ad;                                          // LDS address of (local row rr, column block nb) for the stores below
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const int chunk = (wn * WN + (nb < NB ? nb : 0) * 16 + 4 * q) >> 2;
            ad[nb] = smem_base + (uint32_t)(rr * 1024 + ((chunk ^ rr) << 4));
        }
#ifdef GEMM_ASM_TRACE
        uint32_t t_e0 = 0, t_e1 = 0, t_e2 = 0;
#endif
        // The row passes, in two copies: tiles inside one batch element (all of them at one video per call) run without the
        // per-row choice between the two batch elements' vectors (the compiler turns a branch on the block-uniform flag inside the
        // row loop into per-element selects: 16 v_cndmask per row on a one-wave-per-SIMD epilogue, 2.4 us per tile).
        auto run_passes = [&](auto str_tag) {
        constexpr bool STR = decltype(str_tag)::value;
#pragma unroll
        for (int p = 0; p < BM / RP; ++p) {
            __syncthreads();                               // K loop (or the previous pass) is done with the LDS image in every wave
#ifdef GEMM_ASM_TRACE
            if (p == 0) t_e0 = (uint32_t)wall_clock64();
#endif
            // the wave's accumulator blocks of this pass, AGPR -> LDS (generated: gemm_asm16_store_*; the tuples stay opaque here)
            AsmLoop16<BM, BN, WGM, WGN>::store(p, wm, c, ad);
#ifdef GEMM_ASM_TRACE
            if (p == 0) t_e1 = (uint32_t)wall_clock64();
#endif
            __syncthreads();
#ifdef GEMM_ASM_TRACE
            if (p == 0) t_e2 = (uint32_t)wall_clock64();
#endif
            // UN rows per step: all LDS reads and residual loads of the step are issued before the first use (rows beyond M
            // are computed on clamped addresses and not stored - no branch in front of a load)
            constexpr int UN = RP % 32 == 0 ? 4 : 5;
            static_assert((RP / 8) % UN == 0, "row steps");
#pragma unroll (PF_R ? RP / 8 / UN : 1)       // (round 6, measured: unrolled everywhere q2 -0.25 us, qkv +0.3, ff1's GELU rows +3 us)
            for (int k0 = 0; k0 < RP / 8; k0 += UN) {
                f32x4 vA[UN], vB[UN];
                bf16x4 qA[UN] = {}, qB[UN] = {};
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    const int row = r0 + 8 * (k0 + j);
                    const unsigned char* rowp = asm_smem + row * 1024;
                    vA[j] = *reinterpret_cast<const f32x4*>(rowp + ((cg ^ (row & 15)) << 4));
                    vB[j] = *reinterpret_cast<const f32x4*>(rowp + (((cg + 32) ^ (row & 15)) << 4));
                    if constexpr (PF_R) {                  // requested inside the K loop: lane row i = tile row r0 + 8 i, i = p * RP / 8 + k0 + j
                        const int i = p * (RP / 8) + k0 + j;
                        qA[j] = __builtin_bit_cast(bf16x4, (u32x2){pf[(2 * i) / 16][(2 * i) % 16], pf[(2 * i) / 16][(2 * i) % 16 + 1]});
                        qB[j] = __builtin_bit_cast(bf16x4, (u32x2){pf[(40 + 2 * i) / 16][(40 + 2 * i) % 16], pf[(40 + 2 * i) / 16][(40 + 2 * i) % 16 + 1]});
                    } else if constexpr (HAS_R) {
                        const uint32_t ro = (uint32_t)(p * RP + row) * (uint32_t)g.ldr * 2u;
                        qA[j] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rres, (int)(ro + cA), 0, 0));
                        qB[j] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rres, (int)(ro + cB), 0, 0));
                    }
                }
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    const int trow = p * RP + r0 + 8 * (k0 + j);
                    const int m = m0 + trow < g.M ? m0 + trow : g.M - 1;       // gate row of a straddling tile only
                    float oA[4], oB[4];
                    float rrow = 1.f;
                    if constexpr (FOLD) rrow = reinterpret_cast<const float*>(asm_smem + 2 * STAGE)[trow];
                    finish(str_tag, vA[j], bA, bA1, gA, gA1, qA[j], nA, m, oA, rrow);
                    finish(str_tag, vB[j], bB, bB1, gB, gB1, qB[j], nB, m, oB, rrow);
                    const uint32_t co = (uint32_t)trow * (uint32_t)g.ldc * 2u;
                    const bf16x4 pA = {(bf16_t)oA[0], (bf16_t)oA[1], (bf16_t)oA[2], (bf16_t)oA[3]}, pB = {(bf16_t)oB[0], (bf16_t)oB[1], (bf16_t)oB[2], (bf16_t)oB[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pA), rc, (int)(co + cA), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pB), rc, (int)(co + cB), 0, 0);
                    if constexpr (HAS_R && RSQ) {
                        if (fold_out) {                        // h (.) (1 + scale) of the rows as stored: the A operand of the layer behind the next norm
                            float tA[4] = {s2A[0], s2A[1], s2A[2], s2A[3]}, tB[4] = {s2B[0], s2B[1], s2B[2], s2B[3]};
                            if constexpr (STR) {
                                const bool second = m >= m_split;
#pragma unroll
                                for (int i = 0; i < 4; ++i) { tA[i] = second ? s2A1[i] : s2A[i]; tB[i] = second ? s2B1[i] : s2B[i]; }
                            }
                            const bf16x4 hA = {(bf16_t)((float)pA[0] * tA[0]), (bf16_t)((float)pA[1] * tA[1]), (bf16_t)((float)pA[2] * tA[2]), (bf16_t)((float)pA[3] * tA[3])};
                            const bf16x4 hB = {(bf16_t)((float)pB[0] * tB[0]), (bf16_t)((float)pB[1] * tB[1]), (bf16_t)((float)pB[2] * tB[2]), (bf16_t)((float)pB[3] * tB[3])};
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hA), rc2, (int)(co + cA), 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hB), rc2, (int)(co + cB), 0, 0);
                        }
                    }
                    if constexpr (RSQ) {
                        // GemmArgs::rowsq: the leaves of this thread's two 4-column groups (values as stored) go into the first
                        // dword of its OWN two 16-byte slots of the row, which it has just consumed - nobody else reads them
                        const int row = r0 + 8 * (k0 + j);
                        const float lA = ltx_rowsq_leaf(pA) * fA, lB = ltx_rowsq_leaf(pB) * fB;   // fA / fB: 1, or 0 for a column group beyond N (no branch)
                        if constexpr (RSQ_COMPACT) {       // LDS left over behind the pass: 36-dword rows of 32 leaves per (row, half)
                            float* lf = reinterpret_cast<float*>(asm_smem + RP * 1024) + (2 * row) * 36 + cg;
                            lf[0] = lA; lf[36] = lB;
                        } else {                           // no room (256 x 256 tile): first dword of the thread's own two consumed slots
                            float* rowf = reinterpret_cast<float*>(asm_smem + row * 1024);
                            rowf[(cg ^ (row & 15)) << 2] = lA;
                            rowf[((cg + 32) ^ (row & 15)) << 2] = lB;
                        }
                    }
                }
            }
            if constexpr (RSQ) {
                // one thread per (row, 128-column half): the 32 leaves summed in ascending column order (the canonical order of
                // ltx_launch_rowsq, rownorm.hip: same bits whichever kernel produced the matrix)
                __syncthreads();
                const int ng = (g.N + 127) >> 7;
#pragma unroll 1
                for (int pi = tid; pi < RP * 2; pi += 256) {
                    const int row = pi >> 1, half = pi & 1, sw = row & 15;
                    float sum;
                    if constexpr (RSQ_COMPACT) {
                        const f32x4* lf = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(asm_smem + RP * 1024) + pi * 36);
                        f32x4 t[8];
#pragma unroll
                        for (int c4 = 0; c4 < 8; ++c4) t[c4] = lf[c4];
                        sum = t[0][0]; sum += t[0][1]; sum += t[0][2]; sum += t[0][3];
#pragma unroll
                        for (int c4 = 1; c4 < 8; ++c4) { sum += t[c4][0]; sum += t[c4][1]; sum += t[c4][2]; sum += t[c4][3]; }
                    } else {
                        const float* rowf = reinterpret_cast<const float*>(asm_smem + row * 1024);
                        sum = rowf[((32 * half) ^ sw) << 2];
#pragma unroll
                        for (int c2 = 1; c2 < 32; ++c2) sum += rowf[((c2 + 32 * half) ^ sw) << 2];
                    }
                    const int trow = p * RP + row, ncol = n0 + 128 * half;
                    if (m0 + trow < g.M && ncol < g.N) g.rowsq[(int64_t)(m0 + trow) * ng + (ncol >> 7)] = sum;
                }
            }
        }
        };
        if (straddle) run_passes(std::true_type{}); else run_passes(std::false_type{});
#ifdef GEMM_ASM_TRACE
        if (lane == 0 && blockIdx.x < 1024) { uint32_t* o = g_asm16_trace + (blockIdx.x * 4 + wave) * 16; o[13] = t_e0; o[14] = t_e1; o[15] = t_e2; }
#endif
    }
#ifdef GEMM_ASM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && blockIdx.x < 1024) {
        uint32_t* o = g_asm16_trace + (blockIdx.x * 4 + wave) * 16;
        for (int i = 0; i < 8; ++i) o[i] = tr[i];
        uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[8] = t_entry; o[9] = t_loop0; o[10] = t_loop1; o[11] = (uint32_t)wall_clock64(); o[12] = hwid;
    }
#endif
}

// ---- conv mode of the 16x16x32 loop (round 5): 3x3x3 conv3d as an implicit GEMM with the activation rows re-staged per tap, tile
// 256 x 256 - the VAE's 1024-channel mid block (M = 4992 voxels x N = 1024 x K = 27 x 1024: 80 tiles, cut into the shape rule's three
// K ranges = one frame tap each), conv_in and the first upsampler.  Same K-step order (frame tap, 64-channel slice, in-plane tap),
// same K partition and canonical sum of the parts as gemm_big's conv mode: bit-identical to it (tests/test_gpu_gemm_asm.py).
// gemm_big's 16-wave 256 x 256 tile ran these at 0.38 of the MFMA peak (profiles/r4zf_*: 300 us per mid-block conv).
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_asm16_conv_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, WGN = 2, WM = 128, WN = 128, NB = 8, AI = 8, BI = 8, STAGE = (BM + BN) * 128, RP = 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, q = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;
    const int ntn = (g.N + BN - 1) / BN;
    const int sf = g.sk_sf > 1 ? g.sk_sf : 1;
    const int tile = (int)blockIdx.x / sf, part = (int)blockIdx.x - tile * sf;      // the parts of a tile are neighbours in the grid
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int KC = g.Cin / 64, nk = g.ntaps * KC;
    const int kt0 = (int)((int64_t)part * nk / sf), kt1 = (int)((int64_t)(part + 1) * nk / sf);
    // fetch state of the first K-step of the range
    const int per_it = 9 * KC, it0 = kt0 / per_it, r2 = kt0 - it0 * per_it, kc0 = r2 / 9, hw0 = r2 - kc0 * 9, ih0 = hw0 / 3, iw0 = hw0 - ih0 * 3;
    // the descriptor is based a little before the tile's first voxel (gemm_big.hip): every tap of every row lies at a non-negative
    // 32-bit offset whatever the size of the tensor
    const int64_t m_base = m0 > 2 * g.H * g.Wd + g.Wd + 1 ? (int64_t)m0 - (2 * g.H * g.Wd + g.Wd + 1) : 0;
    const int lr = lane >> 3, pc = lane & 7;
    const uint32_t frame_bytes = (uint32_t)g.H * (uint32_t)g.Wd * (uint32_t)g.Cin * 2u;
    u32x16 dma0; u32x8 ab[3], vmask;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 8 * (j * 4 + wave) + lr;
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
        const int w = m % g.Wd, t1 = m / g.Wd, h = t1 % g.H, t = (t1 / g.H) % g.T;
        vmask[j] = (uint32_t)((h > 0 ? 1 : 0) | 2 | (h < g.H - 1 ? 4 : 0) | (w > 0 ? 8 : 0) | 16 | (w < g.Wd - 1 ? 32 : 0));
        const uint32_t a_off = ((uint32_t)(m - m_base) * (uint32_t)g.Cin + (uint32_t)(pc ^ ((row >> 1) & 7)) * 8u) * 2u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {                      // frame taps it0, it0 + 1, it0 + 2 (the last ones unused past the third)
            int tt = t + (it0 + i) - g.pad_t; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);      // replicate pad on T (vae.rs:374-413)
            ab[i][j] = a_off + (uint32_t)(tt - t) * frame_bytes;
        }
        dma0[j] = 0x80000000u;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (j * 4 + wave) + lr;
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
        dma0[AI + j] = ((uint32_t)n * (uint32_t)g.K + (uint32_t)(pc ^ ((row >> 1) & 7)) * 8u) * 2u;
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)asm_smem;
    u32x8 rbase;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const uint32_t ch = (uint32_t)(((4 * kh + q) ^ ((rr >> 1) & 7)) << 4);
            rbase[st * 2 + kh] = smem_base + st * STAGE + BM * 128 + (wn * WN + rr) * 128 + ch;
            rbase[4 + st * 2 + kh] = smem_base + st * STAGE + (wm * WM + rr) * 128 + ch;
        }
    const uint64_t ap = (uint64_t)(uintptr_t)(reinterpret_cast<const bf16_t*>(g.A) + m_base * g.Cin), wp = (uint64_t)(uintptr_t)g.W;
    const u32x4 ra = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ap), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ap >> 32)) & 0xffffu, 0x80000000u, 0x00020000u};
    const u32x4 rw = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wp >> 32)) & 0xffffu, 0x80000000u, 0x00020000u};
    const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 1024u));
    const uint32_t cinb = (uint32_t)g.Cin * 2u, rowb = (uint32_t)g.Wd * cinb, nk2 = (uint32_t)g.N * (uint32_t)g.K * 2u;
    auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    f32x32 c[10];
    gemm_asm16_conv_loop_256_256(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], rbase, dma0, ab[0], ab[1], ab[2], vmask, ra, rw,
                                 __builtin_amdgcn_readfirstlane(kt1 - kt0), ldsw,
                                 sgpr((uint32_t)iw0), sgpr((uint32_t)ih0), sgpr((uint32_t)kc0), sgpr((uint32_t)((ih0 - 1) * g.Wd + (iw0 - 1)) * cinb), sgpr((uint32_t)kc0 * 128u),
                                 sgpr(((uint32_t)(it0 * 9 + hw0) * (uint32_t)g.N * (uint32_t)g.K + (uint32_t)kc0 * 64u) * 2u),
                                 sgpr(cinb), sgpr(nk2), sgpr(rowb - 3u * cinb), sgpr(0u - 3u * rowb), sgpr(128u - 9u * nk2), sgpr(9u * nk2 - (uint32_t)KC * 128u), sgpr((uint32_t)KC));

    // Epilogue: the f32 accumulators pass through LDS in two row passes of 128 tile rows (as in the linear kernel); every thread
    // then owns two 4-column groups of its rows and hands them to the shared epilogue() - the arithmetic of every other conv kernel.
    // A split tile's parts leave their f32 rows in slabs; the part that draws the last ticket adds them in part order (the canonical
    // sum of gemm_big.hip) and finishes.
    const int r0 = tid >> 5, cg = tid & 31;
    const int nA = n0 + 4 * cg, nB = nA + 128;
    u32x8 ad;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
        const int chunk = (wn * WN + (nb < NB ? nb : 0) * 16 + 4 * q) >> 2;
        ad[nb] = smem_base + (uint32_t)(rr * 1024 + ((chunk ^ rr) << 4));
    }
    float* slab = sf > 1 ? g.sk_ws + ((int64_t)tile * sf + part) * (BM * BN) : nullptr;
    // Round 6: no load inside the row loops.  The bias of the thread's two column groups is loaded once; the depth-to-space
    // residual's gathers of four rows are issued together.  (A load per row waited vmcnt(0), i.e. also for the row before's stores:
    // 64 store -> load round trips per thread and tile.)  epilogue() is handed a GemmArgs without bias / residual: it places and stores.
    float bA4[4] = {0.f, 0.f, 0.f, 0.f}, bB4[4] = {0.f, 0.f, 0.f, 0.f};
    const int nAc = nA < g.N ? nA : g.N - 4, nBc = nB < g.N ? nB : g.N - 4;       // (clamped: no condition in front of a load)
    if (g.bias) { load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + nAc, bA4); load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + nBc, bB4); }
    GemmArgs gs = g; gs.bias = nullptr; if (EPI == EPI_D2S) gs.resid = nullptr;
    const bool d2s_res = EPI == EPI_D2S && g.resid != nullptr;
#pragma unroll
    for (int p = 0; p < BM / RP; ++p) {
        __syncthreads();
        AsmLoop16<256, 256, 2, 2>::store(p, wm, c, ad);
        __syncthreads();
        constexpr int UQ = 4;
#pragma unroll 1
        for (int k0 = 0; k0 < RP / 8; k0 += UQ) {
            f32x4 vA[UQ], vB[UQ];
            bf16_t xa[UQ][4], xb[UQ][4];
#pragma unroll
            for (int j = 0; j < UQ; ++j) {
                const int row = r0 + 8 * (k0 + j);
                const unsigned char* rowp = asm_smem + row * 1024;
                vA[j] = *reinterpret_cast<const f32x4*>(rowp + ((cg ^ (row & 15)) << 4));
                vB[j] = *reinterpret_cast<const f32x4*>(rowp + (((cg + 32) ^ (row & 15)) << 4));
            }
            if (sf > 1) {
#pragma unroll
                for (int j = 0; j < UQ; ++j) {
                    const int trow = p * RP + r0 + 8 * (k0 + j);
                    *reinterpret_cast<f32x4*>(slab + trow * BN + 4 * cg) = vA[j];
                    *reinterpret_cast<f32x4*>(slab + trow * BN + 128 + 4 * cg) = vB[j];
                }
                continue;
            }
            if constexpr (EPI == EPI_D2S) {
                if (d2s_res) {
                    const int nsub = g.d2s_sp ? 4 : 8;
                    const int sA = nAc / g.Cf, coA = nAc - sA * g.Cf, sB = nBc / g.Cf, coB = nBc - sB * g.Cf;
#pragma unroll
                    for (int j = 0; j < UQ; ++j) {
                        int m = m0 + p * RP + r0 + 8 * (k0 + j); if (m > g.M - 1) m = g.M - 1;
                        const bf16_t* xr = reinterpret_cast<const bf16_t*>(g.resid) + (int64_t)m * g.Cin;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { xa[j][i] = xr[((coA + i) % g.Cr) * nsub + sA]; xb[j][i] = xr[((coB + i) % g.Cr) * nsub + sB]; }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UQ; ++j) {
                const int m = m0 + p * RP + r0 + 8 * (k0 + j);
                if (m >= g.M) continue;
                float a4[4], b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a4[i] = vA[j][i] + bA4[i]; b4[i] = vB[j][i] + bB4[i]; }
                if constexpr (EPI == EPI_D2S) {
                    if (d2s_res) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { a4[i] += to_f32(xa[j][i]); b4[i] += to_f32(xb[j][i]); }
                    }
                }
                if (nA < g.N) epilogue<bf16_t, EPI>(gs, m, nA, a4);
                if (nB < g.N) epilogue<bf16_t, EPI>(gs, m, nB, b4);
            }
        }
    }
    if (sf > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(asm_smem);      // all LDS reads are behind the barrier above
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(g.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)sf - 1u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // back at zero for the next launch on this stream
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;
        const float* base = g.sk_ws + (int64_t)tile * sf * (BM * BN);
        // four rows per step, every load of the step (the parts' slab rows, the residual groups) issued before the first use: the
        // reducer works alone on its tile while the other parts' CUs are already idle
        constexpr int UR = 4;
#pragma unroll 1
        for (int k0 = 0; k0 < BM / 8; k0 += UR) {
            f32x4 vA[UR], vB[UR];
            bf16x4 qA[UR] = {}, qB[UR] = {};
#pragma unroll
            for (int j = 0; j < UR; ++j) {
                const int trow = r0 + 8 * (k0 + j);
                int m = m0 + trow; if (m > g.M - 1) m = g.M - 1;
                const float* rp = base + trow * BN + 4 * cg;
                vA[j] = *reinterpret_cast<const f32x4*>(rp); vB[j] = *reinterpret_cast<const f32x4*>(rp + 128);
                if constexpr (EPI == EPI_RESID) {
                    const bf16_t* rrow = reinterpret_cast<const bf16_t*>(g.resid) + (int64_t)m * g.ldr;
                    if (nA < g.N) qA[j] = *reinterpret_cast<const bf16x4*>(rrow + nA);
                    if (nB < g.N) qB[j] = *reinterpret_cast<const bf16x4*>(rrow + nB);
                }
            }
            for (int pq = 1; pq < sf; ++pq) {                    // ((s0 + s1) + s2) + ...: the order every plan of the shape uses
#pragma unroll
                for (int j = 0; j < UR; ++j) {
                    const float* rp = base + (int64_t)pq * (BM * BN) + (r0 + 8 * (k0 + j)) * BN + 4 * cg;
                    vA[j] += *reinterpret_cast<const f32x4*>(rp); vB[j] += *reinterpret_cast<const f32x4*>(rp + 128);
                }
            }
#pragma unroll
            for (int j = 0; j < UR; ++j) {
                const int m = m0 + r0 + 8 * (k0 + j);
                if (m >= g.M) continue;
                float a4[4] = {vA[j][0], vA[j][1], vA[j][2], vA[j][3]}, b4[4] = {vB[j][0], vB[j][1], vB[j][2], vB[j][3]};
                if constexpr (EPI == EPI_D2S) {
                    if (nA < g.N) epilogue<bf16_t, EPI>(g, m, nA, a4);
                    if (nB < g.N) epilogue<bf16_t, EPI>(g, m, nB, b4);
                } else {                                          // epilogue()'s expressions for bias / residual, residual already here, bias loaded once above
                    bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc;
                    if (nA < g.N) {
                        for (int i = 0; i < 4; ++i) a4[i] += bA4[i];
                        if constexpr (EPI == EPI_RESID) for (int i = 0; i < 4; ++i) a4[i] += (float)qA[j][i];
                        store4<bf16_t>(crow + nA, a4);
                    }
                    if (nB < g.N) {
                        for (int i = 0; i < 4; ++i) b4[i] += bB4[i];
                        if constexpr (EPI == EPI_RESID) for (int i = 0; i < 4; ++i) b4[i] += (float)qB[j][i];
                        store4<bf16_t>(crow + nB, b4);
                    }
                }
            }
        }
    }
}

struct AsmTile { int bm, bn; const char* name; };
const AsmTile kAsmTiles[] = {{256, 256, "asm256x256"}, {320, 256, "asm320x256"}, {160, 256, "asm160x256"}};

template <int BM, int BN, int WGM, int WGN, int EPI, bool MF16 = false>
int launch_asm(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static std::atomic<unsigned long long> attr_devs{0}, attr_devs_rsq{0};      // one mask per kernel (the attribute is per kernel and device)
    void (*kern)(const GemmArgs);
#ifdef LTX_EXPERIMENTS
    if constexpr (MF16) kern = gemm_asm16_kernel<BM, BN, WGM, WGN, EPI>; else kern = gemm_asm_kernel<BM, BN, WGM, WGN, EPI>;
#else
    static_assert(MF16, "the 32x32x16 kernel is an experiment build's");
    kern = gemm_asm16_kernel<BM, BN, WGM, WGN, EPI>;
#endif
    bool wrote_rowsq = false;
    // the residual epilogues of the 160 x 256 tile request their residual rows inside the K loop where it is long enough
    constexpr bool CAN_PF = MF16 && BM == 160 && BN == 256 && (EPI == EPI_GATE_RESID || EPI == EPI_RESID);
    bool pf_r = false;
    if constexpr (CAN_PF) pf_r = g.K / 64 >= 12 && ltx_exp("gemm_asm16_resid_prefetch", 1);
    if constexpr (MF16 && (EPI == EPI_BIAS || EPI == EPI_GATE_RESID || EPI == EPI_RESID)) {   // GemmArgs::rowsq as a by-product of the wide epilogue
        if (g.rowsq) { kern = gemm_asm16_kernel<BM, BN, WGM, WGN, EPI, true>; wrote_rowsq = true; }
    }
    static std::atomic<unsigned long long> attr_devs_pf{0}, attr_devs_pf_rsq{0};
    if constexpr (CAN_PF) {
        if (pf_r) kern = g.rowsq ? gemm_asm16_kernel<BM, BN, WGM, WGN, EPI, true, true> : gemm_asm16_kernel<BM, BN, WGM, WGN, EPI, false, true>;
    }
    // norm fold (kernels.h): the consumer side is its own instantiation (+ 2 KiB of LDS for the rows' 1 / rms); the producer side
    // lives in the RSQ instantiations of the residual epilogues
    static std::atomic<unsigned long long> attr_devs_fold{0};
    bool fold_in = false;
    if constexpr (MF16 && (EPI == EPI_BIAS || EPI == EPI_GELU)) {
        if (g.rs_sq) {
            if (g.rowsq || !g.cvec || g.rs_n < 4 || g.rs_n > 16 || g.rs_n % 4 || g.rows_per_batch < 1) LTX_FAIL(LTX_ERR_ARG, "gemm_asm16: norm fold needs cvec, 4 .. 16 row partials in groups of four and no rowsq by-product");
            kern = gemm_asm16_kernel<BM, BN, WGM, WGN, EPI, false, false, true>; fold_in = true;
        }
    } else if (g.rs_sq) LTX_FAIL(LTX_ERR_ARG, "gemm_asm16: norm fold (rs_sq) on an epilogue that does not carry it");
    if (g.C2 && !(wrote_rowsq && (EPI == EPI_GATE_RESID || EPI == EPI_RESID))) LTX_FAIL(LTX_ERR_ARG, "gemm_asm16: the second output (C2) rides on the residual epilogues with rowsq");
    const int smem_l = smem + (fold_in ? 2048 : 0);
    LTX_TRY(ltx_set_max_dyn_smem(fold_in ? attr_devs_fold : pf_r ? (wrote_rowsq ? attr_devs_pf_rsq : attr_devs_pf) : (wrote_rowsq ? attr_devs_rsq : attr_devs), reinterpret_cast<const void*>(kern), smem_l));
    GemmArgs ga = g;
    ga.xcd_remap = ltx_exp("xcd_remap", 1);
    {   // near-square patch of tiles per XCD (gemm_big.hip launch_one): one block per CU
        const int C = 32;
        int gm = 1; while ((gm + 1) * (gm + 1) * BM <= C * BN) ++gm;
        { const int x_gm = ltx_exp("gemm_group_m", -1); if (x_gm >= 0) gm = x_gm; }
        const int ntm = cdiv(g.M, BM);
        if (gm > ntm) gm = ntm;
        ga.group_m = gm < 2 ? 0 : gm;
    }
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    LTX_LAUNCH_TIMED(kern, dim3((unsigned)tiles), dim3(256), smem_l, s, ga);
    LTX_CHECK_LAUNCH();
    if (wrote_rowsq) ltx_gemm_rowsq_done();
    return LTX_OK;
}

template <int BM, int BN, int WGM, int WGN, bool MF16 = false>
int launch_asm_epi(const GemmArgs& g, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_asm<BM, BN, WGM, WGN, EPI_BIAS, MF16>(g, s);
        case EPI_GELU: return launch_asm<BM, BN, WGM, WGN, EPI_GELU, MF16>(g, s);
        case EPI_GATE_RESID: return launch_asm<BM, BN, WGM, WGN, EPI_GATE_RESID, MF16>(g, s);
        case EPI_RESID: return launch_asm<BM, BN, WGM, WGN, EPI_RESID, MF16>(g, s);
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_asm: bad epilogue");
}

}  // namespace

// Shape-only eligibility (see the header comment): bf16 linear layers large enough to fill the chip with 256-wide tiles.
// What the 16x16x32 kernel needs of a call (shape, strides, pointers, epilogue) - no environment: the plan family
// "asm16:*" of gemm_big.hip measures it beside the gemm_big tiles wherever this holds.
bool ltx_gemm_asm16_fits(const GemmArgs& g, int epi) {
    if (g.conv || g.pn_on) return false;
    if (epi != EPI_BIAS && epi != EPI_GELU && epi != EPI_GATE_RESID && epi != EPI_RESID) return false;
    if (g.K < 128 || g.K % 64 != 0 || g.lda % 8 != 0 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return false;
    if ((double)g.M * g.lda * 2.0 >= 2147483648.0 || (double)g.N * g.K * 2.0 >= 2147483648.0) return false;     // 32-bit buffer offsets
    // epilogue: 4-column groups inside or outside N as a whole, 8-byte aligned rows, a tile inside one column segment
    const bool seg_ok = !g.c_seg_shift || ((1 << g.c_seg_shift) % 256 == 0 && g.c_seg_stride % 4 == 0);
    if ((double)g.ldc * 2.0 * 320.0 >= 2147483648.0 || (double)g.ldr * 2.0 * 320.0 >= 2147483648.0) return false;     // 32-bit offsets inside a tile
    if (g.N % 8 != 0 || g.ldc % 4 != 0 || ((uintptr_t)g.C & 7) || !seg_ok) return false;
    if (g.bias && ((uintptr_t)g.bias & 7)) return false;
    if ((epi == EPI_GATE_RESID || epi == EPI_RESID) && (!g.resid || g.ldr % 4 != 0 || ((uintptr_t)g.resid & 7))) return false;
    if (epi == EPI_GATE_RESID && (!g.gate || ((uintptr_t)g.gate & 15) || g.gate_stride % 4 != 0 || g.rows_per_batch < 320)) return false;      // (a tile straddles at most one batch boundary)
    if (ltx_gemm_split_factor(g) > 1) return false;       // small outputs keep the split-K tiles of gemm_big
    return g.M > 512 && g.N >= 512;                       // (up to 512 rows: gemm_ring.hip's tiles, never split)
}

// Norm fold (GemmArgs::C2 / ::rs_sq): what the wide epilogue needs on top of ltx_gemm_asm16_fits
bool ltx_gemm_fold_ok(const GemmArgs& g, int epi) {
    if (!ltx_gemm_asm16_fits(g, epi) || (ltx_opt().gemm_off & LTX_FAM_ASM16) || !ltx_opt().gemm_wide_epi) return false;
    if (g.C2) {
        if ((epi != EPI_GATE_RESID && epi != EPI_RESID) || !g.rowsq || !g.scale2 || g.scale2_stride % 4 || ((uintptr_t)g.scale2 & 15) || ((uintptr_t)g.C2 & 7) || g.rows_per_batch < 320 || g.c_seg_shift) return false;
    }
    if (g.rs_sq) {
        if ((epi != EPI_BIAS && epi != EPI_GELU) || g.rowsq || !g.cvec || g.cvec_stride % 4 || ((uintptr_t)g.cvec & 15) || g.rs_n < 4 || g.rs_n > 16 || g.rs_n % 4 || ((uintptr_t)g.rs_sq & 15) || g.rows_per_batch < 320) return false;      // (a tile straddles at most one batch boundary)
    }
    return true;
}

// Experiment builds, x_gemm_asm=1: round 1's 32x32x16 loop for every shape it serves (tests, A/B).  The 16x16x32 kernel is a plan
// family of gemm_big.hip ("asm16:*"; forced with the gemm_plan option).
bool ltx_gemm_asm_eligible(const GemmArgs& g, int dtype, int epi) {
#ifdef LTX_EXPERIMENTS
    if (dtype != LTX_DT_BF16 || g.conv || ltx_exp("gemm_asm", 0) != 1) return false;
    if (epi != EPI_BIAS && epi != EPI_GELU && epi != EPI_GATE_RESID && epi != EPI_RESID) return false;
    if (g.K < 128 || g.K % 64 != 0 || g.lda % 8 != 0 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return false;
    if ((double)g.M * g.lda * 2.0 >= 2147483648.0 || (double)g.N * g.K * 2.0 >= 2147483648.0) return false;     // 32-bit buffer offsets
    if (ltx_gemm_split_factor(g) > 1) return false;       // small outputs keep the split-K tiles of gemm_big
    return g.M >= 2048 && g.N >= 1024;
#else
    (void)g; (void)dtype; (void)epi;
    return false;
#endif
}

int ltx_gemm_asm_pick_tile(int M, int N) {
    { const int f = ltx_exp("gemm_asm_tile", -1); if (f >= 0 && f < 3) return f; }      // experiment builds: 0 / 1 / 2 = kAsmTiles order
    double best = 1e30; int bi = 0;
    for (int i = 0; i < 3; ++i) {
        const int64_t tiles = (int64_t)cdiv(M, kAsmTiles[i].bm) * cdiv(N, kAsmTiles[i].bn);
        const double cost = (double)cdiv64(tiles, 256) * (double)(kAsmTiles[i].bm * kAsmTiles[i].bn);
        if (cost < best * 0.999) { best = cost; bi = i; }
    }
    return bi;
}
const char* ltx_gemm_asm_tile_name(int i) { return i >= 0 && i < 3 ? kAsmTiles[i].name : ""; }

#ifdef GEMM_ASM_TRACE
extern "C" int ltx_dbg_gemm_asm16_trace(uint32_t* out, int n_words) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_asm16_trace), (size_t)n_words * 4) == hipSuccess ? 0 : -1;
}
#endif

int ltx_launch_gemm_asm16(const GemmArgs& g, int epi, int tile, hipStream_t s) {     // tile 0: 256 x 256, 1: 160 x 256, 2: 320 x 256
    ltx_prof_kernel(LTX_PROFK_GEMM_ASM16);
    if (tile == 0) return launch_asm_epi<256, 256, 2, 2, true>(g, epi, s);
    return tile == 1 ? launch_asm_epi<160, 256, 1, 4, true>(g, epi, s) : launch_asm_epi<320, 256, 2, 2, true>(g, epi, s);
}

// What the conv-mode kernel needs of a call: bf16 3x3x3 conv with whole 64-channel slices, a plain / residual / depth-to-space
// epilogue (no fused norm, no segmented output), more rows than the small-plane tiles serve, operands the 32-bit offsets reach.
bool ltx_gemm_asm16_conv_fits(const GemmArgs& g, int epi) {
    if (!g.conv || g.ntaps != 27 || g.kh != 3 || g.kw != 3 || g.pn_on || g.c_seg_shift || g.rowsq) return false;
    if (epi != EPI_BIAS && epi != EPI_RESID && epi != EPI_D2S) return false;
    if (g.Cin % 64 != 0 || g.K != g.Cin || g.N % 8 != 0 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15) || ((uintptr_t)g.C & 7)) return false;
    if (g.bias && ((uintptr_t)g.bias & 7)) return false;
    if (epi == EPI_RESID && (!g.resid || g.ldr % 4 != 0 || ((uintptr_t)g.resid & 7) || g.ldc % 4 != 0)) return false;
    if (epi == EPI_BIAS && g.ldc % 4 != 0) return false;
    if (epi == EPI_D2S && (g.Cf % 4 != 0 || g.N != 8 * g.Cf)) return false;
    if ((double)g.N * g.K * 2.0 * 10.0 >= 4294967296.0) return false;          // 9 N K 2 and its differences stay inside 32 bits
    // from 1024 output channels up: below that the halo-staged kernel (one staging of the activation per nine taps, tiles as wide as
    // the output) is the better structure, and a plan is measured with a bias epilogue only - on one box the 256-channel stage's
    // residual convs were handed to this kernel at 955 us against 755 (profiles/r5z_bench_c2_kernel_stats.md, launches 24-32)
    return g.M > 512 && g.N >= 1024 && ltx_gemm_big_fits(g);
}

int ltx_launch_gemm_asm16_conv(const GemmArgs& g, int epi, hipStream_t s) {
    if (!ltx_gemm_asm16_conv_fits(g, epi)) LTX_FAIL(LTX_ERR_ARG, "gemm_asm16 (conv): shape not eligible");
    ltx_prof_kernel(LTX_PROFK_GEMM_ASM16);
    constexpr int smem = 2 * (256 + 256) * 128;
    static std::atomic<unsigned long long> devs[3] = {{0}, {0}, {0}};
    void (*kern)(const GemmArgs) = epi == EPI_BIAS ? gemm_asm16_conv_kernel<EPI_BIAS> : (epi == EPI_RESID ? gemm_asm16_conv_kernel<EPI_RESID> : gemm_asm16_conv_kernel<EPI_D2S>);
    LTX_TRY(ltx_set_max_dyn_smem(devs[epi == EPI_BIAS ? 0 : (epi == EPI_RESID ? 1 : 2)], reinterpret_cast<const void*>(kern), smem));
    GemmArgs ga = g;
    const int tiles = cdiv(g.M, 256) * cdiv(g.N, 256);
    LTX_TRY(ltx_gemm_split_workspace(&ga, tiles, 256, 256, s));       // the shape rule's K ranges + slabs / tickets (gemm_big.hip)
    const int sf = ga.sk_sf > 1 ? ga.sk_sf : 1;
    LTX_LAUNCH_TIMED(kern, dim3((unsigned)(tiles * sf)), dim3(256), smem, s, ga);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

int ltx_launch_gemm_asm(const GemmArgs& g, int epi, hipStream_t s) {
#ifdef LTX_EXPERIMENTS
    ltx_prof_kernel(LTX_PROFK_GEMM_ASM);
    switch (ltx_gemm_asm_pick_tile(g.M, g.N)) {
        case 0: return launch_asm_epi<256, 256, 2, 2>(g, epi, s);
        case 1: return launch_asm_epi<320, 256, 2, 2>(g, epi, s);
        default: return launch_asm_epi<160, 256, 1, 4>(g, epi, s);
    }
#else
    (void)g; (void)epi; (void)s;
    LTX_FAIL(LTX_ERR_UNSUPPORTED, "gemm_asm (32x32x16) is compiled into experiment builds only");
#endif
}
