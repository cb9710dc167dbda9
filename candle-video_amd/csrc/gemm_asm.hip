// One-wave-per-SIMD bf16 GEMM for the DiT's linear layers (nn::Linear + the fused epilogues of gemm_common.h):
//   C[M, N] = epi(A[M, K] @ W[N, K]^T + bias)
// The K loop is ONE generated inline-asm statement per tile shape (tools/gen_gemm_asm.py -> gemm_asm_loop.inc): four waves
// of 128 x 128 (or 160 x 128 / 160 x 64) output each, accumulators in the AGPR half of the 512-register file,
// v_mfma_f32_32x32x16_bf16, and the K-step's 16 LDS-DMA pieces + 32 fragment reads placed one per MFMA gap - the
// interleave that hipcc cannot be made to emit (round 1's compiler-scheduled form of this layout lost 8-25 %).
// LDS image, source-side swizzle, buffer addressing with out-of-range = zeros and the epilogue arithmetic are those of
// gemm_big.hip.  The MFMA shape differs (32x32x16 here, 16x16x32 there) yet the results are BIT-IDENTICAL to gemm_big's: the
// matrix core accumulates its bf16 products in ascending k as an f32 chain, so only the k order matters
// (tests/test_gpu_gemm_asm.py).  Kept as a measured experiment behind LTX_GEMM_ASM=1: operand delivery (13-18 TB/s of L2->LDS
// over the chip) and MFMA time are both ~1 us per 64-deep K-step of a 256 x 256 tile, and a single in-order wave per SIMD
// overlaps them worse than gemm_big's two (tools/gemm_asm_tune.py ablations).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "gemm_common.h"

namespace {

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#ifndef GEMM_ASM_LOOP_INC
#define GEMM_ASM_LOOP_INC "gemm_asm_loop.inc"
#endif
#include GEMM_ASM_LOOP_INC

extern __shared__ __attribute__((aligned(16))) unsigned char asm_smem[];

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

struct AsmTile { int bm, bn; const char* name; };
const AsmTile kAsmTiles[] = {{256, 256, "asm256x256"}, {320, 256, "asm320x256"}, {160, 256, "asm160x256"}};

template <int BM, int BN, int WGM, int WGN, int EPI>
int launch_asm(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static std::atomic<unsigned long long> attr_devs{0};
    auto kern = gemm_asm_kernel<BM, BN, WGM, WGN, EPI>;
    LTX_TRY(ltx_set_max_dyn_smem(attr_devs, reinterpret_cast<const void*>(kern), smem));
    GemmArgs ga = g;
    const char* xr = getenv("LTX_XCD_REMAP");
    ga.xcd_remap = xr ? (xr[0] == '1') : 1;
    {   // near-square patch of tiles per XCD (gemm_big.hip launch_one): one block per CU
        const int C = 32;
        int gm = 1; while ((gm + 1) * (gm + 1) * BM <= C * BN) ++gm;
        if (const char* e = getenv("LTX_GEMM_GROUP_M")) { const int env_gm = atoi(e); if (env_gm >= 0) gm = env_gm; }
        const int ntm = cdiv(g.M, BM);
        if (gm > ntm) gm = ntm;
        ga.group_m = gm < 2 ? 0 : gm;
    }
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), smem, s, ga);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

template <int BM, int BN, int WGM, int WGN>
int launch_asm_epi(const GemmArgs& g, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_asm<BM, BN, WGM, WGN, EPI_BIAS>(g, s);
        case EPI_GELU: return launch_asm<BM, BN, WGM, WGN, EPI_GELU>(g, s);
        case EPI_GATE_RESID: return launch_asm<BM, BN, WGM, WGN, EPI_GATE_RESID>(g, s);
        case EPI_RESID: return launch_asm<BM, BN, WGM, WGN, EPI_RESID>(g, s);
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_asm: bad epilogue");
}

}  // namespace

// Shape-only eligibility (see the header comment): bf16 linear layers large enough to fill the chip with 256-wide tiles.
bool ltx_gemm_asm_eligible(const GemmArgs& g, int dtype, int epi) {
    if (dtype != LTX_DT_BF16 || g.conv) return false;
    // EXPERIMENT, off unless LTX_GEMM_ASM=1: measured 4-12 % behind gemm_big on the DiT shapes ("one wave per SIMD")
    const char* e = getenv("LTX_GEMM_ASM");
    if (!e || e[0] != '1') return false;
    if (epi != EPI_BIAS && epi != EPI_GELU && epi != EPI_GATE_RESID && epi != EPI_RESID) return false;
    if (g.K < 128 || g.K % 64 != 0 || g.lda % 8 != 0 || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return false;
    if ((double)g.M * g.lda * 2.0 >= 2147483648.0 || (double)g.N * g.K * 2.0 >= 2147483648.0) return false;     // 32-bit buffer offsets
    if (ltx_gemm_split_factor(g) > 1) return false;       // small outputs keep the split-K tiles of gemm_big
    return g.M >= 2048 && g.N >= 1024;
}

int ltx_gemm_asm_pick_tile(int M, int N) {
    if (const char* f = getenv("LTX_GEMM_ASM_TILE")) for (int i = 0; i < 3; ++i) if (!strcmp(f, kAsmTiles[i].name)) return i;
    double best = 1e30; int bi = 0;
    for (int i = 0; i < 3; ++i) {
        const int64_t tiles = (int64_t)cdiv(M, kAsmTiles[i].bm) * cdiv(N, kAsmTiles[i].bn);
        const double cost = (double)cdiv64(tiles, 256) * (double)(kAsmTiles[i].bm * kAsmTiles[i].bn);
        if (cost < best * 0.999) { best = cost; bi = i; }
    }
    return bi;
}
const char* ltx_gemm_asm_tile_name(int i) { return i >= 0 && i < 3 ? kAsmTiles[i].name : ""; }

int ltx_launch_gemm_asm(const GemmArgs& g, int epi, hipStream_t s) {
    switch (ltx_gemm_asm_pick_tile(g.M, g.N)) {
        case 0: return launch_asm_epi<256, 256, 2, 2>(g, epi, s);
        case 1: return launch_asm_epi<320, 256, 2, 2>(g, epi, s);
        default: return launch_asm_epi<160, 256, 1, 4>(g, epi, s);
    }
}
