// Large-tile bf16 MFMA GEMM / implicit-GEMM conv3d for gfx950 (the production path of the DiT
// Linears and the VAE convs; gemm.hip's 128x128 register-staged kernel covers f32 and small shapes).
//
//   * tile BM x BN (256/192/128 x 256/128), K-step 64 bf16 (128-B LDS rows), 512 threads =
//     8 waves as 2(M) x 4(N), each wave (BM/2) x (BN/4) with v_mfma_f32_16x16x32_bf16;
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip): one wave
//     instruction fills 8 rows x 128 B, LDS destination is lane-linear, so the bank swizzle
//     (16-B chunk ^ ((row>>1)&7)) is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 fragment reads (same involution on both sides);
//   * two LDS stages (<= 128 KiB): the loads of K-step t+1 are in flight while step t is
//     multiplied; one vmcnt(0)+barrier per K-step;
//   * conv mode gathers the A rows per tap from the channels-last activation: replicate padding
//     on T is a clamp, zero padding on H/W (and K tails) redirect the lane to a zero page;
//   * D = Wfrag x Afrag so a lane owns 4 consecutive output columns (shared fused epilogues).
#include "gemm_common.h"

namespace {

constexpr int BIG_THREADS = 512;
constexpr int ROWB = 128;          // bytes per LDS row (64 bf16)

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];   // all zeros (static init)

__device__ __forceinline__ int swz_big(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];

template <int BM, int BN, int EPI, bool CONV>
__global__ __launch_bounds__(BIG_THREADS) void gemm_big_kernel(const GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 4, FM = WM / 16, FN = WN / 16;
    constexpr int AI = BM / 64, BI = BN / 64;          // glds instructions per wave per K-step (A, B)
    constexpr int STAGE = (BM + BN) * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = (g.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
    const int Kdim = g.K;
    const int ktiles = (Kdim + 63) / 64;
    const int ntaps = CONV ? g.ntaps : 1;
    const int nk = ktiles * ntaps;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);

    // per-lane staging geometry: instruction j of this wave fills LDS rows 8*(j*8+wave) .. +7;
    // lane -> (row in group = lane>>3, physical chunk = lane&7), logical chunk = pc ^ ((row>>1)&7)
    const int lr = lane >> 3, pc = lane & 7;
    int a_chunk[AI], b_chunk[BI];
    int64_t a_off[AI];                 // linear mode: element offset of the row
    int cb[AI], ct[AI], chh[AI], cw[AI];
    int64_t b_off[BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 8 * (j * 8 + wave) + lr;
        a_chunk[j] = pc ^ ((row >> 1) & 7);
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
        if constexpr (CONV) {
            int w = m % g.Wd; int t1 = m / g.Wd;
            int h = t1 % g.H; int t2 = t1 / g.H;
            ct[j] = t2 % g.T; cb[j] = t2 / g.T; chh[j] = h; cw[j] = w;
        } else {
            a_off[j] = (int64_t)m * g.lda;
        }
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 8 * (j * 8 + wave) + lr;
        b_chunk[j] = pc ^ ((row >> 1) & 7);
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
        b_off[j] = (int64_t)n * Kdim;
    }

    auto stage = [&](int kt, int buf) {
        unsigned char* As = big_smem + buf * STAGE;
        unsigned char* Bs = As + BM * ROWB;
        int tap = 0, kk = kt;
        if constexpr (CONV) { tap = kt / ktiles; kk = kt - tap * ktiles; }
        int dt = 0, dh = 0, dw = 0;
        if constexpr (CONV) {
            const int khw = g.kh * g.kw;
            const int it = tap / khw; const int rem = tap - it * khw;
            const int ih = rem / g.kw; const int iw = rem - ih * g.kw;
            dt = it - g.pad_t; dh = ih - g.kh / 2; dw = iw - g.kw / 2;
        }
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            const int k = kk * 64 + a_chunk[j] * 8;
            const unsigned char* src = zero;
            if constexpr (CONV) {
                int tt = ct[j] + dt; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);      // replicate pad (vae.rs:374-413)
                const int hh = chh[j] + dh, ww = cw[j] + dw;
                if (k < Kdim && hh >= 0 && hh < g.H && ww >= 0 && ww < g.Wd)                 // zero pad (vae.rs:337-349)
                    src = reinterpret_cast<const unsigned char*>(A + ((((int64_t)cb[j] * g.T + tt) * g.H + hh) * g.Wd + ww) * (int64_t)g.Cin + k);
            } else {
                if (k < Kdim) src = reinterpret_cast<const unsigned char*>(A + a_off[j] + k);
            }
            glds16(src, As + (j * 8 + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int k = kk * 64 + b_chunk[j] * 8;
            const unsigned char* src = zero;
            if (k < Kdim) src = reinterpret_cast<const unsigned char*>(W + (int64_t)tap * g.N * Kdim + b_off[j] + k);
            glds16(src, Bs + (j * 8 + wave) * 1024);
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __syncthreads();                                   // emits vmcnt(0) for the outstanding LDS-DMA

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const unsigned char* As = big_smem + buf * STAGE;
        const unsigned char* Bs = As + BM * ROWB;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Chunk16 af[FM], wf[FN];
#pragma unroll
            for (int f = 0; f < FM; ++f) af[f].u = *reinterpret_cast<const u32x4*>(As + swz_big(wm * WM + f * 16 + frow, kb * 4 + fq));
#pragma unroll
            for (int f = 0; f < FN; ++f) wf[f].u = *reinterpret_cast<const u32x4*>(Bs + swz_big(wn * WN + f * 16 + frow, kb * 4 + fq));
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = Mma<bf16_t>::run(wf[fn], af[fm], acc[fm][fn]);
        }
        __syncthreads();
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
            epilogue<bf16_t, EPI>(g, m, nb, v);
        }
    }
}

template <int BM, int BN, int EPI, bool CONV>
int launch_one(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * ROWB;
    static bool attr_set = false;
    auto kern = gemm_big_kernel<BM, BN, EPI, CONV>;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid((unsigned)(cdiv(g.M, BM) * cdiv(g.N, BN))), block(BIG_THREADS);
    hipLaunchKernelGGL(kern, grid, block, smem, s, g);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

template <int BM, int BN, bool CONV>
int launch_epi(const GemmArgs& g, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_one<BM, BN, EPI_BIAS, CONV>(g, s);
        case EPI_GELU: return launch_one<BM, BN, EPI_GELU, CONV>(g, s);
        case EPI_GATE_RESID: return launch_one<BM, BN, EPI_GATE_RESID, CONV>(g, s);
        case EPI_RESID: return launch_one<BM, BN, EPI_RESID, CONV>(g, s);
        case EPI_D2S: if constexpr (CONV) return launch_one<BM, BN, EPI_D2S, CONV>(g, s); break;
        case EPI_UNPATCH: if constexpr (CONV) return launch_one<BM, BN, EPI_UNPATCH, CONV>(g, s); break;
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_big: bad epilogue");
}

template <bool CONV>
int launch_tile(const GemmArgs& g, int epi, int bm, int bn, hipStream_t s) {
    if (bm == 256 && bn == 256) return launch_epi<256, 256, CONV>(g, epi, s);
    if (bm == 192 && bn == 256) return launch_epi<192, 256, CONV>(g, epi, s);
    if (bm == 128 && bn == 256) return launch_epi<128, 256, CONV>(g, epi, s);
    if (bm == 256 && bn == 128) return launch_epi<256, 128, CONV>(g, epi, s);
    if (bm == 192 && bn == 128) return launch_epi<192, 128, CONV>(g, epi, s);
    if (bm == 128 && bn == 128) return launch_epi<128, 128, CONV>(g, epi, s);
    LTX_FAIL(LTX_ERR_ARG, "gemm_big: unsupported tile");
}

}  // namespace

// Tile choice: minimise (rounds over 256 CUs) x (tile area); ties go to the larger tile (less operand re-reading).
void ltx_gemm_big_pick_tile(int M, int N, int* bm_out, int* bn_out) {
    static const int cand[6][2] = {{256, 256}, {192, 256}, {128, 256}, {256, 128}, {192, 128}, {128, 128}};
    const char* force = getenv("LTX_GEMM_TILE");       // e.g. "256x128" (tuning aid)
    if (force) { int a = 0, b = 0; if (sscanf(force, "%dx%d", &a, &b) == 2) { *bm_out = a; *bn_out = b; return; } }
    // per-CU sustained rate of each tile at full occupancy (TFLOP/s, measured with tools/microbench.py tiles on
    // MI355X, random data) and how many 512-thread blocks of it fit one CU (LDS: 2*(BM+BN)*128 B of 160 KiB)
    static const double rate[6] = {1160, 1190, 1023, 989, 1400, 1032};
    static const int per_cu[6] = {1, 1, 1, 1, 2, 2};
    double best = 1e30; int bi = 0;
    for (int i = 0; i < 6; ++i) {
        const int bm = cand[i][0], bn = cand[i][1];
        if (bn > 128 && N <= 128) continue;
        const int64_t tiles = (int64_t)cdiv(M, bm) * cdiv(N, bn);
        const double cost = (double)cdiv64(tiles, 256 * per_cu[i]) * (double)(bm * bn * per_cu[i]) / rate[i];
        if (cost < best * 0.999) { best = cost; bi = i; }
    }
    *bm_out = cand[bi][0]; *bn_out = cand[bi][1];
}

bool ltx_gemm_big_eligible(const GemmArgs& g, int dtype) {
    if (dtype != LTX_DT_BF16) return false;
    const char* off = getenv("LTX_GEMM_BIG");
    if (off && off[0] == '0') return false;
    return g.M >= 1024 && g.N >= 64;
}

int ltx_launch_gemm_big(const GemmArgs& g, int epi, hipStream_t s) {
    int bm, bn;
    ltx_gemm_big_pick_tile(g.M, g.N, &bm, &bn);
    return g.conv ? launch_tile<true>(g, epi, bm, bn, s) : launch_tile<false>(g, epi, bm, bn, s);
}
