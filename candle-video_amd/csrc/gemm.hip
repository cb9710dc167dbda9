// MFMA GEMM / implicit-GEMM conv3d for gfx950.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        (Linear: ltx_transformer.rs nn::Linear call sites)
//   conv  : A rows are gathered per kernel tap from a channels-last activation
//           [B,T,H,W,Cin]; W is packed [tap][N][Cin]  (LtxVideoCausalConv3d, vae.rs:415-464:
//           replicate padding on T, zero padding on H/W, bias once after the sum)
//
// Tile 128x128, K-step 128 BYTES (64 bf16 / 32 f32), 256 threads = 2x2 waves of 64x64,
// double-buffered LDS (64 KiB), register-staged prefetch (global loads for tile t+1 are
// issued before the MFMAs of tile t; LDS write after them; one barrier per K-step).
// LDS rows are 128 B with the 16-B chunk index XOR-swizzled by ((row>>1)&7) so that every
// ds_read_b128 fragment read (16 rows x 4 chunks) lands on 16 distinct 16-B bank slots.
// The MFMA is issued as D = Wfrag x Afrag so each lane ends up with 4 CONSECUTIVE output
// columns n of one row m (8-B / 16-B stores, vector bias/gate loads).
// T = bf16 uses v_mfma_f32_16x16x32_bf16; T = float uses v_mfma_f32_16x16x4_f32 (exact f32).
#include "gemm_common.h"
#include "options.h"

namespace {

constexpr int BM = 128, BN = 128, BKB = 128;   // BKB: K-step in bytes
constexpr int NTHREADS = 256;

__device__ __forceinline__ int swz(int row, int chunk) { return (row * 128) + (((chunk ^ ((row >> 1) & 7))) << 4); }

template <typename T, int EPI, bool CONV>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmArgs g) {
    constexpr int CH = ElemTraits<T>::CHUNK;       // elements per 16-B chunk
    constexpr int BK = BKB / (int)sizeof(T);       // elements per K-step
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BKB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
    const T* __restrict__ W = reinterpret_cast<const T*>(g.W);

    const int lrow = tid >> 3;        // 0..31
    const int lchunk = tid & 7;       // 0..7
    const int Kdim = g.K;             // per-tap K
    const int ktiles = (Kdim + BK - 1) / BK;
    const int ntaps = CONV ? g.ntaps : 1;
    const int nk = ktiles * ntaps;

    // per-thread row bookkeeping (4 rows each for A and W)
    int64_t a_off[4];     // GEMM: row offset in elements.  CONV: unused
    int cb[4], ct[4], chh[4], cw[4]; bool mval[4];
    int64_t w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = lrow + 32 * i;
        int m = m0 + r; if (m > g.M - 1) m = g.M - 1;
        if constexpr (CONV) {
            int w = m % g.Wd; int t1 = m / g.Wd;
            int h = t1 % g.H; int t2 = t1 / g.H;
            ct[i] = t2 % g.T; cb[i] = t2 / g.T; chh[i] = h; cw[i] = w;
        } else {
            a_off[i] = (int64_t)m * g.lda;
        }
        mval[i] = true;
        int n = n0 + r; if (n > g.N - 1) n = g.N - 1;
        w_off[i] = (int64_t)n * Kdim;
    }

    Chunk16 ra[4], rw[4];
    auto gload = [&](int kt) {
        int tap = 0, kk = kt;
        if constexpr (CONV) { tap = kt / ktiles; kk = kt - tap * ktiles; }
        const int k = kk * BK + lchunk * CH;
        const bool kval = k < Kdim;
        int dt = 0, dh = 0, dw = 0;
        if constexpr (CONV) {
            int khw = g.kh * g.kw;
            int it = tap / khw; int rem = tap - it * khw;
            int ih = rem / g.kw; int iw = rem - ih * g.kw;
            dt = it - g.pad_t; dh = ih - g.kh / 2; dw = iw - g.kw / 2;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Chunk16 z; z.u = (u32x4){0u, 0u, 0u, 0u};
            if constexpr (CONV) {
                int tt = ct[i] + dt; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);   // replicate pad (vae.rs:374-413)
                int hh = chh[i] + dh, ww = cw[i] + dw;
                bool v = kval && hh >= 0 && hh < g.H && ww >= 0 && ww < g.Wd;           // zero pad (vae.rs:337-349)
                if (v) {
                    const T* p = A + ((((int64_t)cb[i] * g.T + tt) * g.H + hh) * g.Wd + ww) * (int64_t)g.Cin + k;
                    z.u = *reinterpret_cast<const u32x4*>(p);
                }
            } else {
                if (kval) z.u = *reinterpret_cast<const u32x4*>(A + a_off[i] + k);
            }
            ra[i] = z;
            Chunk16 y; y.u = (u32x4){0u, 0u, 0u, 0u};
            if (kval) y.u = *reinterpret_cast<const u32x4*>(W + (int64_t)tap * g.N * Kdim + w_off[i] + k);
            rw[i] = y;
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* As = smem + buf * ((BM + BN) * BKB);
        unsigned char* Bs = As + BM * BKB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = lrow + 32 * i;
            *reinterpret_cast<u32x4*>(As + swz(r, lchunk)) = ra[i].u;
            *reinterpret_cast<u32x4*>(Bs + swz(r, lchunk)) = rw[i].u;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gload(0);
    swrite(0);
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* As = smem + buf * ((BM + BN) * BKB);
        const unsigned char* Bs = As + BM * BKB;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Chunk16 af[4], wf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                int ar = wm * 64 + f * 16 + frow;
                af[f].u = *reinterpret_cast<const u32x4*>(As + swz(ar, kb * 4 + fq));
                int br = wn * 64 + f * 16 + frow;
                wf[f].u = *reinterpret_cast<const u32x4*>(Bs + swz(br, kb * 4 + fq));
            }
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = Mma<T>::run(wf[fn], af[fm], acc[fm][fn]);
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds, for fragment (fm,fn): row m = .. + (lane&15), cols nb .. nb+3 with nb = .. + 4*(lane>>4)
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        int m = m0 + wm * 64 + fm * 16 + frow;
        if (m >= g.M) continue;
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            int nb = n0 + wn * 64 + fn * 16 + 4 * fq;
            if (nb >= g.N) continue;
            float v[4] = {acc[fm][fn][0], acc[fm][fn][1], acc[fm][fn][2], acc[fm][fn][3]};
            epilogue<T, EPI>(g, m, nb, v);
        }
    }
}

template <typename T, bool CONV>
int launch_t(const GemmArgs& g, int epi, hipStream_t s) {
    dim3 grid((unsigned)(cdiv(g.M, BM) * cdiv(g.N, BN))), block(NTHREADS);
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_kernel<T, EPI_BIAS, CONV>), grid, block, 0, s, g); break;
        case EPI_GELU: hipLaunchKernelGGL((gemm_kernel<T, EPI_GELU, CONV>), grid, block, 0, s, g); break;
        case EPI_GATE_RESID: hipLaunchKernelGGL((gemm_kernel<T, EPI_GATE_RESID, CONV>), grid, block, 0, s, g); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_kernel<T, EPI_RESID, CONV>), grid, block, 0, s, g); break;
        case EPI_D2S: hipLaunchKernelGGL((gemm_kernel<T, EPI_D2S, CONV>), grid, block, 0, s, g); break;
        case EPI_UNPATCH: hipLaunchKernelGGL((gemm_kernel<T, EPI_UNPATCH, CONV>), grid, block, 0, s, g); break;
        default: LTX_FAIL(LTX_ERR_ARG, "gemm: bad epilogue");
    }
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

}  // namespace

static thread_local bool t_rowsq_done = false;
void ltx_gemm_rowsq_done() { t_rowsq_done = true; }

int ltx_launch_gemm(const GemmArgs& g, int dtype, int epi, hipStream_t s) {
    const int ch = dtype == LTX_DT_BF16 ? 8 : 4;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) LTX_FAIL(LTX_ERR_ARG, "gemm: empty problem");
    if (g.K % ch != 0) LTX_FAIL(LTX_ERR_ARG, "gemm: K must be a multiple of the 16-byte chunk");
    if (g.N % 4 != 0) LTX_FAIL(LTX_ERR_ARG, "gemm: N must be a multiple of 4");
    if (!g.conv && g.lda % ch != 0) LTX_FAIL(LTX_ERR_ARG, "gemm: lda must be 16-byte aligned");
    if ((epi == EPI_D2S || epi == EPI_UNPATCH) && !g.conv) LTX_FAIL(LTX_ERR_ARG, "gemm: d2s/unpatch need conv mode");
    if (g.rowsq && (g.conv || g.c_seg_shift || epi == EPI_D2S || epi == EPI_UNPATCH)) LTX_FAIL(LTX_ERR_ARG, "gemm: rowsq needs a dense linear output");
    // K ranges left to the consumer are gemm_ring's (through gemm_big's plan dispatch); no other kernel honours the field
    if (g.defer_parts && (ltx_gemm_asm_eligible(g, dtype, epi) || !ltx_gemm_big_eligible(g, dtype)))
        LTX_FAIL(LTX_ERR_ARG, "gemm: defer_parts set on a call that is not routed to the ring tiles (ask ltx_gemm_defer_ok first)");
    t_rowsq_done = false;
    void* tok = nullptr;
    ltx_prof_begin(g.conv ? LTX_PROF_CONV : LTX_PROF_GEMM, 2.0 * g.M * (double)g.N * g.K * (g.conv ? g.ntaps : 1), s, &tok);
    int rc;
    if (ltx_opt().gemm_trace && dtype == LTX_DT_BF16 && !ltx_gemm_asm_eligible(g, dtype, epi) && !ltx_gemm_big_eligible(g, dtype))     // debugging aid: bf16 shapes left to the 128 x 128 kernel
        fprintf(stderr, "[ltx] gemm128 serves M=%d N=%d K=%d conv=%d ntaps=%d B=%d T=%d H=%d W=%d epi=%d fits=%d\n", g.M, g.N, g.K, g.conv, g.ntaps, g.B, g.T, g.H, g.Wd, epi, (int)ltx_gemm_big_fits(g));
    if (ltx_gemm_asm_eligible(g, dtype, epi)) rc = ltx_launch_gemm_asm(g, epi, s);
    else if (ltx_gemm_big_eligible(g, dtype)) rc = ltx_launch_gemm_big(g, epi, s);
    else if (dtype == LTX_DT_BF16) rc = g.conv ? launch_t<bf16_t, true>(g, epi, s) : launch_t<bf16_t, false>(g, epi, s);
    else rc = g.conv ? launch_t<float, true>(g, epi, s) : launch_t<float, false>(g, epi, s);
    ltx_prof_end(tok, s);
    // by-product not written by the kernel that ran (every kernel but gemm_asm16): the stand-alone pass, same canonical order
    if (rc == LTX_OK && g.rowsq && !t_rowsq_done) rc = ltx_launch_rowsq(g.C, dtype, g.M, g.N, g.ldc, g.rowsq, s);
    return rc;
}
