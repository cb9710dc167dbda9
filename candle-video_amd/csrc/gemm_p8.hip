// Phase-interleaved 256-row bf16 GEMM / implicit-GEMM conv3d for gfx950: one 512-thread block per CU, the two
// waves of every SIMD ping-pong between an MFMA phase and a memory phase, and LDS-DMA prefetch stays in flight
// across the barriers (counted vmcnt, raw s_barrier) -- the structure that gets past the ~900 TF/s ceiling of the
// "vmcnt(0) + __syncthreads per K-step" loop in gemm_big.hip (PMC there: MFMA busy 45-49 %, 48 % of wave cycles in waits).
//
// Geometry (BN = 256 | 128):
//   tile 256 x BN, K-tile 64 bf16; 8 waves as WGM x WGN (2x4 | 4x2); the tile is cut in half-tiles A0,A1 (128 rows)
//   and B0,B1 (BN/2 rows); a wave owns a (128/WGM) x (BN/2/WGN) piece of EVERY quadrant (A-half x B-half), so one
//   phase = one quadrant x K=64 = FMQ*FNQ*2 MFMAs (16 | 8) on operands of exactly one A half and one B half.
//   LDS: 2 buffers x (A0|A1|B0|B1) = 128 KiB | 96 KiB, 128-B rows, 16-B chunk c of row r stored at c ^ ((r>>1)&7).
// Schedule (global phase g = 4*t + p, K-tile t, p = 0..3; half-tiles are numbered s = 4*t + {0:B0, 1:A0, 2:B1, 3:A1}):
//   MFMA    p0: Q(A0,B0)  p1: Q(A0,B1)  p2: Q(A1,B1)  p3: Q(A1,B0)
//   reads   phase g reads half-tile g+1:  p0: A0 (8 x b128)  p1: B1 (4)  p2: A1 (8)  p3: B0 of K-tile t+1 (4)
//           -> two B register sets that swap roles every K-tile, hence the K loop is unrolled by two K-tiles
//   stage   phase g issues half-tile g+6 (6 half-tiles in the prologue)
//   every phase: ds_reads, LDS-DMA issue, s_waitcnt vmcnt(4 half-tiles), s_barrier, lgkmcnt(0), MFMAs, s_barrier.
//   RAW: the wait of phase g (+ one barrier) certifies half-tile g+2 complete in LDS (all waves' pieces); it is read
//        in phase g+1.   WAR: half-tile s = g+6 overwrites half-tile s-8 = g-2, last read in phase g-3 -- at least two
//        full phases earlier, safe also for the wave group that runs one barrier behind.
//   The groups (waves 0-3 / 4-7 = one wave per SIMD each) are offset by ONE barrier, so while one group issues
//   MFMAs (s_setprio 1) the other does its LDS reads / DMA issue.
#include <cstring>
#include <type_traits>
#include "gemm_common.h"

namespace {

extern __shared__ __attribute__((aligned(16))) unsigned char p8_smem[];

template <int BN, int WGM, int WGN, int EPI, bool CONV>
__global__ __launch_bounds__(512) void gemm_p8_kernel(const GemmArgs g) {
    constexpr int BM = 256, HB = BN / 2;
    constexpr int A_HALF = 128 * 128, B_HALF = HB * 128;          // bytes
    constexpr int BUF = 2 * A_HALF + 2 * B_HALF;
    constexpr int WMH = 128 / WGM, WNH = HB / WGN;                 // wave's rows of an A half / of a B half
    constexpr int FMQ = WMH / 16, FNQ = WNH / 16;
    constexpr int GA = 2, GB = HB / 64;                            // LDS-DMA instructions per wave per half-tile
    static_assert(WGM * WGN == 8 && WMH % 16 == 0 && WNH % 16 == 0 && GB >= 1, "layout");
    // every phase leaves 4 half-tiles in flight: the DMA count of A0,B0,B1,A1 in any rotation
    constexpr int VM_INFLIGHT = 2 * GA + 2 * GB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2;                                     // ping-pong group
    const int wm = wave / WGN, wn = wave % WGN;
    const int ntn = (g.N + BN - 1) / BN;
    int bid = blockIdx.x;
    if (g.xcd_remap) {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
    const int Kdim = g.K;
    const int ktiles = (Kdim + 63) / 64;
    const int ntaps = CONV ? g.ntaps : 1;
    const int nk = ktiles * ntaps;
    // ---- staging: buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds).  A and W are raw buffers (stride 0); a lane's
    // address is a loop-invariant 32-bit byte offset (row start + its 16-B chunk) in a VGPR plus a wave-uniform scalar
    // offset (K position, conv tap) in an SGPR, so a DMA issue costs no 64-bit VALU address arithmetic; padding (K tails,
    // the zero halo of the conv) is an out-of-range offset, which the buffer unit turns into zeros.
    constexpr uint32_t OOB = 0x80000000u;                          // >= num_records: launcher guarantees tensors < 2 GiB
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)OOB, 0x00020000);
    // DMA piece j of a half covers rows (j*8+wave)*8 .. +8; lane -> row +(lane>>3), physical chunk lane&7
    const int srow = wave * 8 + (lane >> 3);                       // + j*64
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);             // logical chunk this lane fetches (same for every j)
    uint32_t a_off[2][GA]; int a_geo[2][GA];                       // byte offsets; a_geo (conv) = ct << 6 | validity bits
    uint32_t b_off[2][GB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            int m = m0 + h * 128 + j * 64 + srow; if (m > g.M - 1) m = g.M - 1;
            if constexpr (CONV) {
                const int w = m % g.Wd; const int t1 = m / g.Wd;
                const int hh = t1 % g.H; const int t2 = t1 / g.H;
                const int vm = (hh > 0 ? 1 : 0) | 2 | (hh < g.H - 1 ? 4 : 0) | (w > 0 ? 8 : 0) | 16 | (w < g.Wd - 1 ? 32 : 0);
                a_geo[h][j] = ((t2 % g.T) << 6) | vm;
                a_off[h][j] = ((uint32_t)m * (uint32_t)g.Cin + schunk * 8) * 2u;
            } else {
                a_geo[h][j] = 0;
                a_off[h][j] = ((uint32_t)m * (uint32_t)g.lda + schunk * 8) * 2u;
            }
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            int n = n0 + h * HB + j * 64 + srow; if (n > g.N - 1) n = g.N - 1;
            b_off[h][j] = ((uint32_t)n * (uint32_t)Kdim + schunk * 8) * 2u;
        }
    }
    const uint32_t frame_bytes = CONV ? (uint32_t)g.H * g.Wd * g.Cin * 2u : 0u;
    // stage cursor: K-tile being staged -> (tap, k offset); advanced after the 4th half (A1) of a K-tile
    int c_kk = 0, c_it = 0, c_ih = 0, c_iw = 0, c_tap = 0, c_tile = 0;
    auto cursor_next = [&]() {
        // same K-step order as gemm_big's conv mode: frame tap, 64-channel slice, in-plane taps (fastest)
        ++c_tile;
        if constexpr (CONV) {
            if (++c_iw == g.kw) { c_iw = 0; if (++c_ih == g.kh) { c_ih = 0; if (++c_kk == ktiles) { c_kk = 0; ++c_it; } } }
            c_tap = (c_it * g.kh + c_ih) * g.kw + c_iw;
        } else {
            ++c_kk;
        }
    };
    auto dma = [&](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, unsigned char* lds) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
    };
    // which: 0 = B0, 1 = A0, 2 = B1, 3 = A1 (compile-time after inlining) -- the staging order of a K-tile
    auto stage = [&](int which) {
        unsigned char* buf = p8_smem + (c_tile & 1) * BUF;
        const int kbase = c_kk * 64;
        const bool kin = kbase + schunk * 8 < Kdim;                // only the last K-tile of a ragged K can fail
        if (which == 1 || which == 3) {
            const int h = which == 3 ? 1 : 0;
            unsigned char* dst = buf + h * A_HALF;
            int dt = 0, vbit = 0; uint32_t soff = (uint32_t)kbase * 2u;
            if constexpr (CONV) {
                const int dh = c_ih - g.kh / 2, dw = c_iw - g.kw / 2;
                dt = c_it - g.pad_t;
                vbit = (1 << (dh + 1)) | (8 << (dw + 1));
                soff += (uint32_t)((dh * g.Wd + dw) * g.Cin * 2);   // may wrap: added modulo 2^32 to the lane offset below
            }
#pragma unroll
            for (int j = 0; j < GA; ++j) {
                uint32_t voff = a_off[h][j];
                bool ok = kin;
                if constexpr (CONV) {
                    const int ct = a_geo[h][j] >> 6;
                    int tt = ct + dt; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);
                    // the tap delta is folded into the VGPR offset (it can be negative; the scalar offset is unsigned)
                    voff += (uint32_t)(tt - ct) * frame_bytes + soff;
                    ok = ok && (a_geo[h][j] & vbit) == vbit;
                    dma(ra, ok ? voff : OOB, 0u, dst + (j * 8 + wave) * 1024);
                } else {
                    dma(ra, ok ? voff : OOB, soff, dst + (j * 8 + wave) * 1024);
                }
            }
        } else {
            const int h = which == 2 ? 1 : 0;
            unsigned char* dst = buf + 2 * A_HALF + h * B_HALF;
            const uint32_t soff = ((uint32_t)c_tap * (uint32_t)g.N * (uint32_t)Kdim + (uint32_t)kbase) * 2u;
#pragma unroll
            for (int j = 0; j < GB; ++j) dma(rw, kin ? b_off[h][j] : OOB, soff, dst + (j * 8 + wave) * 1024);
        }
        if (which == 3) cursor_next();
    };

    // ---- fragment read lanes: row = wave base + f*16 + (lane&15), chunk = kb*4 + (lane>>4), swizzle depends on lane&15 only
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (frow >> 1) & 7;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        a_rd[kb] = (wm * WMH + frow) * 128 + (((kb * 4 + fq) ^ fsw) << 4);
        b_rd[kb] = 2 * A_HALF + (wn * WNH + frow) * 128 + (((kb * 4 + fq) ^ fsw) << 4);
    }

    f32x4 acc[2][2][FMQ][FNQ];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < FMQ; ++i)
#pragma unroll
                for (int j = 0; j < FNQ; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Chunk16 af[FMQ][2], bf0[FNQ][2], bf1[FNQ][2];

    auto read_a = [&](const unsigned char* base, int h) {
#pragma unroll
        for (int f = 0; f < FMQ; ++f)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) af[f][kb].u = *reinterpret_cast<const u32x4*>(base + a_rd[kb] + h * A_HALF + f * 2048);
    };
    auto read_b = [&](const unsigned char* base, int h, Chunk16 (&bf)[FNQ][2]) {
#pragma unroll
        for (int f = 0; f < FNQ; ++f)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) bf[f][kb].u = *reinterpret_cast<const u32x4*>(base + b_rd[kb] + h * B_HALF + f * 2048);
    };
    auto quad = [&](f32x4 (&c)[FMQ][FNQ], const Chunk16 (&bf)[FNQ][2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int fm = 0; fm < FMQ; ++fm)
#pragma unroll
                for (int fn = 0; fn < FNQ; ++fn) c[fm][fn] = Mma<bf16_t>::run(bf[fn][kb], af[fm][kb], c[fm][fn]);
    };
#define P8_SYNC_COMPUTE(ACC, BF, NVM)                                           \
    do {                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");              \
        __builtin_amdgcn_s_barrier();                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                      \
        __builtin_amdgcn_s_setprio(1);                                          \
        quad(ACC, BF);                                                          \
        __builtin_amdgcn_s_setprio(0);                                          \
        __builtin_amdgcn_sched_barrier(0);                                      \
        __builtin_amdgcn_s_barrier();                                           \
    } while (0)

    // ---- prologue: half-tiles 0..5 (K-tile 0 complete, B0/A0 of K-tile 1); nk >= 2 is a launch precondition
    stage(0); stage(1); stage(2); stage(3); stage(0); stage(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_INFLIGHT) : "memory");         // half-tiles 0,1 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    read_b(p8_smem, 0, bf0);                                                   // "phase -1": B0 of K-tile 0
    if (grp == 1) __builtin_amdgcn_s_barrier();                                // group 1 runs one barrier behind

    // One K-tile = 4 phases.  MODE 0 = steady (every phase stages), 1 = K-tile nk-2 (only B1,A1 of the last K-tile
    // left to stage), 2 = K-tile nk-1 (drain).  bf0 holds B0(t); bf1 receives B1(t), then B0(t+1), which moves to bf0
    // after the last quadrant (16 v_mov per K-tile keep one instruction stream for both LDS buffers).
    auto ktile = [&](auto MODE, int t) {
        constexpr int mode = decltype(MODE)::value;
        const unsigned char* base = p8_smem + (t & 1) * BUF;
        const unsigned char* nbase = p8_smem + ((t & 1) ^ 1) * BUF;
        read_a(base, 0);
        if constexpr (mode <= 1) stage(2);
        P8_SYNC_COMPUTE(acc[0][0], bf0, mode == 2 ? GA : VM_INFLIGHT);
        read_b(base, 1, bf1);
        if constexpr (mode <= 1) stage(3);
        P8_SYNC_COMPUTE(acc[0][1], bf1, mode == 2 ? 0 : VM_INFLIGHT);
        read_a(base, 1);
        if constexpr (mode == 0) stage(0);
        P8_SYNC_COMPUTE(acc[1][1], bf1, mode == 0 ? VM_INFLIGHT : (mode == 1 ? 2 * GA + GB : 0));
        if constexpr (mode <= 1) read_b(nbase, 0, bf1);
        if constexpr (mode == 0) stage(1);
        P8_SYNC_COMPUTE(acc[1][0], bf0, mode == 0 ? VM_INFLIGHT : (mode == 1 ? GA + GB : 0));
        if constexpr (mode <= 1) {
#pragma unroll
            for (int f = 0; f < FNQ; ++f) { bf0[f][0].u = bf1[f][0].u; bf0[f][1].u = bf1[f][1].u; }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(I0{}, t);
    ktile(I1{}, t); ++t;
    ktile(I2{}, t);
    if (grp == 0) __builtin_amdgcn_s_barrier();                                // re-balance the barrier count
#undef P8_SYNC_COMPUTE

#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int fm = 0; fm < FMQ; ++fm) {
            const int m = m0 + hm * 128 + wm * WMH + fm * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int fn = 0; fn < FNQ; ++fn) {
                    const int nb = n0 + hn * HB + wn * WNH + fn * 16 + 4 * fq;
                    if (nb >= g.N) continue;
                    float v[4] = {acc[hm][hn][fm][fn][0], acc[hm][hn][fm][fn][1], acc[hm][hn][fm][fn][2], acc[hm][hn][fm][fn][3]};
                    epilogue<bf16_t, EPI>(g, m, nb, v);
                }
        }
}

template <int BN, int WGM, int WGN, int EPI, bool CONV>
int launch_one8(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = 2 * (2 * 128 * 128 + BN * 128);
    static std::atomic<unsigned long long> attr_devs{0};
    auto kern = gemm_p8_kernel<BN, WGM, WGN, EPI, CONV>;
    LTX_TRY(ltx_set_max_dyn_smem(attr_devs, reinterpret_cast<const void*>(kern), smem));
    dim3 grid((unsigned)(cdiv(g.M, 256) * cdiv(g.N, BN))), block(512);
    LTX_LAUNCH_TIMED(kern, grid, block, smem, s, g);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

template <int BN, int WGM, int WGN, bool CONV>
int launch_epi8(const GemmArgs& g, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_one8<BN, WGM, WGN, EPI_BIAS, CONV>(g, s);
        case EPI_GELU: return launch_one8<BN, WGM, WGN, EPI_GELU, CONV>(g, s);
        case EPI_GATE_RESID: return launch_one8<BN, WGM, WGN, EPI_GATE_RESID, CONV>(g, s);
        case EPI_RESID: return launch_one8<BN, WGM, WGN, EPI_RESID, CONV>(g, s);
        case EPI_D2S: if constexpr (CONV) return launch_one8<BN, WGM, WGN, EPI_D2S, CONV>(g, s); break;
        case EPI_UNPATCH: if constexpr (CONV) return launch_one8<BN, WGM, WGN, EPI_UNPATCH, CONV>(g, s); break;
    }
    LTX_FAIL(LTX_ERR_ARG, "gemm_p8: bad epilogue");
}

}  // namespace


// buffer-addressed staging uses 32-bit byte offsets with 0x80000000 as the out-of-range marker
bool ltx_gemm_p8_fits(const GemmArgs& g) {
    const double a_bytes = g.conv ? (double)g.M * g.Cin * 2.0 : (double)g.M * g.lda * 2.0;
    const double w_bytes = (double)(g.conv ? g.ntaps : 1) * g.N * g.K * 2.0;
    return a_bytes < 2147483648.0 && w_bytes < 2147483648.0;
}

int ltx_launch_gemm_p8(const GemmArgs& g, int epi, int bn, hipStream_t s) {
    ltx_prof_kernel(LTX_PROFK_GEMM_P8);
    if ((g.K + 63) / 64 * (g.conv ? g.ntaps : 1) < 2) LTX_FAIL(LTX_ERR_ARG, "gemm_p8: needs at least two K-tiles");
    if (!ltx_gemm_p8_fits(g)) LTX_FAIL(LTX_ERR_ARG, "gemm_p8: operands must be smaller than 2 GiB (32-bit buffer offsets)");
    if (bn == 256) return g.conv ? launch_epi8<256, 2, 4, true>(g, epi, s) : launch_epi8<256, 2, 4, false>(g, epi, s);
    return g.conv ? launch_epi8<128, 4, 2, true>(g, epi, s) : launch_epi8<128, 4, 2, false>(g, epi, s);
}
