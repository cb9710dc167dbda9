// 3x3x3 conv3d (LtxVideoCausalConv3d, vae.rs:308-465) as an implicit GEMM whose activation operand is staged ONCE per
// (frame tap, 64-channel slice) for all nine in-plane taps.
//
// gemm_big / gemm_p8 in conv mode re-stage the M-tile's activation rows for every one of the 27 taps; their loads
// (global -> LDS by LDS-DMA) cost 35-50 % of the kernel on the 128..512-channel VAE stages (timing ablations in
// docs/lab_notes.md).  Here the output tile is a 16 x 16 patch of voxels of one frame and LDS holds the patch plus its one-voxel
// rim (18 x 18 "halo rows" of 64 channels, 41 KiB): the 3 x 3 in-plane taps of that frame tap and channel slice are
// nine shifted fragment-read patterns over the same LDS image.  Per nine taps a block stages 41.5 KiB of activation
// (was 9 x 32 KiB for a 256-row tile) plus the nine weight tiles: 2.5 x fewer LDS-DMA instructions per MFMA.
//
//   * 512 threads = 8 waves; BN = 128: 4(M) x 2(N) waves of 64 x 64, BN = 256: 2(M) x 4(N) waves of 128 x 64;
//     v_mfma_f32_16x16x32_bf16, D = Wfrag x Afrag (a lane owns 4 consecutive output channels: shared epilogues);
//   * K-step order = the order of every bf16 conv kernel here (frame tap, channel slice, in-plane tap), same MFMA and
//     k-grouping, so this plan is bit-identical to the others;
//   * halo image double-buffered (next slice's pieces are issued one round per step during the first six steps of the
//     current slice), weight tiles double-buffered, one vmcnt(0) + barrier per step;
//   * zero padding in H/W and the halo rows outside the image are out-of-range buffer offsets (read as zeros); the
//     temporal replicate padding is a clamp of the frame index (vae.rs:374-413).
#include <type_traits>
#include "gemm_common.h"
#include "options.h"

#ifndef HALO_WIDE_EPI
#define HALO_WIDE_EPI 1
#endif
#ifndef HALO_LOADERS
#define HALO_LOADERS 4
#endif
#ifndef HALO_PABL         // timing ablations of the PIPELINED form (wrong results; tools/conv_variants.py): 1 = no barrier, 2 = no LDS-DMA in the loop,
#define HALO_PABL 0       // 4 = fragments read once (no LDS reads in the loop), 8 = no vmcnt wait at the step barrier
#endif
#ifndef HALO_ABL          // timing ablations (wrong results; tools/conv_variants.py): 1 = one A fragment read per k half, 2 = one W fragment read, 4 = no LDS-DMA in the loop
#define HALO_ABL 0
#endif

#ifdef HALO_TRACE         // diagnostic builds (tools/conv_trace.py): per-block stamps on the 100 MHz clock + the shader-clock cycles of the K loop
__device__ unsigned long long g_halo_trace[16384 * 8];
extern "C" int ltx_dbg_halo_trace(unsigned long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_trace), (size_t)n * 8) == hipSuccess ? 0 : -1; }
#define HSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_halo_trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define HCYC(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_halo_trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#define HCYC(i) do { } while (0)
#endif

namespace {

constexpr int ROWB = 128;
constexpr int PH = 16, PW = 16, HW = PW + 2, HROWS = (PH + 2) * HW;      // 18 x 18 = 324 halo rows
constexpr int A_PIECES = (HROWS + 7) / 8;                                  // 41 pieces of 8 rows (1 KiB)
constexpr int A_STAGE = A_PIECES * 1024;
constexpr uint32_t OOB = 0x80000000u;
template <int I, int N, typename F> __device__ __forceinline__ void sfor_h(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor_h<I + 1, N>(f); }
}

__device__ __forceinline__ int swz_h(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

extern __shared__ __attribute__((aligned(16))) unsigned char halo_smem[];

// PIPE (BN = 128, round 3): the per-step barrier sits BETWEEN the two k halves of a step and every fragment is read one half
// ahead into a second register set, so no MFMA waits on an LDS read issued after a barrier (the bubble of the plain form: with
// two waves per SIMD both wait at the same barrier, then both wait for their first fragments); the weight tiles run three
// deep (tile of step s + 2 issued in step s, counted vmcnt at the barrier of step s covers tile s + 1).  LDS-DMA counts per
// step are made static: out-of-range dummy pieces (zeros into padding / free buffers) where the plain form issues nothing.
// (round 5, built, bit-identical, measured and removed - tools/archive/ab_round5/conv_halo_persistent_tile_loop.patch: a PERSISTENT tile loop, one
// block per CU, the next tile's halo image and first two weight tiles issued as the pieces the last group otherwise issues out of
// range.  It hides the tile prologue as designed (tile top -> loop 2.8 -> 1.4 us) but its K loop runs 1205-1233 cycles per step against
// 1156-1170 (same instruction count; 253 instead of 234 VGPRs): 128-channel conv 1743-1770 vs 1590-1609 us, C2 conv class 34.9-35.0
// vs 34.6-34.9 ms: docs/lab_notes.md R5.15)
// ONE WAVE PER SIMD (WGM x WGN = 4 waves, round 4): the pipelined form with per-wave tiles of 128 x 64 - a fragment serves twice
// the MFMAs (12 reads per 32 instead of 8 per 16), the accumulators move to the AGPR half of the 512-register file, and the
// step's LDS-DMA pieces are issued one at a time BETWEEN chunks of MFMAs (pinned with sched_barrier) instead of back to back
// at the top of the step: the timing ablations of the two-waves-per-SIMD form (HALO_PABL, tools/conv_variants.py) put 13 %
// of it on the DMA issue and 14 % on the fragment reads, and the one-wave-per-SIMD GEMM (gemm_asm.hip) showed that a burst
// of pieces stalls the only instruction stream of a SIMD while the same pieces spread over the step cost almost nothing.
template <int BN, int WGM, int WGN, int EPI, bool PIPE>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_halo_kernel(const GemmArgs g) {
    // (the 64-wide tile serves conv_out's 48 columns: three 16-column blocks are multiplied, the fourth - zero weight rows - is not)
    constexpr int NW = WGM * WGN, BM = PH * PW, WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = BN == 64 ? 3 : WN / 16;
    static_assert((NW == 8 || (NW == 4 && PIPE)) && WM % 16 == 0 && WN % 16 == 0, "wave layout");
    constexpr int NTHR = NW * 64;
    // loader waves: the first wave of every SIMD issues all LDS-DMA pieces, its partner (wave + 4) starts on its MFMAs at
    // once (see gemm_big.hip)
    constexpr int NL = (HALO_LOADERS == 4 && (BN == 128 || (BN == 64 && PIPE))) ? 4 : NW;      // measured: +3..9 % at BN = 128, -4 % at BN = 256
    constexpr int AJ = (A_PIECES + NL - 1) / NL, BJ = BN / (8 * NL);
    constexpr int B_STAGE = BN * ROWB;
    constexpr int A_ST = PIPE ? AJ * NL * 1024 : A_STAGE;       // PIPE: padded to whole piece rounds (dummy pieces land in the padding)
    static_assert(!PIPE || (NL == 4 && (BN == 128 || BN == 64)), "pipelined form: four loader waves, 128-wide tile (or conv_out's 64-wide one)");
    unsigned char* Abuf = halo_smem;
    unsigned char* Bbuf = halo_smem + 2 * A_ST;
    HSTAMP(0);
    // the kernel arguments of the tile set-up requested together (left to the compiler: the geometry first, the operand pointers
    // 450 instructions later behind a second scalar-cache miss, in front of the first LDS-DMA piece; gemm_asm.hip)
#ifndef HALO_NO_KERNARG_BATCH
    asm volatile("" :: "s"(g.A), "s"(g.W), "s"(g.C), "s"(g.bias), "s"(g.resid), "s"(g.N), "s"(g.K), "s"(g.Cin), "s"(g.T), "s"(g.H), "s"(g.Wd), "s"(g.ntaps), "s"(g.pad_t),
                 "s"(g.ldc), "s"(g.ldr), "s"(gridDim.x));
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lwave = wave & (NL - 1);
    const bool loader = wave < NL;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- tile: (batch*frame, patch row, patch column, n tile); each XCD gets a contiguous run (neighbouring patches
    // share their rims and every patch's n tiles share the whole halo image through one L2)
    const int ntn = (g.N + BN - 1) / BN, pwn = (g.Wd + PW - 1) / PW, phn = (g.H + PH - 1) / PH;
    int bid = blockIdx.x;
    {
        const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    // (round 5, measured and left out: frames fastest inside a run - one patch position in 32 consecutive frames at a time, so that
    // a frame's patch is fetched once for the three output frames that read it - moved 8 % fewer bytes past the L2s and not a
    // microsecond: profiles/r5a_conv_tile_order_ab.json)
#ifdef HALO_SAME_TILE    // timing experiment (wrong results; tools/conv_trace.py build hot -DHALO_SAME_TILE): every block stages ONE tile's operands, so
    bid = bid % ntn;    // its prologue, halo prefetches and residual rows are served by the L2s (docs/lab_notes.md R6.8)
#endif
    const int nt = bid % ntn; int rr = bid / ntn;
    const int px = rr % pwn; rr /= pwn;
    const int py = rr % phn; const int bt = rr / phn;
    const int t = bt % g.T, b = bt / g.T;
    const int y0 = py * PH, x0 = px * PW, n0 = nt * BN;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
    // The activation descriptor is based at the earliest frame this tile reads (the clamped frame of tap 0), so the 32-bit
    // offsets span at most three frames (3 * H * W * Cin * 2 B < 2 GiB) and the tensor itself may be any size - the 13B
    // model's last stages are 9 and 35 GB (BASELINE C5), which used to fall back to the 128 x 128 register-staged kernel.
    const int64_t frame_elems = (int64_t)g.H * g.Wd * g.Cin;
    const int tt0 = t - g.pad_t < 0 ? 0 : (t - g.pad_t > g.T - 1 ? g.T - 1 : t - g.pad_t);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + ((int64_t)b * g.T + tt0) * frame_elems), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)OOB, 0x00020000);
    auto dma = [&](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, unsigned char* lds) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
    };

    // ---- per-lane staging geometry.  Piece p of the halo image = halo rows 8p .. 8p+7; lane -> (row = lane>>3, physical
    // 16-B chunk = lane&7); the bank swizzle (chunk ^ ((row>>1)&7)) is applied to the SOURCE chunk (lane-linear LDS image).
    const int lr = lane >> 3, pc = lane & 7;
    uint32_t a_voff[AJ], b_voff[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int hrow = (j * NL + lwave) * 8 + lr;
        const int hy = hrow / HW, hx = hrow - hy * HW;
        const int lc = pc ^ ((hx >> 1) & 7);                 // halo image: swizzle by the COLUMN of the halo row (see the reads)
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = hrow < HROWS && y >= 0 && y < g.H && x >= 0 && x < g.Wd;
        a_voff[j] = ok ? (uint32_t)(((y * g.Wd + x) * g.Cin + lc * 8) * 2) : OOB;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = (j * NL + lwave) * 8 + lr;
        const int lc = pc ^ ((row >> 1) & 7);
        b_voff[j] = n0 + row < g.N ? (uint32_t)((((n0 + row) * g.K) + lc * 8) * 2) : OOB;      // weight rows beyond N (BN = 64 on conv_out's 48): zeros
    }
    const int KC = g.Cin / 64;                              // 64-channel slices
    const int KT = g.ntaps / 9;                             // frame taps (3)
    const int G = KT * KC;                                  // (frame tap, slice) groups, nine steps each
    const uint32_t frame_bytes = (uint32_t)g.H * g.Wd * g.Cin * 2u;
    auto issue_a = [&](int grp, int j) {                    // piece round j of group grp's halo image
        const int piece = j * NL + lwave;
        if (!loader || piece >= A_PIECES) return;
        const int it = grp / KC, kc = grp - it * KC;
        int tt = t + it - g.pad_t; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);          // replicate pad on T (vae.rs:374-413)
        const uint32_t soff = (uint32_t)(tt - tt0) * frame_bytes + (uint32_t)kc * 128u;
        dma(ra, a_voff[j], soff, Abuf + (grp & 1) * A_ST + piece * 1024);
    };
    auto issue_b = [&](int grp, int hw, int buf) {          // weight tile of step (grp, hw)
        if (!loader) return;
        const int it = grp / KC, kc = grp - it * KC;
        const uint32_t soff = ((uint32_t)(it * 9 + hw) * (uint32_t)g.N * (uint32_t)g.K + (uint32_t)kc * 64u) * 2u;
#pragma unroll
        for (int j = 0; j < BJ; ++j) dma(rw, b_voff[j], soff, Bbuf + buf * B_STAGE + (j * NL + lwave) * 1024);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Fragment-read addresses of the halo image.  MFMA block fm = patch row wm*FM + fm, lane = patch column frow; tap
    // (ih, iw) reads halo row (prow + ih) * HW + frow + iw.  The bank swizzle is a function of the halo COLUMN
    // (frow + iw), so the address is  rowbase[fm] + colpart[iw][kb] + ih * HW * 128 : the ih shift is an instruction
    // immediate and only 3 x 2 lane-dependent column terms exist (a swizzle by halo row would need one address
    // register per tap, block and k half).
    int rowbase[FM], colpart[3][2];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) rowbase[fm] = ((wm * FM + fm) * HW + frow) * ROWB;
#pragma unroll
    for (int iw = 0; iw < 3; ++iw)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) colpart[iw][kb] = iw * ROWB + (((kb * 4 + fq) ^ (((frow + iw) >> 1) & 7)) << 4);

    if constexpr (PIPE) {
        // uniform issue: every loader wave issues BJ weight pieces and its halo piece rounds in every step, out of range where
        // there is nothing to load (past the last step / group, pieces beyond the image's 41)
        auto issue_a_p = [&](int grp, int j) {
            if (!loader) return;
            const int piece = j * NL + lwave;
            const int it = grp / KC, kc = grp - it * KC;
            int tt = t + it - g.pad_t; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);
            const uint32_t soff = (uint32_t)(tt - tt0) * frame_bytes + (uint32_t)kc * 128u;
            const bool live = grp < G && piece < A_PIECES;
            if ((HALO_PABL & 2) && grp > 0) return;
            dma(ra, live ? a_voff[j] : OOB, live ? soff : 0u, Abuf + (grp & 1) * A_ST + piece * 1024);
        };
        auto issue_b_p = [&](int grp, int hw) {               // weight tile of step (grp, hw) into ring slot hw % 3 (9 steps per group)
            if (!loader) return;
            const int it = grp / KC, kc = grp - it * KC;
            const bool live = grp < G;
            const uint32_t soff = live ? ((uint32_t)(it * 9 + hw) * (uint32_t)g.N * (uint32_t)g.K + (uint32_t)kc * 64u) * 2u : 0u;
            if ((HALO_PABL & 2) && (grp > 0 || hw > 1)) return;
#pragma unroll
            for (int j = 0; j < BJ; ++j) dma(rw, live ? b_voff[j] : OOB, soff, Bbuf + (hw % 3) * B_STAGE + (j * NL + lwave) * 1024);
        };
        Chunk16 w0[FN], a0[FM], w1[FN], a1[FM];
        bool frags_once = false;
        auto read_frags = [&](int grp, int hw, int kb, Chunk16 (&wf)[FN], Chunk16 (&af)[FM]) {
            if (HALO_PABL & 4) {                               // timing ablation: the registers keep their first contents (made opaque)
                if (frags_once) {
#pragma unroll
                    for (int f = 0; f < FN; ++f) asm volatile("" : "+v"(wf[f].u));
#pragma unroll
                    for (int f = 0; f < FM; ++f) asm volatile("" : "+v"(af[f].u));
                    return;
                }
            }
            const int ih = hw / 3, iw = hw % 3;
            const unsigned char* As = Abuf + (grp & 1) * A_ST;
            const unsigned char* Bs = Bbuf + (hw % 3) * B_STAGE;
#pragma unroll
            for (int f = 0; f < FN; ++f) wf[f].u = *reinterpret_cast<const u32x4*>(Bs + swz_h(wn * WN + f * 16 + frow, kb * 4 + fq));
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) af[fm].u = *reinterpret_cast<const u32x4*>(As + rowbase[fm] + colpart[iw][kb] + ih * HW * ROWB);
        };
        auto mma_all = [&](Chunk16 (&wf)[FN], Chunk16 (&af)[FM]) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = Mma<bf16_t>::run(wf[fn], af[fm], acc[fm][fn]);
        };
        auto interleave = [&]() {                             // one fragment read, then its share of the MFMAs
#pragma unroll
            for (int i = 0; i < FN + FM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, FM * FN / (FN + FM), 0);
            }
        };
#pragma unroll
        for (int j = 0; j < AJ; ++j) issue_a_p(0, j);
        issue_b_p(0, 0); issue_b_p(0, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        HSTAMP(1); HCYC(5);
        read_frags(0, 0, 0, w0, a0);
        if (HALO_PABL & 4) { read_frags(0, 0, 1, w1, a1); frags_once = true; }
        auto step_p = [&](int grp, auto hw_tag) {
            constexpr int hw = decltype(hw_tag)::value;
            constexpr int RPS = (AJ + 5) / 6;                 // halo piece rounds per step, over the first six steps of a group
            constexpr int na = (hw + 1) * RPS <= AJ ? RPS : (hw * RPS < AJ ? AJ - hw * RPS : 0);
            if (hw + 2 < 9) issue_b_p(grp, hw + 2); else issue_b_p(grp + 1, hw + 2 - 9);
#pragma unroll
            for (int rr = 0; rr < na; ++rr) issue_a_p(grp + 1, hw * RPS + rr);
            read_frags(grp, hw, 1, w1, a1);                   // second k half of this step: its tiles are long visible
            mma_all(w0, a0);
            interleave();
            // everything issued before this step has landed (tile of step + 1, halo rounds); a raw barrier: __syncthreads() would
            // add a fence, i.e. vmcnt(0), and wait for the tile of step + 2 as well
            if ((HALO_PABL & 9) == 9 || (HALO_PABL & 3) == 3) {}                                     // neither wait nor barrier
            else if (HALO_PABL & 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BJ + na) : "memory");
            else if (HALO_PABL & (8 | 2)) asm volatile("s_barrier" ::: "memory");
            else
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(BJ + na) : "memory");
            if (hw + 1 < 9) read_frags(grp, hw + 1, 0, w0, a0); else read_frags(grp + 1, 0, 0, w0, a0);
            mma_all(w1, a1);
            interleave();
        };
        if constexpr (NW == 4) {
            // ---- one wave per SIMD.  (1) MFMAs as inline asm with the accumulator tied in place in the AGPR file: left to itself
            // hipcc renames the 128 accumulator registers from MFMA to MFMA and pays ~190 v_accvgpr_mov per group to put them back.
            // (2) The scalars of a step's DMA pieces (weight tile of step + 2, halo image of the next group) are carried from group
            // to group instead of being derived from the group number by a division in front of every piece.  (3) The pieces sit
            // between chunks of MFMAs, one per chunk.
            auto mfma = [&](f32x4& c, const Chunk16& w, const Chunk16& a_) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w.u), "v"(a_.u));
            };
            const uint32_t nk2 = (uint32_t)g.N * (uint32_t)g.K * 2u;
            auto a_soff_of = [&](int it, int kc) {
                int tt = t + it - g.pad_t; tt = tt < 0 ? 0 : (tt > g.T - 1 ? g.T - 1 : tt);
                return (uint32_t)(tt - tt0) * frame_bytes + (uint32_t)kc * 128u;
            };
            int it_n = KC > 1 ? 0 : 1, kc_n = KC > 1 ? 1 : 0;       // (frame tap, slice) of group grp + 1
            uint32_t w_cur = 0u;                                      // weight offset of (grp, tap 0)
            for (int grp = 0; grp < G; ++grp) {
                const bool live_n = grp + 1 < G;
                const uint32_t a_next = live_n ? a_soff_of(it_n, kc_n) : 0u;
                const uint32_t w_next = live_n ? ((uint32_t)(it_n * 9) * (uint32_t)g.N * (uint32_t)g.K + (uint32_t)kc_n * 64u) * 2u : 0u;
                unsigned char* const a_dst = Abuf + ((grp + 1) & 1) * A_ST;
                sfor_h<0, 9>([&](auto hw_tag) {
                    constexpr int hw = decltype(hw_tag)::value;
                    constexpr int RPS = (AJ + 5) / 6;
                    constexpr int na = (hw + 1) * RPS <= AJ ? RPS : (hw * RPS < AJ ? AJ - hw * RPS : 0);
                    constexpr int NP = BJ + na, NM = FM * FN;
                    constexpr int h2 = hw + 2 < 9 ? hw + 2 : hw + 2 - 9;
                    const bool w_live = hw + 2 < 9 ? true : live_n;
                    const uint32_t w_soff = (hw + 2 < 9 ? w_cur : w_next) + (uint32_t)h2 * nk2;
                    // one fragment read (of the NEXT half) in front of every second MFMA, the step's NP DMA pieces spread over the
                    // first half; everything in program order, pinned by sched_barrier
                    auto read_one = [&](int rg, int rh, int kb, auto idx_tag, Chunk16 (&wf)[FN], Chunk16 (&af)[FM]) {
                        constexpr int idx = decltype(idx_tag)::value;
                        if (HALO_PABL & 4) return;
                        const int ih = rh / 3, iw = rh % 3;
                        if constexpr (idx < FN) wf[idx].u = *reinterpret_cast<const u32x4*>(Bbuf + (rh % 3) * B_STAGE + swz_h(wn * WN + idx * 16 + frow, kb * 4 + fq));
                        else af[idx - FN].u = *reinterpret_cast<const u32x4*>(Abuf + (rg & 1) * A_ST + rowbase[idx - FN] + colpart[iw][kb] + ih * HW * ROWB);
                    };
                    auto piece = [&](auto i_tag) {
                        constexpr int i = decltype(i_tag)::value;
                        if constexpr (i < BJ) {
                            if (!((HALO_PABL & 2) && (grp > 0 || hw > 1)))
                                dma(rw, w_live ? b_voff[i] : OOB, w_live ? w_soff : 0u, Bbuf + (h2 % 3) * B_STAGE + (i * NL + lwave) * 1024);
                        } else {
                            constexpr int j = hw * RPS + (i - BJ);
                            const bool live = live_n && j * NL + lwave < A_PIECES;
                            if (!(HALO_PABL & 2)) dma(ra, live ? a_voff[j] : OOB, live ? a_next : 0u, a_dst + (j * NL + lwave) * 1024);
                        }
                    };
                    sfor_h<0, NM>([&](auto m_tag) {                 // first k half on (w0, a0); reads of this step's second half into (w1, a1)
                        constexpr int mi = decltype(m_tag)::value;
                        if constexpr (mi % 2 == 0 && mi / 2 < FN + FM) read_one(grp, hw, 1, std::integral_constant<int, mi / 2>{}, w1, a1);
                        if constexpr (mi % 2 == 1 && (mi / 2) % (16 / NP > 0 ? 16 / NP : 1) == 0 && (mi / 2) / (16 / NP > 0 ? 16 / NP : 1) < NP)
                            piece(std::integral_constant<int, (mi / 2) / (16 / NP > 0 ? 16 / NP : 1)>{});
                        mfma(acc[mi / FN][mi % FN], w0[mi % FN], a0[mi / FN]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    if ((HALO_PABL & 9) == 9 || (HALO_PABL & 3) == 3) {}
                    else if (HALO_PABL & 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BJ + na) : "memory");
                    else if (HALO_PABL & (8 | 2)) asm volatile("s_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(BJ + na) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    sfor_h<0, NM>([&](auto m_tag) {                 // second k half on (w1, a1); reads of the next step's first half into (w0, a0)
                        constexpr int mi = decltype(m_tag)::value;
                        if constexpr (mi % 2 == 0 && mi / 2 < FN + FM) {
                            if (hw + 1 < 9) read_one(grp, hw + 1, 0, std::integral_constant<int, mi / 2>{}, w0, a0);
                            else read_one(grp + 1, 0, 0, std::integral_constant<int, mi / 2>{}, w0, a0);
                        }
                        mfma(acc[mi / FN][mi % FN], w1[mi % FN], a1[mi / FN]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                w_cur = w_next;
                if (++kc_n == KC) { kc_n = 0; ++it_n; }
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results before the epilogue reads the accumulators
        } else
        for (int grp = 0; grp < G; ++grp) {
            step_p(grp, std::integral_constant<int, 0>{}); step_p(grp, std::integral_constant<int, 1>{}); step_p(grp, std::integral_constant<int, 2>{});
            step_p(grp, std::integral_constant<int, 3>{}); step_p(grp, std::integral_constant<int, 4>{}); step_p(grp, std::integral_constant<int, 5>{});
            step_p(grp, std::integral_constant<int, 6>{}); step_p(grp, std::integral_constant<int, 7>{}); step_p(grp, std::integral_constant<int, 8>{});
        }
        HCYC(6); HSTAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dummy pieces of the last steps
        __syncthreads();
    } else {
#pragma unroll
    for (int j = 0; j < AJ; ++j) issue_a(0, j);
    issue_b(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto step = [&](int grp, auto hw_tag, int buf) {
        constexpr int hw = decltype(hw_tag)::value;
        constexpr int ih = hw / 3, iw = hw % 3;                     // dh = ih - 1, dw = iw - 1
        // next step's weight tile; next group's halo image, one piece round per step
        if (!(HALO_ABL & 4)) { if (hw < 8) issue_b(grp, hw + 1, buf ^ 1); else if (grp + 1 < G) issue_b(grp + 1, 0, buf ^ 1); }
        constexpr int RPS = (AJ + 8) / 9;                           // halo piece rounds per step
        if (!(HALO_ABL & 4) && grp + 1 < G) {
#pragma unroll
            for (int rr = 0; rr < RPS; ++rr) if (hw * RPS + rr < AJ) issue_a(grp + 1, hw * RPS + rr);
        }
        const unsigned char* As = Abuf + (grp & 1) * A_ST;
        const unsigned char* Bs = Bbuf + buf * B_STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Chunk16 af[FM], wf[FN];
#pragma unroll
            for (int f = 0; f < FN; ++f) {
                if ((HALO_ABL & 2) && f > 0) { wf[f].u = wf[0].u + (u32x4){(uint32_t)f, 0u, 0u, 0u}; continue; }
                wf[f].u = *reinterpret_cast<const u32x4*>(Bs + swz_h(wn * WN + f * 16 + frow, kb * 4 + fq));
            }
            // (the six column terms are recomputed where they are used: kept in registers across the nine steps they were spilled -
            // 256 registers at two waves per SIMD - and reloaded from scratch in front of their step's first fragment read)
            int fr_ = frow; asm volatile("" : "+v"(fr_));
            const int cp = BN == 256 ? iw * ROWB + (((kb * 4 + fq) ^ (((fr_ + iw) >> 1) & 7)) << 4) : colpart[iw][kb];
            af[0].u = *reinterpret_cast<const u32x4*>(As + rowbase[0] + cp + ih * HW * ROWB);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                if ((HALO_ABL & 1) && fm + 1 < FM) af[fm + 1].u = af[0].u + (u32x4){(uint32_t)fm, 0u, 0u, 0u};
                else
                if (fm + 1 < FM) af[fm + 1].u = *reinterpret_cast<const u32x4*>(As + rowbase[fm + 1] + cp + ih * HW * ROWB);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = Mma<bf16_t>::run(wf[fn], af[fm], acc[fm][fn]);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            __builtin_amdgcn_sched_group_barrier(0x100, FN + 1, 0);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                if (fm + 1 < FM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int grp = 0; grp < G; ++grp) {
        // nine steps = 9 weight buffers alternations: the parity of the buffer flips between groups (9 is odd)
        const int p = grp & 1;
        step(grp, std::integral_constant<int, 0>{}, p);
        step(grp, std::integral_constant<int, 1>{}, p ^ 1);
        step(grp, std::integral_constant<int, 2>{}, p);
        step(grp, std::integral_constant<int, 3>{}, p ^ 1);
        step(grp, std::integral_constant<int, 4>{}, p);
        step(grp, std::integral_constant<int, 5>{}, p ^ 1);
        step(grp, std::integral_constant<int, 6>{}, p);
        step(grp, std::integral_constant<int, 7>{}, p ^ 1);
        step(grp, std::integral_constant<int, 8>{}, p);
    }

    }
    // ---- wide epilogue (bias / residual): the result tile goes through LDS and leaves as 16-byte stores along the channel
    // rows, the residual tile comes in the same way by LDS-DMA (see gemm_big.hip); same arithmetic as epilogue()
    constexpr bool WIDE = HALO_WIDE_EPI && (EPI == EPI_BIAS || EPI == EPI_RESID);
    if constexpr (WIDE) {
        if (g.wide_epi) {
            constexpr int CPR = BN / 8, XM = 15, RPP = 64 / CPR;
            // the residual is addressed inside the tile's own frame (32-bit offsets span one frame of the output)
            const int64_t frame_row0 = ((int64_t)b * g.T + t) * g.H * g.Wd;
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(g.resid ? g.resid : g.C) + frame_row0 * g.ldr), 0, (int)OOB, 0x00020000);
            auto row_m = [&](int row, bool& inside) {                 // tile row (patch voxel) -> output row index
                const int y = y0 + (row >> 4), x = x0 + (row & 15);
                inside = y < g.H && x < g.Wd;
                return ((b * g.T + t) * g.H + y) * g.Wd + x;
            };
            if constexpr (EPI == EPI_RESID) {
                for (int pi = wave; pi < BM / RPP; pi += NW) {
                    const int row = pi * RPP + lane / CPR, pc = lane % CPR;
                    const int lc = pc ^ (row & XM);
                    bool inside; const int m = row_m(row, inside);
                    dma(rr, inside ? (uint32_t)((((int64_t)m - frame_row0) * g.ldr + n0 + lc * 8) * 2) : OOB, 0u, halo_smem + pi * 1024);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // The bias of this lane's FN column groups, loaded ONCE and together (round 6: the load used to sit inside the fm / fn loops
            // - FM x FN dependent global loads per thread, each waited for with vmcnt(0) before its add: 35 drains per tile, about
            // 3 us of a 46-us tile on a kernel whose epilogue nothing overlaps).  Same values, same adds: same bits.
            float bias4[FN][4];
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bias4[fn][i] = 0.f;
            }
            if (g.bias) {
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + n0 + wn * WN + fn * 16 + 4 * fq, bias4[fn]);
            }
            __syncthreads();
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int row = (wm * FM + fm) * 16 + frow;
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const int col = wn * WN + fn * 16 + 4 * fq;
                    unsigned char* slot = halo_smem + row * (CPR * 16) + (((col >> 3) ^ (row & XM)) << 4) + ((col >> 2) & 1) * 8;
                    float v[4] = {acc[fm][fn][0], acc[fm][fn][1], acc[fm][fn][2], acc[fm][fn][3]};
                    {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += bias4[fn][i];
                    }
                    if constexpr (EPI == EPI_RESID) {
                        float r[4];
                        load4<bf16_t>(reinterpret_cast<const bf16_t*>(slot), r);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += r[i];
                    }
                    store4<bf16_t>(reinterpret_cast<bf16_t*>(slot), v);
                }
            }
            __syncthreads();
            bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
            // a thread's chunk column is the same in every iteration (NTHR % CPR == 0): its modulation values load once
            float pn_sc[8], pn_sh[8];
            bool pn_mod = false;
            // (round 5, measured and removed: the residual epilogue writing a second, normalised copy of its rows - the next resnet's
            // norm1 - cost the conv class +2.5 ms per C2 video for the 2.7 ms of norm passes it replaced: a tile's epilogue is not
            // overlapped with anything, so work moved into it is paid in full; profiles/r5g_vae_norm1_in_conv2_epilogue_ab.jsonl)
            if constexpr (EPI == EPI_BIAS) {
                if (g.pn_on && g.pn_scale) {
                    pn_mod = true;
                    const int c0 = tid % CPR;
                    const f32x4* scp = reinterpret_cast<const f32x4*>(g.pn_scale + (int64_t)b * g.pn_mod_stride + c0 * 8);
                    const f32x4* shp = reinterpret_cast<const f32x4*>(g.pn_shift + (int64_t)b * g.pn_mod_stride + c0 * 8);
                    const f32x4 s0 = scp[0], s1 = scp[1], h0 = shp[0], h1 = shp[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pn_sc[j] = 1.0f + s0[j]; pn_sc[4 + j] = 1.0f + s1[j]; pn_sh[j] = h0[j]; pn_sh[4 + j] = h1[j]; }
                }
            }
            for (int id = tid; id < BM * CPR; id += NTHR) {
                const int row = id / CPR, c = id - row * CPR;
                bool inside; const int m = row_m(row, inside);
                Chunk16 cc; cc.u = *reinterpret_cast<const u32x4*>(halo_smem + row * (CPR * 16) + ((c ^ (row & XM)) << 4));
                if constexpr (EPI == EPI_BIAS) {
                    if (g.pn_on) {                                   // the tile spans all N channels: CPR consecutive lanes hold one voxel's row
                        float f[8]; chunk_to_f32<bf16_t>(cc, f);
                        float ss = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
#pragma unroll
                        for (int o = CPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
                        const float rinv = __builtin_amdgcn_rsqf(ss * (1.0f / (float)BN) + g.pn_eps);      // as rownorm.hip's narrow rows
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float n = f[j] * rinv;
                            if (pn_mod) n = n * pn_sc[j] + pn_sh[j];
                            if (g.pn_act == 1) n = silu_f(n);
                            f[j] = n;
                        }
                        f32_to_chunk<bf16_t>(f, cc);
                    }
                }
                if (!inside) continue;
                *reinterpret_cast<u32x4*>(Cb + (int64_t)m * g.ldc + n0 + c * 8) = cc.u;
            }
#ifdef HALO_TRACE
            HSTAMP(3); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); HSTAMP(4);
            if (threadIdx.x == 0 && blockIdx.x < 16384) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_halo_trace[blockIdx.x * 8 + 7] = hw; }
#endif
            return;
        }
    }
    // Fragment-wise epilogues (depth-to-space, unpatchify, narrow bias / residual outputs).  Round 6: every load of the tile - the bias
    // of the lane's FN column groups, the depth-to-space residual's gathers - is issued FIRST and waited for once; inside the fm / fn
    // loops each fragment's load waited vmcnt(0), which on gfx950 also waits for the STORE of the fragment before it: FM x FN
    // store -> load round trips in a row on an epilogue nothing overlaps.  Same values added in the same order: same bits.
    if constexpr (EPI == EPI_D2S || EPI == EPI_UNPATCH || EPI == EPI_BIAS) {
        float bias4[FN][4];
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int nb = n0 + wn * WN + fn * 16 + 4 * fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) bias4[fn][i] = 0.f;
        }
        if (g.bias) {
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                int nb = n0 + wn * WN + fn * 16 + 4 * fq; nb = nb < g.N ? nb : g.N - 4;
                load4<bf16_t>(reinterpret_cast<const bf16_t*>(g.bias) + nb, bias4[fn]);
            }
        }
        const bool d2s_res = EPI == EPI_D2S && g.resid != nullptr;
        GemmArgs gs = g; gs.bias = nullptr; gs.resid = nullptr;      // epilogue() then only places and stores
        // rows in chunks of FC: a chunk's residual gathers are all in flight together (one wait per chunk - it also waits for the
        // stores of the chunk before, the only serialisation left), FC sized so that nothing is spilled (8 x 4 fragments: 2 rows)
        constexpr int FC = FM * FN > 16 ? 2 : FM;
#pragma unroll
        for (int f0 = 0; f0 < FM; f0 += FC) {
            bf16_t raw[FC][FN][4];
            if constexpr (EPI == EPI_D2S) {
                if (d2s_res) {                              // (uniform; inside: no condition in front of a load - rows / columns outside the tensor read clamped addresses and are never stored)
                    const int nsub = g.d2s_sp ? 4 : 8;
#pragma unroll
                    for (int fc = 0; fc < FC; ++fc) {
                        int y = y0 + wm * FM + f0 + fc, x = x0 + frow;
                        y = y < g.H ? y : g.H - 1; x = x < g.Wd ? x : g.Wd - 1;
                        const int m = ((b * g.T + t) * g.H + y) * g.Wd + x;
                        const bf16_t* xr = reinterpret_cast<const bf16_t*>(g.resid) + (int64_t)m * g.Cin;
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn) {
                            int nb = n0 + wn * WN + fn * 16 + 4 * fq; nb = nb < g.N ? nb : g.N - 4;
                            const int sidx = nb / g.Cf, co = nb - sidx * g.Cf;
#pragma unroll
                            for (int i = 0; i < 4; ++i) raw[fc][fn][i] = xr[((co + i) % g.Cr) * nsub + sidx];
                        }
                    }
                }
            }
#pragma unroll
            for (int fc = 0; fc < FC; ++fc) {
                const int fm = f0 + fc;
                const int y = y0 + wm * FM + fm, x = x0 + frow;
                if (y >= g.H || x >= g.Wd) continue;
                const int m = ((b * g.T + t) * g.H + y) * g.Wd + x;
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const int nb = n0 + wn * WN + fn * 16 + 4 * fq;
                    if (nb >= g.N) continue;                // N % 4 == 0: a 4-column group is inside or outside as a whole
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[fm][fn][i] + bias4[fn][i];
                    if constexpr (EPI == EPI_D2S) {
                        if (d2s_res) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += to_f32(raw[fc][fn][i]);
                        }
                    }
                    epilogue<bf16_t, EPI>(gs, m, nb, v);
                }
            }
        }
    } else {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int y = y0 + wm * FM + fm, x = x0 + frow;
        if (y >= g.H || x >= g.Wd) continue;
        const int m = ((b * g.T + t) * g.H + y) * g.Wd + x;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int nb = n0 + wn * WN + fn * 16 + 4 * fq;
            if (nb >= g.N) continue;                        // N % 4 == 0: a 4-column group is inside or outside as a whole
            float v[4] = {acc[fm][fn][0], acc[fm][fn][1], acc[fm][fn][2], acc[fm][fn][3]};
            epilogue<bf16_t, EPI>(g, m, nb, v);
        }
    }
    }
}

template <int BN, int WGM, int WGN, int EPI, bool PIPE = false>
int launch_halo(const GemmArgs& g, hipStream_t s) {
    constexpr int smem = PIPE ? 2 * 44 * 1024 + 3 * BN * ROWB : 2 * A_STAGE + 2 * BN * ROWB;
    static std::atomic<unsigned long long> attr_devs{0};
    auto kern = conv_halo_kernel<BN, WGM, WGN, EPI, PIPE>;
    LTX_TRY(ltx_set_max_dyn_smem(attr_devs, reinterpret_cast<const void*>(kern), smem));
    const int tiles = g.B * g.T * cdiv(g.H, PH) * cdiv(g.Wd, PW) * cdiv(g.N, BN);
    GemmArgs ga = g;
    ga.wide_epi = ltx_opt().gemm_wide_epi && g.ldc % 8 == 0 && ((uintptr_t)g.C & 15) == 0 &&
                  (!g.resid || (g.ldr % 8 == 0 && ((uintptr_t)g.resid & 15) == 0 && (double)g.H * g.Wd * g.ldr * 2.0 < 2147483648.0));
    if (g.pn_on && (!ga.wide_epi || EPI != EPI_BIAS || BN != g.N || !HALO_WIDE_EPI)) LTX_FAIL(LTX_ERR_ARG, "conv_halo: the fused output norm needs the wide bias epilogue and BN == N");
    LTX_LAUNCH_TIMED(kern, dim3((unsigned)tiles), dim3(WGM * WGN * 64), smem, s, ga);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

template <int BN, int WGM, int WGN, bool PIPE = false>
int launch_halo_epi(const GemmArgs& g, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_halo<BN, WGM, WGN, EPI_BIAS, PIPE>(g, s);
        case EPI_RESID: return launch_halo<BN, WGM, WGN, EPI_RESID, PIPE>(g, s);
        case EPI_D2S: return launch_halo<BN, WGM, WGN, EPI_D2S, PIPE>(g, s);
    }
    LTX_FAIL(LTX_ERR_ARG, "conv_halo: unsupported epilogue");
}

}  // namespace

// whether the halo-staged kernel can run this conv with tile width bn (128 / 256)
bool ltx_conv_halo_eligible(const GemmArgs& g, int epi, int bn) {
    if (!g.conv || g.ntaps != 27 || g.kh != 3 || g.kw != 3) return false;
    if (g.Cin % 64 != 0 || g.K != g.Cin) return false;
    if (bn == 64) {                                         // the narrow tile: conv_out (N = 48, unpatchify epilogue), one n tile
        if (epi != EPI_UNPATCH || g.N > 48 || g.N % 4 != 0) return false;
    } else {
        if (g.N % bn != 0) return false;
        if (epi != EPI_BIAS && epi != EPI_RESID && epi != EPI_D2S) return false;
    }
    if (g.c_seg_shift) return false;
    // a tile addresses the three frames around it: those (not the tensor) and the weights must stay below 2 GiB
    const double frame_bytes = (double)g.H * g.Wd * g.Cin * 2.0, w_bytes = 27.0 * g.N * g.K * 2.0;
    return 3.0 * frame_bytes < 2147483648.0 && w_bytes < 2147483648.0 && (double)g.B * g.T * g.H * g.Wd < 2147483648.0;
}

int ltx_launch_conv_halo(const GemmArgs& g, int epi, int bn, hipStream_t s) {
    ltx_prof_kernel(LTX_PROFK_CONV_HALO);
    if (!ltx_conv_halo_eligible(g, epi, bn)) LTX_FAIL(LTX_ERR_ARG, "conv_halo: shape not eligible");
#ifdef HALO_CONV_OUT_PLAIN
    if (bn == 64) return launch_halo<64, 8, 1, EPI_UNPATCH, false>(g, s);      // conv_out: eight waves of 32 x 64, barrier-per-step form
#else
    if (bn == 64) return launch_halo<64, 8, 1, EPI_UNPATCH, true>(g, s);       // conv_out: eight waves of 32 x 64, pipelined form
#endif
    if (bn == 256) return launch_halo_epi<256, 2, 4>(g, epi, s);
#if HALO_LOADERS == 4
#ifdef LTX_EXPERIMENTS     // x_conv_halo_pipe=0: the barrier-per-step form; x_conv_halo_w4=1: one wave per SIMD (four waves of 128 x 64; round 4: -3 %)
    if (!ltx_exp("conv_halo_pipe", 1)) return launch_halo_epi<128, 4, 2>(g, epi, s);
    if (ltx_exp("conv_halo_w4", 0)) return launch_halo_epi<128, 2, 2, true>(g, epi, s);
#endif
    return launch_halo_epi<128, 4, 2, true>(g, epi, s);
#else
    return launch_halo_epi<128, 4, 2>(g, epi, s);
#endif
}
