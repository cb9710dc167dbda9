// Internal launcher interface between the C++ orchestration (dit.hip / vae.hip /
// pipeline.cpp / capi_ops.hip) and the HIP kernels.  Not part of the public C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

enum { LTX_DT_F32 = 0, LTX_DT_BF16 = 1 };
static inline size_t ltx_dt_size(int dt) { return dt == LTX_DT_BF16 ? 2 : 4; }

// ---------------- optional per-kernel-class event timing (prof.hip) ----------------
enum { LTX_PROF_GEMM = 0, LTX_PROF_CONV = 1, LTX_PROF_ATTN_SELF = 2, LTX_PROF_ATTN_CROSS = 3, LTX_PROF_ROWNORM = 4, LTX_PROF_NKINDS = 5 };
bool ltx_prof_begin(int kind, double work, hipStream_t s, void** token);   // work: algorithmic flops (or bytes for ROWNORM)
void ltx_prof_end(void* token, hipStream_t s);
// which KERNEL served the launch being timed (set by the launcher that actually enqueues it; read by ltx_prof_end)
enum { LTX_PROFK_GEMM128 = 0, LTX_PROFK_GEMM_BIG = 1, LTX_PROFK_GEMM_P8 = 2, LTX_PROFK_CONV_HALO = 3, LTX_PROFK_GEMM_ASM = 4, LTX_PROFK_GEMM_ASM16 = 5, LTX_PROFK_GEMM_RING = 6, LTX_PROFK_N = 7 };
void ltx_prof_kernel(int which);
// Kernel-level start / stop events for the launch being timed (the dispatch packet's own timestamps, what rocprofv3's kernel
// trace reports): stream-level hipEventRecord brackets also count the dispatch latency and the end-of-kernel release that the
// un-profiled pipeline overlaps with the neighbouring kernels (+5..10 us on a 180 us launch).  True at most ONCE per
// ltx_prof_begin on this thread; a launch that enqueues more than one kernel keeps its stream-level bracket.
bool ltx_prof_kernel_events(hipEvent_t* a, hipEvent_t* b);
#define LTX_LAUNCH_TIMED(kernel, grid, block, shmem, stream, ...) do {                                           \
        hipEvent_t _pa = nullptr, _pb = nullptr;                                                                   \
        if (ltx_prof_kernel_events(&_pa, &_pb)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, _pa, _pb, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                  \
    } while (0)

// ---------------- GEMM / implicit-GEMM conv (gemm.hip) ----------------
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_GATE_RESID = 2, EPI_RESID = 3, EPI_D2S = 4, EPI_UNPATCH = 5 };

struct GemmArgs {
    const void* A = nullptr;      // [M, lda] (linear) or channels-last activation [B,T,H,W,Cin] (conv)
    const void* W = nullptr;      // [N, K] (linear) or [ntaps][N][K] (conv), K contiguous
    void* C = nullptr;
    const void* bias = nullptr;   // T [N] or null
    const void* resid = nullptr;  // T [M, ldr] (GATE_RESID / RESID) or conv input x (D2S residual)
    const float* gate = nullptr;  // f32 [batch, gate_stride] (GATE_RESID)
    int M = 0, N = 0, K = 0;      // K = Cin in conv mode
    int lda = 0, ldc = 0, ldr = 0;
    int rows_per_batch = 1, gate_stride = 0;
    // conv geometry
    int conv = 0;
    int B = 1, T = 1, H = 1, Wd = 1, Cin = 0;
    int ntaps = 1, kh = 1, kw = 1, pad_t = 0;   // pad_t: frames of left temporal replicate padding
    // D2S / UNPATCH
    int Cf = 0, Cr = 0, To = 0, Ho = 0, Wo = 0, post = 0;
    int d2s_sp = 0;                 // EPI_D2S: 1 = the spatial-only (1, 2, 2) depth-to-space of an up-block without temporal scaling (N = 4 Cf, To = T, no frame dropped; vae.rs:1225-1236)
    // EPI_BIAS only: output split into column segments of width 1 << c_seg_shift, segment j a dense [M, ldc] matrix at
    // C + j * c_seg_stride elements (fused q|k|v projection -> three contiguous matrices); 0 = one [M, ldc] matrix
    int c_seg_shift = 0; int64_t c_seg_stride = 0;
    // Optional by-product (linear layers): per-row partial sums of squares of the OUTPUT as stored (after the epilogue,
    // rounded to T), one f32 per 128-column group: rowsq[m * ceil(N / 128) + g].  The consumer's row norm then needs no pass
    // over the matrix (cross-attention q-norm, dit.hip).  Summation order is CANONICAL - a fixed function of (m, g) alone,
    // ltx_rowsq_leaf / ltx_launch_rowsq - so the value does not depend on which kernel the plan picked: gemm_asm16's
    // epilogue produces it in place, every other kernel is followed by the stand-alone pass (ltx_launch_gemm).
    float* rowsq = nullptr;
    // Norm fold (round 6; bf16, gemm_asm16's wide epilogue only - dit.hip asks ltx_gemm_fold_ok).  RMS norm + modulation
    //   y = h * r_m * (1 + sc) + sh,  r_m = 1 / sqrt(mean(h_m^2) + eps)     (LtxVideoTransformerBlock::forward, ltx_transformer.rs:847-851, 905-909)
    // followed by a linear layer equals  r_m * ((h (.) (1 + sc)) W^T) + (sh W^T + b): the layer that WRITES h also stores
    //   C2[m][n] = bf16(float(C[m][n] as stored) * (1 + scale2[b(m)][n]))     (RESID / GATE_RESID; leading dimension ldc)
    // and the layer that READS the normalised rows takes A = C2 and finishes  out = epi(r_m * acc + cvec[b(m)][n])  with r_m from
    // the producer's row partials rs_sq[m][0 .. rs_n) (GemmArgs::rowsq of that launch) and cvec = sh W^T + b in f32
    // (ltx_launch_shift_gemv, cached per timestep): the stand-alone norm pass (a read and a write of [M, D]) disappears.
    // b(m) = m / rows_per_batch in both.
    void* C2 = nullptr; const float* scale2 = nullptr; int scale2_stride = 0;
    const float* rs_sq = nullptr; int rs_n = 0, rs_D = 0; float rs_eps = 0.f; const float* cvec = nullptr; int cvec_stride = 0;
    // Optional second copy of W for gemm_ring.hip (linear layers of at most 512 rows): [ceil(N/32)][ceil(K/64)][32 rows][64] bf16,
    // zero padded - a tile's K-step is then 4-KiB blocks and a block's whole K range one contiguous stream, instead of 128-byte
    // pieces of rows K * 2 bytes apart (ltx_pack_ring_weights; same values; measured neutral: opt-in, LTX_RING_PACK=1)
    const void* Wp = nullptr;
    int xcd_remap = 0;            // gemm_big: give each XCD a contiguous run of tiles
    int group_m = 0;              // gemm_big: tile order inside that run: columns of group_m row-tiles (0/1: row-major)
    int wide_epi = 0;             // gemm_big (set by its launcher): result tile through LDS, 16-byte row-contiguous stores
    // conv_halo, EPI_BIAS, tile as wide as the output (BN == N): the per-voxel RMS norm + modulation + SiLU that the
    // consumer would run as its own pass (LtxVideoResnetBlock3d norm2, vae.rs:779-801) applied to the bf16-rounded
    // result rows while they sit in LDS: y = silu(x * rsqrt(mean(x^2) + pn_eps) * (1 + scale[b][c]) + shift[b][c])
    int pn_on = 0; float pn_eps = 0.f; int pn_act = 0; int pn_mod_stride = 0;
    const float* pn_scale = nullptr; const float* pn_shift = nullptr;   // f32 [B, pn_mod_stride] or null
    // gemm_big tail split (set by its launcher): tiles [0, sk_full) whole, the rest cut into sk_sf K-ranges each
    int sk_full = 0, sk_sf = 1, sk_plain = 0;     // sk_plain: round 3's publish protocol (plain stores + agent release), A/B aid
    // Deferred reduction (gemm_ring.hip, linear layers of at most 512 rows): the K ranges of the shape rule leave their f32 sums in
    // defer_parts[part][m][n] (row-major, leading dimension N) and the launch ends there - no ticket, no epilogue; the consumer
    // (RowNormArgs::parts: the row norm that follows the layer) adds the ranges in part order and applies the epilogue's expression.
    float* defer_parts = nullptr;
    float* sk_ws = nullptr;       // f32 slabs [tail tile][part][BM*BN]
    unsigned* sk_cnt = nullptr;   // arrival counters [tail tile], zeroed before the launch
};
int ltx_launch_gemm(const GemmArgs& g, int dtype, int epi, hipStream_t s);
// canonical per-row partial sums of squares of x [rows, N] (row stride ld elements): out[row * ceil(N / 128) + g] (rownorm.hip)
int ltx_launch_rowsq(const void* x, int dtype, int64_t rows, int N, int ld, float* out, hipStream_t s);
void ltx_gemm_rowsq_done();      // a GEMM kernel that wrote GemmArgs::rowsq itself tells ltx_launch_gemm so (thread-local)
// large-tile LDS-DMA bf16 variant (gemm_big.hip); ltx_launch_gemm dispatches to it when eligible
bool ltx_gemm_big_eligible(const GemmArgs& g, int dtype);
int ltx_launch_gemm_big(const GemmArgs& g, int epi, hipStream_t s);
int ltx_gemm_split_factor(const GemmArgs& g);   // gemm_big.hip: K-ranges a small-output shape is cut into (shape only)
bool ltx_gemm_defer_ok(const GemmArgs& g, int epi);   // gemm_ring.hip: the call can leave its K-range sums to the consumer (GemmArgs::defer_parts)
// gemm_asm.hip: one-wave-per-SIMD kernels with a generated asm K loop; eligibility is a function of the shape only
bool ltx_gemm_asm_eligible(const GemmArgs& g, int dtype, int epi);
int ltx_launch_gemm_asm(const GemmArgs& g, int epi, hipStream_t s);
bool ltx_gemm_fold_ok(const GemmArgs& g, int epi);      // gemm_asm.hip: the call (with GemmArgs::C2 or ::rs_sq set) is one the wide epilogue serves
// cvec[b][n] = sum_k shift[b * shift_stride + k] * W[n][k] + bias[n]   (f32 accumulation in ascending k; W, bias bf16; rownorm.hip)
int ltx_launch_shift_gemv(const void* W, const void* bias, const float* shift, int shift_stride, int B, int N, int K, float* cvec, int cvec_stride, hipStream_t s);
bool ltx_gemm_asm16_fits(const GemmArgs& g, int epi);                       // the 16x16x32 one-wave-per-SIMD kernel (plan family asm16:*)
int ltx_launch_gemm_asm16(const GemmArgs& g, int epi, int tile, hipStream_t s);
bool ltx_gemm_asm16_conv_fits(const GemmArgs& g, int epi);                  // the same loop in conv mode (3x3x3, tile 256 x 256; plan "asm16c:256x256")
int ltx_launch_gemm_asm16_conv(const GemmArgs& g, int epi, hipStream_t s);
int ltx_gemm_asm_pick_tile(int M, int N);
const char* ltx_gemm_asm_tile_name(int i);
int ltx_gemm_big_pick_tile(int M, int N);   // index into gemm_big.hip's tile table
int ltx_launch_gemm_p8(const GemmArgs& g, int epi, int bn, hipStream_t s);
bool ltx_gemm_p8_fits(const GemmArgs& g);
// gemm_ring.hip: small-M linear layers on small tiles with a deep ring of LDS stages (plan family ring:*); same K partition and
// summation order as gemm_big (the split workspace below is gemm_big's: slabs + ticket counters per (device, stream))
bool ltx_gemm_ring_fits(const GemmArgs& g, int epi);
bool ltx_gemm_ring_tile_fits(const GemmArgs& g, int epi, int tile);      // conv mode runs on the tiles of at least 64 columns
int ltx_launch_gemm_ring(const GemmArgs& g, int epi, int tile, hipStream_t s);
int ltx_gemm_ring_tiles();
const char* ltx_gemm_ring_tile_name(int i);
int ltx_gemm_ring_tile_bm(int i);
int ltx_gemm_ring_tile_bn(int i);
int ltx_gemm_ring_pick_tile(const GemmArgs& g);
int ltx_gemm_split_workspace(GemmArgs* g, int tiles, int bm, int bn, hipStream_t s);
size_t ltx_ring_packed_bytes(int N, int K);
int ltx_pack_ring_weights(const void* W, int N, int K, void* out, hipStream_t s);      // bf16 [N, K] -> the GemmArgs::Wp layout   // sets sk_sf / sk_full = 0 / sk_ws / sk_cnt
bool ltx_gemm_big_fits(const GemmArgs& g);   // gemm_big.hip: every span its 32-bit buffer offsets address stays below 2 GiB
// conv_halo.hip: 3x3x3 conv with the activation patch + rim staged once per nine in-plane taps; bn = 128 / 256
bool ltx_conv_halo_eligible(const GemmArgs& g, int epi, int bn);
int ltx_launch_conv_halo(const GemmArgs& g, int epi, int bn, hipStream_t s);   // operands addressable with the kernel's 32-bit buffer offsets

// ---------------- row norms (rownorm.hip) ----------------
struct RowNormArgs {
    const void* x = nullptr; void* y = nullptr;
    int64_t rows = 0; int D = 0; int ldx = 0, ldy = 0;
    int kind = 0;                 // 0 = RMS (f32 stats), 1 = LayerNorm without affine
    float eps = 1e-6f;
    const void* weight = nullptr; // T [D] or null
    const float* scale = nullptr; // f32 [batch, mod_stride]: y = n*(1+scale)+shift ; null = no modulation
    const float* shift = nullptr;
    int64_t rows_per_batch = 1; int mod_stride = 0;
    int act = 0;                  // 0 none, 1 SiLU
    // RMS rows whose sum of squares is already known (GemmArgs::rowsq of the GEMM that produced x): presum[row * presum_n + g],
    // summed in ascending g.  The pass is then a pure elementwise map - no wave waits for a row (rownorm_presum_kernel).
    const float* presum = nullptr; int presum_n = 0;
    // Rows that arrive as the K-range sums of the linear layer before them (GemmArgs::defer_parts: parts[p][row][D] f32): the row is
    // first FINISHED the way that layer's gate / residual epilogue would have - h = resid + gate * (((p0 + p1) + ...) + bias),
    // rounded to T, resid = x (read), h written back to x_out (may alias x) - and then normalised as usual.  Rows held whole in
    // registers only (one wave per row, or - at most 512 rows - one block per row); gate f32 [batch, gate_stride] or null (then h = resid + sum + bias).
    const float* parts = nullptr; int nparts = 0; int64_t part_stride = 0;
    const void* d_bias = nullptr; const float* d_gate = nullptr; int d_gate_stride = 0; void* x_out = nullptr;
};
int ltx_launch_rownorm(const RowNormArgs& a, int dtype, hipStream_t s);

struct QkNormRopeArgs {
    void* x = nullptr;            // in place, nseg segments of width D at x + j*D in each row
    int64_t rows = 0; int D = 0; int ld = 0; int nseg = 1;
    int64_t seg_stride = 0;       // elements from segment j to j+1 (0: D, i.e. adjacent column blocks of one row)
    const void* w0 = nullptr; const void* w1 = nullptr;   // T [D]
    const void* w0b = nullptr;    // optional second weight on segment 0: x * rinv * w0 * w0b (a consumer's weight folded in, dit.hip)
    float eps = 1e-5f;
    const float* cos = nullptr; const float* sin = nullptr;  // f32 [rows, D/2] or null (no RoPE)
    float out_scale0 = 1.f;       // extra factor on segment 0's output (q): lets attention fold scale*log2(e) into Q
};
int ltx_launch_qknorm_rope(const QkNormRopeArgs& a, int dtype, hipStream_t s);

struct RopeTableArgs {
    float* cos = nullptr; float* sin = nullptr;   // [B*S, D/2]
    const float* coords = nullptr;                // [B*S, 3] f32 (already pixel/second units) or null
    const float* freqs = nullptr;                 // device f32 [D/6] = theta^linspace * pi/2
    int B = 1, F = 1, H = 1, W = 1, D = 0;
    int use_coords = 0;
    float gscale[3] = {1.f, 1.f, 1.f};            // multiplies coords (1/base) or the raw grid
};
int ltx_launch_rope_table(const RopeTableArgs& a, hipStream_t s);

// ---------------- attention (attention.hip) ----------------
struct AttnArgs {
    const void* q = nullptr; const void* k = nullptr; const void* v = nullptr; void* o = nullptr;
    int ldq = 0, ldk = 0, ldv = 0, ldo = 0;      // row strides in elements
    int B = 1, Sq = 0, Sk = 0, heads = 0, hd = 0;
    float scale = 1.f;
    const float* bias = nullptr;                  // f32 [B, Sk] additive key bias or null
    // f32 [heads, Sq, Sk] additive bias shared by the batch (T5's relative position bias; attn_cross64_kernel only: head_dim 64,
    // Sk <= 128, Sk % 4 == 0)
    const float* bias2d = nullptr;
    int q_prescaled = 0;                          // bf16, no bias: q already carries scale*log2(e) (qknorm_rope out_scale0)
    int xcd_heads = 0;                            // set by the launcher: whole heads per XCD (block order, speed only)
    int wide_o = 0;                               // set by the launcher: 16-byte output stores (ldo % 8 == 0, 16-byte aligned o)
    // Queries that arrive UN-normalised (attn_cross64_kernel only): score row i is multiplied by
    // 1 / sqrt(sum_g q_rowsq[i * q_rowsq_n + g] / q_rowsq_D + q_rowsq_eps), the RMS-norm scalar of query row i (GemmArgs::rowsq
    // of the projection that produced q); the norm's weight vector is folded into k by the caller.
    const float* q_rowsq = nullptr; int q_rowsq_n = 0, q_rowsq_D = 0; float q_rowsq_eps = 0.f;
    // Valid keys per batch row (device int [B], attn_cross64_kernel only): batch row b attends to the FIRST k_count[b] rows of its
    // K / V / bias (row stride between batch rows stays Sk); rows from k_count[b] up to the next multiple of 32 must be finite.
    // The caller compacts the keys an additive mask leaves alive (ltx_launch_key_compact / ltx_launch_gather_rows): a key whose
    // bias is <= -5000 contributes exp(s - 5000 - max) = +0.0f exactly in f32 as long as scores stay within +-4000 of each other,
    // so dropping it changes nothing (ltx_transformer.rs:1059-1070 builds -10000 for masked text tokens).
    const int* k_count = nullptr;
    const int* gate_flag = nullptr; int gate_ticket = 0;   // attn_bf16_kernel<128>: run only if *gate_flag == gate_ticket (exact pass after attn_q128's overflow flag)
};
bool ltx_attention_q128_fits(const AttnArgs& a);           // attn_q128.hip: head_dim 128 one-wave-per-SIMD kernel
int ltx_launch_attention_q128(const AttnArgs& a, hipStream_t s, int** flag_out, int* ticket_out);
int ltx_q64_fallback_read(unsigned long long* out, int reset);      // attn_q64.hip / attention.hip: diagnostic counters of the exact-max second passes
int ltx_q128_fallback_read(unsigned long long* out, int reset);
int ltx_attention_q128_prepare();   // allocate + zero the current device's overflow flags outside any launch path (create / warm-up)
int ltx_launch_attention(const AttnArgs& a, int dtype, hipStream_t s);
int ltx_launch_attention_q64(const AttnArgs& a, hipStream_t s);   // attn_q64.hip: head_dim 64, q prescaled, 64 queries per wave (caller sets xcd_heads / wide_o)
bool ltx_attention_q64_fits(const AttnArgs& a);   // attn_q64.hip: every row offset below 2^31 (its 32-bit buffer arithmetic)
bool ltx_attention_prescale_ok(int hd);
bool ltx_attention_cross64_ok(int hd, int Sk);   // shape-only: the short-key-set kernel serves the launch (k_count / bias2d / q_rowsq)
bool ltx_attention_rowsq_ok(int hd, int Sk, int D);   // shape-only: cross attention can fold the q RMS-norm (AttnArgs::q_rowsq)           // whether the bf16 kernel has a q-prescaled instantiation for this head dim

// ---------------- small elementwise kernels (elementwise.hip) ----------------
constexpr int LTX_MAX_BATCH = 16;      // per-sample scalars a launch carries by value (API batches run as chunks of 8; tile batches use 16)
struct TimeVec { float t[LTX_MAX_BATCH]; int n; };
// out[b][0:half] = cos(t_b*tab), out[b][half:] = sin(t_b*tab); t rounded to T first when round_t
// ggml blocks (device) -> dense tensor (gguf_dequant.hip)
int ltx_launch_gguf_dequant(const void* blocks_dev, int ggml_type, int64_t numel, void* dst, int dst_dtype, hipStream_t s);
int ltx_launch_sinusoid(void* out, int dtype, const TimeVec& tv, const float* tab, int half, int round_t, float tmul, hipStream_t s);
int ltx_launch_silu(const void* x, void* y, int64_t n, int dtype, hipStream_t s);
// y = x + noise[hw] * scale[c] (+ resid), channels-last [rows = B T HW, C], the roundings of vae.rs:741-753 in the model dtype
int ltx_launch_noise_inject(const void* x, void* y, const float* noise, const void* scale, const void* resid, int64_t rows, int C, int64_t HW, int dtype, hipStream_t s);
int ltx_launch_cast(const void* x, int xdt, void* y, int ydt, int64_t n, hipStream_t s);
// ada[l][b][j] = table_l[j] + temb[b][j]  (j < width), tables given as nl pointers packed contiguously [nl][width] (T), out f32
int ltx_launch_ada(float* out, const void* tables, const void* temb, int nl, int B, int width, int dtype, hipStream_t s);
int ltx_launch_mask_bias(float* out, const float* mask, int64_t n, hipStream_t s);
// Keys an additive key bias leaves alive, per batch row: idx[b][0 .. count[b]) = the keys with bias > -5000 in ascending order
// (all K keys when none is: a constant shift of every score, softmax unchanged), bias_c[b][pos] = their biases (-inf past
// count[b]).  bias [B, K] f32 (ltx_launch_mask_bias); idx int [B, K], count int [B].
int ltx_launch_key_compact(const float* bias, int B, int K, int* idx, int* count, float* bias_c, hipStream_t s);
// dst[l][b][pos] = src[l][b][idx[b][pos]] for pos < count[b], zeros past it; rows of row_bytes (a multiple of 16) bytes,
// nl x B x K rows each side
int ltx_launch_gather_rows(const void* src, void* dst, const int* idx, const int* count, int nl, int B, int K, int row_bytes, hipStream_t s);
// h = h*(1-m_b) + orig*m_b
int ltx_launch_skip_blend(void* h, const void* orig, const TimeVec& m, int64_t rows_per_batch, int D, int dtype, hipStream_t s);
int ltx_launch_scale_cols(const void* W, const float* scale, void* out, int64_t N, int K, int dtype, hipStream_t s);      // out[n][k] = W[n][k] * (1 + scale[k])
int ltx_launch_mod_scale(const void* h, const float* scale, int scale_stride, void* y, int B, int64_t rows_per_batch, int D, int dtype, hipStream_t s);   // y = h (.) (1 + scale[b])

struct GuidanceArgs {
    const void* text = nullptr; const void* uncond = nullptr; const void* pert = nullptr;  // model dtype (pred_dtype)
    int pred_dtype = LTX_DT_F32;
    float* latents = nullptr;      // f32 [B, n_per_batch], updated in place: x += dt * noise_pred
    float* noise_out = nullptr;    // optional f32 copy of the combined prediction
    int B = 1; int64_t n_per_batch = 0;
    float guidance_scale = 1.f, guidance_rescale = 0.f, stg_scale = 0.f, dt = 0.f;
    double* stats = nullptr;       // workspace [B][4] doubles (sum_t, sumsq_t, sum_c, sumsq_c)
    // stochastic sampling (scheduler.rs:557-575): x = (1 - sigma_next) * (x - sigma * v) + sigma_next * noise
    const float* step_noise = nullptr;   // f32 [B, n_per_batch] or null (= deterministic Euler with dt)
    float sigma = 0.f, sigma_next = 0.f;
};
int ltx_launch_guidance_step(const GuidanceArgs& a, hipStream_t s);

// tokens [B,S,C] f32 -> channels-last T:  y = (x*std/sf + mean)*(1-ns) + noise*ns ; noise is NCTHW f32 [B,C,S] or null
int ltx_launch_denorm_mix(const float* lat, const float* mean, const float* std_, float inv_sf, const float* noise,
                          const TimeVec& nscale, void* out, int dtype, int B, int64_t S, int C, hipStream_t s);
// NCTHW (src dtype) -> channels-last T
int ltx_launch_ncthw_to_cl(const void* x, int xdt, void* y, int ydt, int B, int C, int64_t S, hipStream_t s);
// tiled-decode blends on f32 NCTHW tiles (vae.rs:1927-2006); dim: 2=T,3=H,4=W
struct BlendArgs {
    const float* a = nullptr; const float* b = nullptr; float* dst = nullptr;   // dst may alias b
    int BC = 1;                       // B*C
    int at = 1, ah = 1, aw = 1;       // physical dims of a
    int a_len = 1;                    // logical length of a along `dim` (its tail is blended)
    int bt = 1, bh = 1, bw = 1;       // physical dims of b
    int dt = 1, dh = 1, dw = 1;       // physical dims of dst
    int ot = 0, oh = 0, ow = 0;       // where b's origin sits inside dst
    int et = 1, eh = 1, ew = 1;       // extent of b to process (along `dim`: <= blend)
    int dim = 3, blend = 0;
    float inv_blend = 0.f;            // 1/blend, filled by ltx_launch_blend
};
int ltx_launch_blend(const BlendArgs& a, hipStream_t s);
// copy a [BC, st, sh, sw] window of src (dims [BC, t,h,w]) into dst at offset (ot,oh,ow) of dims [BC, T,H,W]
int ltx_launch_copy_window(const float* src, int t, int h, int w, float* dst, int T, int H, int W, int BC,
                           int st, int sh, int sw, int ot, int oh, int ow, hipStream_t s);
int ltx_launch_postprocess(float* x, int64_t n, hipStream_t s);
