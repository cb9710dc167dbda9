/* ltxhip_ops.h — kernel-level C entry points of libltxhip.so.
 *
 * These expose the individual fused HIP kernels behind `ltx_dit_forward` / `ltx_vae_decode`
 * so that parity tests and micro-benchmarks can drive each one through the C ABI with plain
 * device pointers.  Each entry cites the reference code the kernel replaces
 * (FerrisMind/candle-video, src/models/ltx_video/...).  dtype: 0 = f32, 1 = bf16 (ltx_dtype).
 */
#ifndef LTXHIP_OPS_H
#define LTXHIP_OPS_H
#include "ltxhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* nn::Linear (+ fused epilogue): y[M,N] = epi(x[M,K] @ w[N,K]^T + bias)
 *   epi 0: none | 1: GELU-tanh (ltx_transformer.rs:214-226) | 2: resid + gate[b,:]*y (:900,:934)
 *   | 3: resid + y (:909).  gate f32 [M/rows_per_batch, N]. */
int ltx_op_linear(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int dtype, int epi,
                  const void* resid, const float* gate, int rows_per_batch, ltx_stream stream);

/* nn::Linear (epi 0, or the residual forms 2 / 3) that also leaves the per-row partial sums of squares of its stored output, one f32 per 128-column
 * group: rowsq[m * ceil(N/128) + g].  The summation order is canonical (independent of the kernel the plan picks), so a
 * consumer can fold an RMS norm of the output rows without a pass over them: the cross-attention q-norm,
 * ltx_transformer.rs:671-678.  ltx_op_rowsq is the stand-alone form on a stored matrix: same values, bit for bit. */
/* bf16 linear layers of at most 512 rows (csrc/gemm_ring.hip): a second, tile-contiguous copy of w ([ceil(N/32)][ceil(K/64)][32][64],
 * zero padded; ltx_op_ring_packed_bytes bytes) lets the small-M kernel stream it in 4-KiB blocks.  Same results as ltx_op_linear;
 * measured to buy nothing for C1 and 4 % for T5-XXL, so the models build these copies only under LTX_RING_PACK=1. */
int64_t ltx_op_ring_packed_bytes(int N, int K);
int ltx_op_ring_pack(const void* w, int N, int K, void* out, ltx_stream stream);
int ltx_op_linear_packed(const void* x, const void* w, const void* w_packed, const void* bias, void* y, int M, int N, int K, int epi,
                         const void* resid, const float* gate, int rows_per_batch, ltx_stream stream);
int ltx_op_linear_rowsq(const void* x, const void* w, const void* bias, void* y, float* rowsq, int M, int N, int K, int dtype, int epi,
                        const void* resid, const float* gate, int rows_per_batch, ltx_stream stream);   /* epi 0, 2, 3 as ltx_op_linear */
int ltx_op_rowsq(const void* x, int64_t rows, int N, int ld, float* rowsq, int dtype, ltx_stream stream);

/* The fused q|k|v projection of LtxAttention (ltx_transformer.rs:655-662: to_q, to_k, to_v on the same input) with the
 * output written as N/seg_width DENSE matrices: y[j][M][seg_width] = (x @ w^T + bias)[:, j*seg_width:(j+1)*seg_width].
 * seg_width a power of two dividing N. */
int ltx_op_linear_segmented(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int seg_width,
                            int dtype, ltx_stream stream);

/* The RMS form of ltx_op_rownorm for rows whose sums of squares are already known (presum [rows, presum_n], the by-product of
 * ltx_op_linear_rowsq / ltx_op_rowsq on x): a pure elementwise map, no row reduction. */
int ltx_op_rownorm_presum(const void* x, void* y, int64_t rows, int D, float eps, const void* weight,
                          const float* scale, const float* shift, int64_t rows_per_batch, int mod_stride, int act,
                          const float* presum, int presum_n, int dtype, ltx_stream stream);

/* A linear layer of at most 512 rows whose K ranges (the shape rule of the library: ltx_op_linear_split_factor(M, N, K), four from
 * K = 8192 up) are left UN-reduced: parts[p][M][N] f32 = x @ w[:, range p]^T, no bias, no epilogue (gemm_ring.hip; the DiT's ff2
 * at few tokens, LtxVideoTransformerBlock::forward ltx_transformer.rs:929-934).  ltx_op_rownorm_deferred finishes the rows:
 * h = resid + gate_b * (((p0 + p1) + ...) + bias) (gate NULL: resid + sum + bias), rounded to the dtype, written to h_out, then
 * y = the row norm of ltx_op_rownorm on h.  Together they return the bits of ltx_op_linear(epi 2 / 3) followed by ltx_op_rownorm. */
int ltx_op_linear_split_factor(int M, int N, int K);
int ltx_op_linear_deferred(const void* x, const void* w, float* parts, int M, int N, int K, ltx_stream stream);
int ltx_op_rownorm_deferred(const float* parts, int nparts, const void* bias, const void* resid, const float* gate, int gate_stride, void* h_out,
                            void* y, int64_t rows, int D, int kind, float eps, const float* scale, const float* shift, int64_t rows_per_batch,
                            int mod_stride, int dtype, ltx_stream stream);

/* RmsNorm / LayerNormNoParams + AdaLN modulate (+SiLU) on rows (ltx_transformer.rs:72-119, 874-889;
 * vae.rs:148-153, 711-739): y = act(norm(x)[*weight]*(1+scale_b)+shift_b). kind 0 RMS / 1 LN. */
int ltx_op_rownorm(const void* x, void* y, int64_t rows, int D, int kind, float eps, const void* weight,
                   const float* scale, const float* shift, int64_t rows_per_batch, int mod_stride, int act,
                   int dtype, ltx_stream stream);

/* q/k RMSNorm(weight, eps) over the full inner dim + apply_rotary_emb, in place (ltx_transformer.rs:671-678, 314-339).
 * cos/sin: f32 [rows, D/2] half-width tables or NULL. */
int ltx_op_qknorm_rope(void* x, int64_t rows, int D, int ld, const void* weight, float eps,
                       const float* cos, const float* sin, int dtype, ltx_stream stream);

/* get_timestep_embedding, dim 256, order [cos | sin]: flavour 0 = the DiT's (ltx_transformer.rs:271-309, frequencies
 * 1/10000^(i/128)), 1 = the VAE's (vae.rs:172-198, exp(-ln(1e4) i / 128), timestep first multiplied by `multiplier` =
 * timestep_scale_multiplier, :1668-1676).  In bf16 the timestep is rounded to bf16 first (:1051).  timesteps: HOST [n], n <= 8;
 * out DEVICE [n,256] of `dtype`.  Blocks until done (it stages its 128-entry table itself). */
int ltx_op_timestep_embedding(const float* timesteps_host, int n, int vae_flavour, float multiplier, int dtype, void* out, ltx_stream stream);

/* LtxVideoRotaryPosEmbed::forward (ltx_transformer.rs:436-524): half-width tables [B*F*H*W, D/2];
 * coords f32 [B*S,3] or NULL (then the (f,h,w) grid scaled by rope_scale*patch/base or raw). */
int ltx_op_rope_table(float* cos, float* sin, const float* coords, int B, int F, int H, int W, int D,
                      const float* rope_scale_host, ltx_stream stream);

/* LtxAttention core (ltx_transformer.rs:699-741): o = softmax(scale q k^T + bias) v, q [B,Sq,heads*hd] etc. */
int ltx_op_attention(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                     int ldq, int ldk, int ldv, int ldo, float scale, const float* key_bias, int dtype, ltx_stream stream);
/* Diagnostic counters of the self-attention kernels' exact-max second pass (their fixed first-tile softmax max overflowed; the
 * result is exact either way, the pass costs time): counts[0] = head_dim-64 workgroups that re-ran since the last reset,
 * counts[1] = head_dim-128 launches whose gated exact pass ran.  Current device; blocks until the device copy is done. */
int ltx_attention_fallback_counts(unsigned long long counts[2], int reset);
/* Cross attention on UN-normalised queries (bf16, head_dim 64, Sk <= 128): softmax(r_i * scale * q_i . k_j + bias_j) v_j with
 * r_i = 1 / sqrt(sum_g q_rowsq[i*n + g] / D + eps), the RMS-norm scalar of query row i; the norm's weight is expected folded
 * into k by the caller (LtxAttention::forward with norm_q, ltx_transformer.rs:671-678, 719-740). */
int ltx_op_attention_rowsq(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                           int ldq, int ldk, int ldv, int ldo, float scale, const float* key_bias,
                           const float* q_rowsq, int q_rowsq_n, int q_rowsq_D, float q_rowsq_eps, ltx_stream stream);
/* ltx_op_attention for bf16, head_dim 64, Sk <= 128 with a key bias, run the way the DiT's cross attention runs it (dit.hip): the
 * keys whose bias is above -5000 are moved to the front of their batch row (order kept) and the kernel multiplies only the key
 * blocks that hold them - a masked text token (bias -10000, ltx_transformer.rs:1059-1070) has softmax weight exp(s - 10000 - max)
 * = +0.0f exactly, so the result is that of ltx_op_attention.  q_rowsq may be NULL (then q is used as it is).  counts_out
 * (DEVICE int [B], may be NULL) receives the number of keys kept per batch row.  Blocks until done (temporary buffers). */
int ltx_op_attention_compact(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                             int ldq, int ldk, int ldv, int ldo, float scale, const float* key_bias,
                             const float* q_rowsq, int q_rowsq_n, int q_rowsq_D, float q_rowsq_eps, int* counts_out, ltx_stream stream);
/* Same core for bf16, head_dim 64 or 128, no key bias, with q ALREADY multiplied by scale*log2(e) (the DiT self-attention
 * path folds that factor into the q RMSNorm+RoPE kernel): o = softmax_base2(q' k^T) v. */
int ltx_op_attention_prescaled(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                               int ldq, int ldk, int ldv, int ldo, ltx_stream stream);

/* LtxVideoCausalConv3d 3x3x3 (vae.rs:415-464) on channels-last x [B,T,H,W,Cin] with reference-layout weight
 * [Cout,Cin,3,3,3] (any dtype `wdtype`); y channels-last [B,T,H,W,Cout]; resid optional [.., Cout]. */
int ltx_op_conv3d(const void* x, const void* w, const void* bias, int wdtype, void* y, const void* resid,
                  int B, int T, int H, int W, int Cin, int Cout, int causal, int dtype, ltx_stream stream);

/* LtxVideoUpsampler3d (vae.rs:1090-1169): conv + depth-to-space(2,2,2) + drop first frame + tiled residual.
 * x [B,T,H,W,Cin] -> y [B,2T-1,2H,2W,Cout/8] channels-last. */
int ltx_op_upsample3d(const void* x, const void* w, const void* bias, int wdtype, void* y,
                      int B, int T, int H, int W, int Cin, int Cout, int causal, int residual, int dtype, ltx_stream stream);

/* conv_out + unpatchify(4) (vae.rs:1626-1654, 1724-1725): x [B,T,H,W,Cin] -> f32 NCTHW [B,Cout/16,T,4H,4W]. */
int ltx_op_conv_out_unpatchify(const void* x, const void* w, const void* bias, int wdtype, float* y,
                               int B, int T, int H, int W, int Cin, int Cout, int causal, int postprocess, int dtype, ltx_stream stream);

/* blend_h / blend_v / blend_t of the tiled decode (vae.rs:1927-2006), in place on b:
 * b[..., x] = a[..., -blend + x]*(1 - x/blend) + b[..., x]*(x/blend) along dim (2=T, 3=H, 4=W); a,b f32 [BC,t,h,w]. */
int ltx_op_blend(const float* a, float* b, int BC, int at, int ah, int aw, int bt, int bh, int bw, int dim, int blend_extent, ltx_stream stream);

/* Diagnostic: which GEMM plan (tile shape / kernel) the dispatcher measured best and cached for a bf16 problem shape
 * (conv = 0: [M,K] x [N,K]^T; conv = 1: K = Cin, ntaps/T/H/W = conv geometry).  Writes a NUL-terminated name
 * ("192x128", "p8:256", ... or "" if the shape has not run yet) into name[cap]. */
int ltx_op_gemm_plan(int M, int N, int K, int conv, int ntaps, int T, int H, int W, char* name, int cap);

#ifdef __cplusplus
}
#endif
#endif
