// Kernel-level C entry points (include/ltxhip_ops.h) — thin argument marshalling only.
#include "model_util.h"
#include "../../include/ltxhip_ops.h"

static inline int dtc(int d) { return d == 1 ? LTX_DT_BF16 : LTX_DT_F32; }

extern "C" int ltx_op_linear(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int dtype, int epi,
                             const void* resid, const float* gate, int rows_per_batch, ltx_stream stream) {
    if (!x || !w || !y) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear: null tensor");
    if (epi < 0 || epi > 3) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear: epi must be 0..3");
    if ((epi == 2 || epi == 3) && !resid) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear: residual epilogue needs resid");
    if (epi == 2 && (!gate || rows_per_batch < 1)) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear: gated epilogue needs gate");
    GemmArgs g; g.A = x; g.W = w; g.C = y; g.bias = bias; g.resid = resid; g.gate = gate;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; g.gate_stride = N;
    return ltx_launch_gemm(g, dtc(dtype), epi, (hipStream_t)stream);
}

extern "C" int64_t ltx_op_ring_packed_bytes(int N, int K) { return (int64_t)ltx_ring_packed_bytes(N, K); }
extern "C" int ltx_op_ring_pack(const void* w, int N, int K, void* out, ltx_stream stream) { return ltx_pack_ring_weights(w, N, K, out, (hipStream_t)stream); }
extern "C" int ltx_op_linear_packed(const void* x, const void* w, const void* w_packed, const void* bias, void* y, int M, int N, int K, int epi,
                                    const void* resid, const float* gate, int rows_per_batch, ltx_stream stream) {
    if (!x || !w || !w_packed || !y) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_packed: null tensor");
    if (epi < 0 || epi > 3) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_packed: epi must be 0..3");
    if ((epi == 2 || epi == 3) && !resid) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_packed: residual epilogue needs resid");
    if (epi == 2 && (!gate || rows_per_batch < 1)) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_packed: gated epilogue needs gate");
    GemmArgs g; g.A = x; g.W = w; g.Wp = w_packed; g.C = y; g.bias = bias; g.resid = resid; g.gate = gate;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; g.gate_stride = N;
    return ltx_launch_gemm(g, LTX_DT_BF16, epi, (hipStream_t)stream);
}

extern "C" int ltx_op_linear_rowsq(const void* x, const void* w, const void* bias, void* y, float* rowsq, int M, int N, int K, int dtype, int epi,
                                   const void* resid, const float* gate, int rows_per_batch, ltx_stream stream) {
    if (!x || !w || !y || !rowsq) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_rowsq: null tensor");
    if (epi != 0 && epi != 2 && epi != 3) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_rowsq: epi must be 0, 2 or 3");
    if (epi >= 2 && !resid) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_rowsq: residual epilogue needs resid");
    if (epi == 2 && (!gate || rows_per_batch < 1)) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_rowsq: gated epilogue needs gate");
    GemmArgs g; g.A = x; g.W = w; g.C = y; g.bias = bias; g.resid = resid; g.gate = gate; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N;
    g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; g.gate_stride = N; g.rowsq = rowsq;
    return ltx_launch_gemm(g, dtc(dtype), epi, (hipStream_t)stream);
}
extern "C" int ltx_op_rowsq(const void* x, int64_t rows, int N, int ld, float* rowsq, int dtype, ltx_stream stream) {
    return ltx_launch_rowsq(x, dtc(dtype), rows, N, ld, rowsq, (hipStream_t)stream);
}

extern "C" int ltx_op_linear_segmented(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int seg_width,
                                       int dtype, ltx_stream stream) {
    if (!x || !w || !y) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_segmented: null tensor");
    if (seg_width <= 0 || (seg_width & (seg_width - 1)) || N % seg_width) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_segmented: seg_width must be a power of two dividing N");
    GemmArgs g; g.A = x; g.W = w; g.C = y; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = seg_width;
    g.c_seg_shift = __builtin_ctz((unsigned)seg_width); g.c_seg_stride = (int64_t)M * seg_width;
    return ltx_launch_gemm(g, dtc(dtype), EPI_BIAS, (hipStream_t)stream);
}

extern "C" int ltx_op_rownorm(const void* x, void* y, int64_t rows, int D, int kind, float eps, const void* weight,
                              const float* scale, const float* shift, int64_t rows_per_batch, int mod_stride, int act,
                              int dtype, ltx_stream stream) {
    if (!x || !y) LTX_FAIL(LTX_ERR_ARG, "ltx_op_rownorm: null tensor");
    RowNormArgs a; a.x = x; a.y = y; a.rows = rows; a.D = D; a.ldx = D; a.ldy = D; a.kind = kind; a.eps = eps; a.weight = weight;
    a.scale = scale; a.shift = shift; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; a.mod_stride = mod_stride; a.act = act;
    return ltx_launch_rownorm(a, dtc(dtype), (hipStream_t)stream);
}

extern "C" int ltx_attention_fallback_counts(unsigned long long counts[2], int reset) {
    if (!counts) LTX_FAIL(LTX_ERR_ARG, "ltx_attention_fallback_counts: null output");
    LTX_TRY(ltx_q64_fallback_read(&counts[0], reset));
    return ltx_q128_fallback_read(&counts[1], reset);
}

extern "C" int ltx_op_linear_split_factor(int M, int N, int K) {
    GemmArgs g; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    return ltx_gemm_split_factor(g);
}
extern "C" int ltx_op_linear_deferred(const void* x, const void* w, float* parts, int M, int N, int K, ltx_stream stream) {
    if (!x || !w || !parts) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_deferred: null tensor");
    GemmArgs g; g.A = x; g.W = w; g.C = parts; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.defer_parts = parts;
    if (!ltx_gemm_defer_ok(g, EPI_GATE_RESID)) LTX_FAIL(LTX_ERR_ARG, "ltx_op_linear_deferred: bf16 linear layers of at most 512 rows, N % 8 == 0, K % 8 == 0");
    return ltx_launch_gemm(g, LTX_DT_BF16, EPI_BIAS, (hipStream_t)stream);
}
extern "C" int ltx_op_rownorm_deferred(const float* parts, int nparts, const void* bias, const void* resid, const float* gate, int gate_stride, void* h_out,
                                       void* y, int64_t rows, int D, int kind, float eps, const float* scale, const float* shift, int64_t rows_per_batch,
                                       int mod_stride, int dtype, ltx_stream stream) {
    if (!parts || !resid || !h_out || !y) LTX_FAIL(LTX_ERR_ARG, "ltx_op_rownorm_deferred: null tensor");
    RowNormArgs a; a.x = resid; a.y = y; a.rows = rows; a.D = D; a.ldx = D; a.ldy = D; a.kind = kind; a.eps = eps;
    a.scale = scale; a.shift = shift; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; a.mod_stride = mod_stride;
    a.parts = parts; a.nparts = nparts; a.part_stride = rows * D; a.d_bias = bias; a.d_gate = gate; a.d_gate_stride = gate_stride; a.x_out = h_out;
    return ltx_launch_rownorm(a, dtc(dtype), (hipStream_t)stream);
}

extern "C" int ltx_op_rownorm_presum(const void* x, void* y, int64_t rows, int D, float eps, const void* weight,
                                     const float* scale, const float* shift, int64_t rows_per_batch, int mod_stride, int act,
                                     const float* presum, int presum_n, int dtype, ltx_stream stream) {
    if (!x || !y || !presum) LTX_FAIL(LTX_ERR_ARG, "ltx_op_rownorm_presum: null tensor");
    RowNormArgs a; a.x = x; a.y = y; a.rows = rows; a.D = D; a.ldx = D; a.ldy = D; a.kind = 0; a.eps = eps; a.weight = weight;
    a.scale = scale; a.shift = shift; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; a.mod_stride = mod_stride; a.act = act;
    a.presum = presum; a.presum_n = presum_n;
    return ltx_launch_rownorm(a, dtc(dtype), (hipStream_t)stream);
}

extern "C" int ltx_op_timestep_embedding(const float* timesteps_host, int n, int vae_flavour, float multiplier, int dtype, void* out, ltx_stream stream) {
    if (!timesteps_host || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_op_timestep_embedding: null argument");
    if (n < 1 || n > 8) LTX_FAIL(LTX_ERR_ARG, "ltx_op_timestep_embedding: batch must be 1..8");
    float tab[128]; ltx_sinusoid_table(vae_flavour, tab);
    float* dtab = nullptr;
    HIP_TRY(hipMalloc((void**)&dtab, sizeof(tab)));
    int rc = LTX_OK;
    if (hipMemcpy(dtab, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) rc = LTX_ERR_HIP;
    if (rc == LTX_OK) {
        TimeVec tv; tv.n = n;
        for (int i = 0; i < n; ++i) tv.t[i] = timesteps_host[i];
        rc = ltx_launch_sinusoid(out, dtc(dtype), tv, dtab, 128, dtc(dtype) == LTX_DT_BF16, multiplier, (hipStream_t)stream);
        if (rc == LTX_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = LTX_ERR_HIP;
    }
    (void)hipFree(dtab);
    return rc;
}

extern "C" int ltx_op_qknorm_rope(void* x, int64_t rows, int D, int ld, const void* weight, float eps,
                                  const float* cos, const float* sin, int dtype, ltx_stream stream) {
    if (!x || !weight) LTX_FAIL(LTX_ERR_ARG, "ltx_op_qknorm_rope: null tensor");
    QkNormRopeArgs a; a.x = x; a.rows = rows; a.D = D; a.ld = ld; a.nseg = 1; a.w0 = weight; a.eps = eps; a.cos = cos; a.sin = sin;
    return ltx_launch_qknorm_rope(a, dtc(dtype), (hipStream_t)stream);
}

extern "C" int ltx_op_rope_table(float* cos, float* sin, const float* coords, int B, int F, int H, int W, int D,
                                 const float* rope_scale_host, ltx_stream stream) {
    if (!cos || !sin) LTX_FAIL(LTX_ERR_ARG, "ltx_op_rope_table: null tensor");
    int steps = D / 6; if (steps < 1) steps = 1;
    std::vector<float> fr(steps);
    const float theta_ln = (float)std::log(10000.0);
    for (int i = 0; i < steps; ++i) {
        float lin = steps <= 1 ? 0.0f : (float)i * (float)(1.0 / (double)(steps - 1));
        fr[i] = (float)std::exp((double)(lin * theta_ln)) * (float)(M_PI / 2.0);
    }
    float* dfr = nullptr;
    HIP_TRY(hipMalloc((void**)&dfr, sizeof(float) * steps));
    HIP_TRY(hipMemcpy(dfr, fr.data(), sizeof(float) * steps, hipMemcpyHostToDevice));
    RopeTableArgs r; r.cos = cos; r.sin = sin; r.freqs = dfr; r.B = B; r.D = D;
    if (coords) { r.use_coords = 1; r.coords = coords; r.F = 1; r.H = 1; r.W = F * H * W;
                  r.gscale[0] = (float)(1.0 / 20.0); r.gscale[1] = (float)(1.0 / 2048.0); r.gscale[2] = (float)(1.0 / 2048.0); }
    else { r.F = F; r.H = H; r.W = W;
           if (rope_scale_host) { r.gscale[0] = (float)((double)rope_scale_host[0] / 20.0); r.gscale[1] = (float)((double)rope_scale_host[1] / 2048.0); r.gscale[2] = (float)((double)rope_scale_host[2] / 2048.0); } }
    int rc = ltx_launch_rope_table(r, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(dfr);
    if (rc != LTX_OK) return rc;
    if (e != hipSuccess) { ltx_set_error(hipGetErrorString(e)); return LTX_ERR_HIP; }
    return LTX_OK;
}

extern "C" int ltx_op_attention(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                                int ldq, int ldk, int ldv, int ldo, float scale, const float* key_bias, int dtype, ltx_stream stream) {
    if (!q || !k || !v || !o) LTX_FAIL(LTX_ERR_ARG, "ltx_op_attention: null tensor");
    AttnArgs a; a.q = q; a.k = k; a.v = v; a.o = o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.Sq = Sq; a.Sk = Sk; a.heads = heads; a.hd = hd; a.scale = scale; a.bias = key_bias;
    return ltx_launch_attention(a, dtc(dtype), (hipStream_t)stream);
}

extern "C" int ltx_op_attention_rowsq(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                                      int ldq, int ldk, int ldv, int ldo, float scale, const float* key_bias,
                                      const float* q_rowsq, int q_rowsq_n, int q_rowsq_D, float q_rowsq_eps, ltx_stream stream) {
    if (!q || !k || !v || !o || !q_rowsq) LTX_FAIL(LTX_ERR_ARG, "ltx_op_attention_rowsq: null tensor");
    AttnArgs a; a.q = q; a.k = k; a.v = v; a.o = o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.Sq = Sq; a.Sk = Sk; a.heads = heads; a.hd = hd; a.scale = scale; a.bias = key_bias;
    a.q_rowsq = q_rowsq; a.q_rowsq_n = q_rowsq_n; a.q_rowsq_D = q_rowsq_D; a.q_rowsq_eps = q_rowsq_eps;
    return ltx_launch_attention(a, LTX_DT_BF16, (hipStream_t)stream);
}

extern "C" int ltx_op_attention_compact(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                                        int ldq, int ldk, int ldv, int ldo, float scale, const float* key_bias,
                                        const float* q_rowsq, int q_rowsq_n, int q_rowsq_D, float q_rowsq_eps, int* counts_out, ltx_stream stream) {
    if (!q || !k || !v || !o || !key_bias) LTX_FAIL(LTX_ERR_ARG, "ltx_op_attention_compact: null tensor");
    if (!ltx_attention_cross64_ok(hd, Sk) || B < 1 || ldk != heads * hd || ldv != heads * hd) LTX_FAIL(LTX_ERR_ARG, "ltx_op_attention_compact: bf16, head_dim 64, at most 128 keys, dense k / v rows");
    hipStream_t s = (hipStream_t)stream;
    const int D = heads * hd;
    void *kc = nullptr, *vc = nullptr, *bc = nullptr, *idx = nullptr, *cnt = nullptr;
    auto free_all = [&]() { for (void* p : {kc, vc, bc, idx, cnt}) if (p) (void)hipFree(p); };
    const size_t rows = (size_t)B * Sk;
    if (hipMalloc(&kc, rows * D * 2) != hipSuccess || hipMalloc(&vc, rows * D * 2) != hipSuccess || hipMalloc(&bc, rows * 4) != hipSuccess ||
        hipMalloc(&idx, rows * 4) != hipSuccess || hipMalloc(&cnt, (size_t)B * 4) != hipSuccess) { free_all(); LTX_FAIL(LTX_ERR_HIP, "hipMalloc"); }
    int rc = ltx_launch_key_compact(key_bias, B, Sk, (int*)idx, (int*)cnt, (float*)bc, s);
    if (rc == LTX_OK) rc = ltx_launch_gather_rows(k, kc, (const int*)idx, (const int*)cnt, 1, B, Sk, D * 2, s);
    if (rc == LTX_OK) rc = ltx_launch_gather_rows(v, vc, (const int*)idx, (const int*)cnt, 1, B, Sk, D * 2, s);
    if (rc == LTX_OK) {
        AttnArgs a; a.q = q; a.k = kc; a.v = vc; a.o = o; a.ldq = ldq; a.ldk = D; a.ldv = D; a.ldo = ldo;
        a.B = B; a.Sq = Sq; a.Sk = Sk; a.heads = heads; a.hd = hd; a.scale = scale; a.bias = (const float*)bc; a.k_count = (const int*)cnt;
        if (q_rowsq) { a.q_rowsq = q_rowsq; a.q_rowsq_n = q_rowsq_n; a.q_rowsq_D = q_rowsq_D; a.q_rowsq_eps = q_rowsq_eps; }
        rc = ltx_launch_attention(a, LTX_DT_BF16, s);
    }
    if (rc == LTX_OK && counts_out && hipMemcpyAsync(counts_out, cnt, (size_t)B * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = LTX_ERR_HIP;
    const hipError_t e = hipStreamSynchronize(s);
    free_all();
    if (rc != LTX_OK) return rc;
    if (e != hipSuccess) { ltx_set_error(hipGetErrorString(e)); return LTX_ERR_HIP; }
    return LTX_OK;
}

extern "C" int ltx_op_attention_prescaled(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int heads, int hd,
                                          int ldq, int ldk, int ldv, int ldo, ltx_stream stream) {
    if (!q || !k || !v || !o) LTX_FAIL(LTX_ERR_ARG, "ltx_op_attention_prescaled: null tensor");
    AttnArgs a; a.q = q; a.k = k; a.v = v; a.o = o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.Sq = Sq; a.Sk = Sk; a.heads = heads; a.hd = hd; a.scale = 1.0f; a.q_prescaled = 1;
    return ltx_launch_attention(a, LTX_DT_BF16, (hipStream_t)stream);
}

namespace {
int conv_common(const void* x, const void* w, const void* bias, int wdtype, void* y, const void* resid,
                int B, int T, int H, int W, int Cin, int Cout, int causal, int dtype, int epi, int mode, int post, hipStream_t s) {
    if (!x || !w || !bias || !y) LTX_FAIL(LTX_ERR_ARG, "conv: null tensor");
    const int dt = dtc(dtype); const size_t esz = ltx_dt_size(dt);
    void *pw = nullptr, *pb = nullptr;
    HIP_TRY(hipMalloc(&pw, (size_t)Cout * Cin * 27 * esz));
    if (hipMalloc(&pb, (size_t)Cout * esz + 16) != hipSuccess) { (void)hipFree(pw); LTX_FAIL(LTX_ERR_HIP, "hipMalloc"); }
    const int Cf = Cout / 8;
    int rc = ltx_pack_conv(w, dtc(wdtype), pw, dt, Cout, Cin, 27, mode, Cf, s);
    if (rc == LTX_OK) rc = ltx_pack_conv(bias, dtc(wdtype), pb, dt, Cout, 1, 1, mode, Cf, s);
    if (rc == LTX_OK) {
        GemmArgs g; g.A = x; g.W = pw; g.C = y; g.bias = pb; g.resid = resid;
        g.M = B * T * H * W; g.N = Cout; g.K = Cin; g.ldc = Cout; g.ldr = Cout;
        g.conv = 1; g.B = B; g.T = T; g.H = H; g.Wd = W; g.Cin = Cin; g.ntaps = 27; g.kh = 3; g.kw = 3; g.pad_t = causal ? 2 : 1; g.post = post;
        if (epi == EPI_D2S) { g.Cf = Cf; g.Cr = Cin / 8; g.To = 2 * T - 1; g.Ho = 2 * H; g.Wo = 2 * W; }
        rc = ltx_launch_gemm(g, dt, epi, s);
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(pw); (void)hipFree(pb);
    if (rc != LTX_OK) return rc;
    if (e != hipSuccess) { ltx_set_error(hipGetErrorString(e)); return LTX_ERR_HIP; }
    return LTX_OK;
}
}  // namespace

extern "C" int ltx_op_conv3d(const void* x, const void* w, const void* bias, int wdtype, void* y, const void* resid,
                             int B, int T, int H, int W, int Cin, int Cout, int causal, int dtype, ltx_stream stream) {
    return conv_common(x, w, bias, wdtype, y, resid, B, T, H, W, Cin, Cout, causal, dtype, resid ? EPI_RESID : EPI_BIAS, LTX_PERM_NONE, 0, (hipStream_t)stream);
}
extern "C" int ltx_op_upsample3d(const void* x, const void* w, const void* bias, int wdtype, void* y,
                                 int B, int T, int H, int W, int Cin, int Cout, int causal, int residual, int dtype, ltx_stream stream) {
    if (Cout % 8 != 0 || Cin % 8 != 0) LTX_FAIL(LTX_ERR_ARG, "upsample3d: channels must be multiples of 8");
    return conv_common(x, w, bias, wdtype, y, residual ? x : nullptr, B, T, H, W, Cin, Cout, causal, dtype, EPI_D2S, LTX_PERM_D2S, 0, (hipStream_t)stream);
}
extern "C" int ltx_op_conv_out_unpatchify(const void* x, const void* w, const void* bias, int wdtype, float* y,
                                          int B, int T, int H, int W, int Cin, int Cout, int causal, int postprocess, int dtype, ltx_stream stream) {
    if (Cout % 16 != 0) LTX_FAIL(LTX_ERR_ARG, "conv_out: Cout must be a multiple of 16 (patch 4x4)");
    return conv_common(x, w, bias, wdtype, y, nullptr, B, T, H, W, Cin, Cout, causal, dtype, EPI_UNPATCH, LTX_PERM_UNPATCH, postprocess, (hipStream_t)stream);
}

extern "C" int ltx_op_blend(const float* a, float* b, int BC, int at, int ah, int aw, int bt, int bh, int bw, int dim, int blend_extent, ltx_stream stream) {
    if (!a || !b || dim < 2 || dim > 4) LTX_FAIL(LTX_ERR_ARG, "ltx_op_blend: bad argument");
    const int ad[3] = {at, ah, aw}, bd[3] = {bt, bh, bw};
    BlendArgs ba; ba.a = a; ba.b = b; ba.dst = b; ba.BC = BC;
    ba.at = at; ba.ah = ah; ba.aw = aw; ba.a_len = ad[dim - 2];
    ba.bt = ba.dt = bt; ba.bh = ba.dh = bh; ba.bw = ba.dw = bw;
    ba.dim = dim; ba.blend = std::min(blend_extent, std::min(ad[dim - 2], bd[dim - 2]));
    ba.et = std::min(at, bt); ba.eh = std::min(ah, bh); ba.ew = std::min(aw, bw);
    if (dim == 2) ba.et = ba.blend; else if (dim == 3) ba.eh = ba.blend; else ba.ew = ba.blend;
    return ltx_launch_blend(ba, (hipStream_t)stream);
}

const char* ltx_gemm_plan_name(int M, int N, int K, int conv, int ntaps, int T, int H, int W);
extern "C" int ltx_op_gemm_plan(int M, int N, int K, int conv, int ntaps, int T, int H, int W, char* name, int cap) {
    if (!name || cap < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_op_gemm_plan: bad argument");
    const char* n = ltx_gemm_plan_name(M, N, K, conv, ntaps, T, H, W);
    snprintf(name, (size_t)cap, "%s", n);
    return LTX_OK;
}
