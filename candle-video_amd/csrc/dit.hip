// LtxVideoTransformer3DModel on MI355X: weight ingestion + forward orchestration.
// Reference: src/models/ltx_video/ltx_transformer.rs (:957-1022 ctor, :1029-1172 forward,
// :820-937 block, :648-750 attention).  One kernel launch per fused stage:
//   per block: rms+AdaLN | fused QKV GEMM | qk-RMSNorm+RoPE | flash attention | to_out GEMM (+gate*x+h)
//              | q GEMM | q-norm | (k,v GEMM | k-norm) | cross attention (key bias) | to_out GEMM (+h)
//              | rms+AdaLN | FF1 GEMM (+GELU-tanh) | FF2 GEMM (+gate*x+h)
#include <cstring>
#include <deque>
#include "model_util.h"
#include "options.h"

struct DitBlock {
    LinearW qkv1, o1, q2, kv2, o2, ff1, ff2;
    void *nq1 = nullptr, *nk1 = nullptr, *nq2 = nullptr, *nk2 = nullptr;
};

struct DitCtx {                      // cached text context (see ltx_dit_forward)
    const void* enc = nullptr; const float* mask = nullptr;
    int B = 0, K = 0, iodt = 0; bool valid = false;
    bool fold_q2 = false;            // k additionally carries attn2.norm_q.weight (the q-norm folded into cross attention)
    DevBuf kv, bias;                 // [L][B*K][2D] (k already RMS-normed), [B*K]
    // keys the mask leaves alive, compacted to the front of every batch row (AttnArgs::k_count): the cross-attention kernel then
    // multiplies ceil(count / 32) key blocks instead of ceil(K / 32) - BASELINE's prompts keep 32 of 128 text tokens
    bool compact = false;
    DevBuf kvc, biasc, kidx, kcount; // [L][B*K][2D], [B*K] f32, [B*K] int, [B] int
};

// AdaLayerNormSingle's output for one set of timesteps (ltx_transformer.rs:262-309): a function of the timestep values and the
// weights alone, so the chain (sinusoid -> Linear -> SiLU -> Linear -> SiLU -> Linear(6D) -> + tables: 8 launches per forward) runs
// once per distinct timestep vector of a model and stream; a sampler revisits the same few timesteps for every video.
struct DitTimeEntry {
    float t[8] = {0}; int B = 0; hipStream_t stream = nullptr; bool valid = false; uint64_t used = 0;
    DevBuf ada, adaf;                // [L][B][6D] f32, [2][B][D] f32
    DevBuf cfold; bool cfold_valid = false;      // norm fold: per layer [B][3D] (shift_msa . W_qkv^T + b_qkv) then [B][4D] (shift_mlp . W_ff1^T + b_ff1), f32
};
// norm fold through the weights (norm_fold=2): per layer W_qkv (.) (1 + scale_msa) [3D, D] then W_ff1 (.) (1 + scale_mlp) [4D, D], model dtype.
// Keyed by the TIMESTEP alone (the modulation of a row depends on nothing else): forwards of any batch size at that timestep share the copy.
struct DitWfold { float t = 0.f; hipStream_t stream = nullptr; bool valid = false; uint64_t used = 0; DevBuf w; };
constexpr int kDitTimeEntries = 64;

struct ltx_dit {
    ltx_dit_config cfg{};
    int dtype = LTX_DT_BF16;
    int device = 0;
    int D = 0;
    LinearW proj_in, te1, te2, te_lin, cap1, cap2, proj_out;
    void* sst_final = nullptr;       // [2, D]
    void* sst_blocks = nullptr;      // [L, 6, D]
    std::vector<DitBlock> blocks;
    float* rope_freqs = nullptr;     // [D/6]
    float* inv_freq = nullptr;       // [128]
    std::vector<int> skip_blocks;
    std::deque<DitCtx> ctxs;         // deque: entries must not move while `ctx` points at one
    bool ctx_mode = false;
    std::deque<DitTimeEntry> tcache; uint64_t tclock = 0;
    std::deque<DitWfold> wcache;
    bool wfold_off = false;          // norm_fold=2 gave up on this handle: more distinct timesteps in flight than scaled-weight copies (a schedule that would re-scale every step)
    // RoPE tables of the caching scope (ltx_dit_context_cache: the caller keeps coords / geometry constant inside it): what cosb / sinb hold
    struct { bool valid = false; const float* coords = nullptr; float rs[3] = {0, 0, 0}; bool has_rs = false; int B = 0, S = 0, F = 0, H = 0, W = 0; hipStream_t stream = nullptr; } rope_key;
    std::vector<void*> owned;        // every hipMalloc'd weight pointer
    // workspaces
    DevBuf xin, encin, h, n, qkv, attn, ff, c1, encp, kv2, tproj, e1, emb, embs, temb, ada, adaf, cosb, sinb, bias, orig, orig_hsq, outT, rsq, hsq, parts;
    void free_all() {
        for (void* p : owned) if (p) (void)hipFree(p);
        owned.clear();
        DevBuf* bs[] = {&xin, &encin, &h, &n, &qkv, &attn, &ff, &c1, &encp, &kv2, &tproj, &e1, &emb, &embs, &temb, &ada, &adaf, &cosb, &sinb, &bias, &orig, &orig_hsq, &outT, &rsq, &hsq, &parts};
        for (DevBuf* b : bs) b->release();
        for (auto& e : ctxs) { e.kv.release(); e.bias.release(); e.kvc.release(); e.biasc.release(); e.kidx.release(); e.kcount.release(); }
        ctxs.clear();
        for (auto& e : tcache) { e.ada.release(); e.adaf.release(); e.cfold.release(); }
        for (auto& e : wcache) e.w.release();
        wcache.clear();
        tcache.clear();
    }
};

namespace {

int own_linear(ltx_dit* m, const WeightMap& wm, const std::string& prefix, int in, int out, LinearW* l) {
    LTX_TRY(ltx_load_linear(wm, prefix, in, out, m->dtype, l));
    m->owned.push_back(l->w); if (l->b) m->owned.push_back(l->b);
    return LTX_OK;
}
int own_tensor(ltx_dit* m, const WeightMap& wm, const std::string& name, int64_t numel, void** out) {
    LTX_TRY(ltx_load_tensor(wm, name, numel, m->dtype, out));
    m->owned.push_back(*out);
    return LTX_OK;
}
// fused [sum(out_i), in] linear from several reference linears sharing the same input
int own_fused(ltx_dit* m, const WeightMap& wm, const std::vector<std::string>& prefixes, int in, int out_each, LinearW* l) {
    const int n = (int)prefixes.size();
    const size_t esz = ltx_dt_size(m->dtype);
    l->in = in; l->out = out_each * n;
    HIP_TRY(hipMalloc(&l->w, (size_t)l->out * in * esz)); m->owned.push_back(l->w);
    bool has_bias = wm.find(prefixes[0] + ".bias") != nullptr;
    if (has_bias) { HIP_TRY(hipMalloc(&l->b, (size_t)l->out * esz)); m->owned.push_back(l->b); }
    for (int i = 0; i < n; ++i) {
        const ltx_weight* w = wm.find(prefixes[i] + ".weight");
        if (!w) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + prefixes[i] + ".weight'");
        LTX_TRY(ltx_upload_cast(w, (char*)l->w + (size_t)i * out_each * in * esz, m->dtype, (int64_t)out_each * in, prefixes[i] + ".weight"));
        if (has_bias) {
            const ltx_weight* b = wm.find(prefixes[i] + ".bias");
            if (!b) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + prefixes[i] + ".bias'");
            LTX_TRY(ltx_upload_cast(b, (char*)l->b + (size_t)i * out_each * esz, m->dtype, out_each, prefixes[i] + ".bias"));
        }
    }
    return LTX_OK;
}

int build(ltx_dit* m, const ltx_weight* weights, size_t n_weights) {
    const ltx_dit_config& c = m->cfg;
    const int D = m->D, L = c.num_layers;
    WeightMap wm(weights, n_weights);
    LTX_TRY(own_linear(m, wm, "proj_in", c.in_channels, D, &m->proj_in));
    LTX_TRY(own_tensor(m, wm, "scale_shift_table", 2 * (int64_t)D, &m->sst_final));
    LTX_TRY(own_linear(m, wm, "time_embed.emb.timestep_embedder.linear_1", 256, D, &m->te1));
    LTX_TRY(own_linear(m, wm, "time_embed.emb.timestep_embedder.linear_2", D, D, &m->te2));
    LTX_TRY(own_linear(m, wm, "time_embed.linear", D, 6 * D, &m->te_lin));
    LTX_TRY(own_linear(m, wm, "caption_projection.linear_1", c.caption_channels, D, &m->cap1));
    LTX_TRY(own_linear(m, wm, "caption_projection.linear_2", D, D, &m->cap2));
    LTX_TRY(own_linear(m, wm, "proj_out", D, c.out_channels, &m->proj_out));
    const size_t esz = ltx_dt_size(m->dtype);
    HIP_TRY(hipMalloc(&m->sst_blocks, (size_t)L * 6 * D * esz)); m->owned.push_back(m->sst_blocks);
    m->blocks.resize(L);
    for (int i = 0; i < L; ++i) {
        const std::string p = "transformer_blocks." + std::to_string(i) + ".";
        DitBlock& b = m->blocks[i];
        LTX_TRY(own_fused(m, wm, {p + "attn1.to_q", p + "attn1.to_k", p + "attn1.to_v"}, D, D, &b.qkv1));
        LTX_TRY(own_linear(m, wm, p + "attn1.to_out.0", D, D, &b.o1));
        LTX_TRY(own_tensor(m, wm, p + "attn1.norm_q.weight", D, &b.nq1));
        LTX_TRY(own_tensor(m, wm, p + "attn1.norm_k.weight", D, &b.nk1));
        LTX_TRY(own_linear(m, wm, p + "attn2.to_q", D, D, &b.q2));
        LTX_TRY(own_fused(m, wm, {p + "attn2.to_k", p + "attn2.to_v"}, c.cross_attention_dim, D, &b.kv2));
        LTX_TRY(own_linear(m, wm, p + "attn2.to_out.0", D, D, &b.o2));
        LTX_TRY(own_tensor(m, wm, p + "attn2.norm_q.weight", D, &b.nq2));
        LTX_TRY(own_tensor(m, wm, p + "attn2.norm_k.weight", D, &b.nk2));
        LTX_TRY(own_linear(m, wm, p + "ff.net.0.proj", D, 4 * D, &b.ff1));
        LTX_TRY(own_linear(m, wm, p + "ff.net.2", 4 * D, D, &b.ff2));
        const ltx_weight* t = wm.find(p + "scale_shift_table");
        if (!t) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + p + "scale_shift_table'");
        LTX_TRY(ltx_upload_cast(t, (char*)m->sst_blocks + (size_t)i * 6 * D * esz, m->dtype, 6 * (int64_t)D, p + "scale_shift_table"));
    }
    // RoPE frequency table: theta^linspace(0,1,D/6) * pi/2 with the reference's f32 roundings
    // (ltx_transformer.rs:475-488); exp evaluated correctly rounded (double -> f32).
    {
        int steps = D / 6; if (steps < 1) steps = 1;
        std::vector<float> fr(steps);
        const float theta_ln = (float)std::log(10000.0);
        for (int i = 0; i < steps; ++i) {
            float lin = steps <= 1 ? 0.0f : (float)i * (float)(1.0 / (double)(steps - 1));
            float x = lin * theta_ln;
            float e = (float)std::exp((double)x);
            fr[i] = e * (float)(M_PI / 2.0);
        }
        HIP_TRY(hipMalloc((void**)&m->rope_freqs, sizeof(float) * steps)); m->owned.push_back(m->rope_freqs);
        HIP_TRY(hipMemcpy(m->rope_freqs, fr.data(), sizeof(float) * steps, hipMemcpyHostToDevice));
        // inv_freq_i = 1 / 10000^(i/128)  (ltx_transformer.rs:288-290)
        std::vector<float> inv(128);
        ltx_sinusoid_table(0, inv.data());
        HIP_TRY(hipMalloc((void**)&m->inv_freq, sizeof(float) * 128)); m->owned.push_back(m->inv_freq);
        HIP_TRY(hipMemcpy(m->inv_freq, inv.data(), sizeof(float) * 128, hipMemcpyHostToDevice));
    }
    return LTX_OK;
}

}  // namespace

extern "C" int ltx_dit_create(const ltx_dit_config* cfg, const ltx_weight* weights, size_t n_weights,
                              ltx_dtype model_dtype, int device, ltx_dit** out) {
    if (!cfg || !weights || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_create: null argument");
    *out = nullptr;
    const int hd = cfg->attention_head_dim;
    if (hd != 16 && hd != 32 && hd != 64 && hd != 128) LTX_FAIL(LTX_ERR_UNSUPPORTED, "attention_head_dim must be 16/32/64/128");
    const int D = cfg->num_attention_heads * hd;
    if (D % 8 != 0 || cfg->in_channels % 8 != 0 || cfg->out_channels % 8 != 0 || cfg->caption_channels % 8 != 0 || cfg->cross_attention_dim % 8 != 0)
        LTX_FAIL(LTX_ERR_UNSUPPORTED, "channel dims must be multiples of 8");
    if (cfg->cross_attention_dim != D) LTX_FAIL(LTX_ERR_UNSUPPORTED, "cross_attention_dim must equal inner_dim (caption projection output feeds attn2)");
    if (cfg->num_layers < 1) LTX_FAIL(LTX_ERR_ARG, "num_layers must be >= 1");
    HIP_TRY(hipSetDevice(device));
    ltx_dit* m = new ltx_dit();
    m->cfg = *cfg; m->dtype = model_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32; m->device = device; m->D = D;
    int rc = build(m, weights, n_weights);
    if (rc == LTX_OK && cfg->attention_head_dim == 128 && m->dtype == LTX_DT_BF16) rc = ltx_attention_q128_prepare();
    if (rc != LTX_OK) { m->free_all(); delete m; return rc; }
    *out = m;
    return LTX_OK;
}

extern "C" void ltx_dit_destroy(ltx_dit* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    m->free_all();
    delete m;
}

extern "C" int ltx_dit_set_skip_blocks(ltx_dit* m, const int* blocks, int n) {
    if (!m || n < 0 || (n > 0 && !blocks)) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_set_skip_blocks: bad argument");
    m->skip_blocks.assign(blocks, blocks + n);
    return LTX_OK;
}

extern "C" int ltx_dit_get_config(const ltx_dit* m, ltx_dit_config* out) {
    if (!m || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_get_config: null argument");
    *out = m->cfg;
    return LTX_OK;
}

// one forward of up to 8 batch rows (the per-batch scalars - timesteps, skip-mask rows - travel as kernel arguments)
static int dit_forward_b8(ltx_dit* m, const void* hidden, const void* enc, const float* timestep,
                          const float* enc_mask, int B, int S, int K, int num_frames, int height, int width,
                          const float* rope_scale, const float* video_coords, const float* skip_layer_mask,
                          ltx_dtype io_dtype, void* out, ltx_stream stream) {
    if (B < 1 || B > 8) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_forward: internal batch chunk must be 1..8");
    if (S < 1 || K < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_forward: empty sequence");
    if (!video_coords && (int64_t)num_frames * height * width != S)
        LTX_FAIL(LTX_ERR_ARG, "ltx_dit_forward: num_frames*height*width must equal S when video_coords is absent");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const ltx_dit_config& c = m->cfg;
    const int dt = m->dtype, iodt = io_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32;
    const size_t esz = ltx_dt_size(dt);
    const int D = m->D, L = c.num_layers, H = c.num_attention_heads, hd = c.attention_head_dim;
    const int64_t M = (int64_t)B * S, MK = (int64_t)B * K;

    LTX_TRY(m->xin.ensure(M * c.in_channels * esz));
    LTX_TRY(m->encin.ensure(MK * c.caption_channels * esz));
    LTX_TRY(m->h.ensure(M * D * esz)); LTX_TRY(m->n.ensure(M * D * esz));
    LTX_TRY(m->qkv.ensure(M * 3 * D * esz)); LTX_TRY(m->attn.ensure(M * D * esz));
    LTX_TRY(m->ff.ensure(M * 4 * D * esz));
    LTX_TRY(m->c1.ensure(MK * D * esz)); LTX_TRY(m->encp.ensure(MK * D * esz)); LTX_TRY(m->kv2.ensure(MK * 2 * D * esz));
    LTX_TRY(m->tproj.ensure((size_t)B * 256 * esz)); LTX_TRY(m->e1.ensure((size_t)B * D * esz));
    LTX_TRY(m->emb.ensure((size_t)B * D * esz)); LTX_TRY(m->embs.ensure((size_t)B * D * esz));
    LTX_TRY(m->temb.ensure((size_t)B * 6 * D * esz));
    LTX_TRY(m->cosb.ensure(M * (D / 2) * sizeof(float))); LTX_TRY(m->sinb.ensure(M * (D / 2) * sizeof(float)));
    LTX_TRY(m->bias.ensure(MK * sizeof(float)));
    LTX_TRY(m->outT.ensure(M * c.out_channels * esz));
    // Cross-attention q-norm folded into the attention kernel (bf16, head_dim 64, <= 128 text keys): scores are linear in q, so
    // rms_norm(q) . k = r_row * (q . (k * w_q)) - the q2 projection's epilogue leaves per-row partial sums of squares
    // (GemmArgs::rowsq), w_q = attn2.norm_q.weight rides on the cached k, and the stand-alone pass over q (read + write of
    // [M, D] per layer) disappears.  ltx_transformer.rs:671-678, 719-740.
    // Only where the projection can emit the partials from its own epilogue (a shape-only test: gemm_asm16's fit): behind any
    // other kernel they cost a stand-alone pass, which at small M (C1: 384 tokens) is dearer than the q-norm pass it replaces.
    bool fold_q2 = dt == LTX_DT_BF16 && ltx_attention_rowsq_ok(hd, K, D);
    if (fold_q2) {
        GemmArgs gq; gq.A = m->h.p; gq.W = m->blocks[0].q2.w; gq.C = m->qkv.p; gq.bias = m->blocks[0].q2.b;
        gq.M = (int)M; gq.N = m->blocks[0].q2.out; gq.K = m->blocks[0].q2.in; gq.lda = D; gq.ldc = D;
        fold_q2 = ltx_gemm_asm16_fits(gq, EPI_BIAS) || ltx_opt().q2_fold == 2;       // q2_fold=2: fold whatever the shape (tests of the stand-alone partials)
    }
    if (fold_q2) LTX_TRY(m->rsq.ensure(M * (D / 128) * sizeof(float)));
    // The two RMS norms of a block take their rows' sums of squares from the epilogue of the GEMM that wrote h (ff2 of the block
    // before, attn2.to_out of this block: GemmArgs::rowsq) and run as a pure elementwise map; same shape-only condition as the
    // fold above (the partials must come from gemm_asm16's epilogue).  LTX_NORM_PRESUM=0: the row-reducing pass (A/B aid); "2" forces
    // the map whatever the shape (tests).  What it buys is not the missing reduction (a first map, one chunk per thread, took the same
    // 13.5 us per launch) but operand re-use: with four rows per thread the 64 B of f32 modulation per 16-byte chunk are loaded once
    // per four chunks - 13.7 -> 11.0 us per launch, 7.3 -> 6.3 ms of norm passes per video against +0.5 ms in the two epilogues
    // (docs/lab_notes.md R4.4 / R4.7).
    bool presum = dt == LTX_DT_BF16 && D % 512 == 0 && (D & (D - 1)) == 0 && D <= 2048;
    if (presum) {
        const int pe = ltx_opt().norm_presum;
        GemmArgs gp; gp.A = m->attn.p; gp.W = m->blocks[0].o2.w; gp.C = m->h.p; gp.bias = m->blocks[0].o2.b; gp.resid = m->h.p;
        gp.M = (int)M; gp.N = m->blocks[0].o2.out; gp.K = m->blocks[0].o2.in; gp.lda = D; gp.ldc = D; gp.ldr = D;
        GemmArgs gf = gp; gf.A = m->ff.p; gf.W = m->blocks[0].ff2.w; gf.bias = m->blocks[0].ff2.b; gf.K = m->blocks[0].ff2.in; gf.lda = 4 * D;
        gf.gate = reinterpret_cast<const float*>(m->h.p); gf.gate_stride = 6 * D; gf.rows_per_batch = S;      // (any aligned non-null pointer: a fit test, nothing is launched)
        presum = (pe != 0 && ltx_gemm_asm16_fits(gp, EPI_RESID) && ltx_gemm_asm16_fits(gf, EPI_GATE_RESID)) || pe == 2;
    }
    if (presum) LTX_TRY(m->hsq.ensure(M * (D / 128) * sizeof(float)));
    bool hsq_valid = false;                                 // m->hsq holds the partials of the CURRENT contents of h
    // Norm fold (round 6; GemmArgs::C2 / ::rs_sq, kernels.h): with the partials in hand the norm pass between the layer that writes h
    // and the layer that reads the normalised rows is gone altogether - norm(h) * (1 + sc) + sh times W^T is
    // r_m * ((h (.) (1 + sc)) W^T) + (sh W^T + b): out2 / ff2 also store h (.) (1 + sc_next) into m->n, qkv / ff1 read it and finish
    // with the row's 1 / rms and the per-timestep vector sh W^T + b (cached with the timestep's modulation).  One rounding less than
    // the pass (bf16 of h (1 + sc) instead of bf16 of the modulated, normalised row); norm_fold=0: the pass (A/B arm).
    bool nfold = presum && ltx_opt().norm_fold != 0;
    if (nfold) {
        GemmArgs gp; gp.A = m->attn.p; gp.W = m->blocks[0].o2.w; gp.C = m->h.p; gp.bias = m->blocks[0].o2.b; gp.resid = m->h.p;
        gp.M = (int)M; gp.N = D; gp.K = D; gp.lda = D; gp.ldc = D; gp.ldr = D; gp.rows_per_batch = S;
        gp.rowsq = m->hsq.as<float>(); gp.C2 = m->n.p; gp.scale2 = reinterpret_cast<const float*>(m->hsq.p); gp.scale2_stride = 6 * D;      // (aligned non-null pointers: a fit test)
        GemmArgs gf = gp; gf.A = m->ff.p; gf.W = m->blocks[0].ff2.w; gf.bias = m->blocks[0].ff2.b; gf.K = 4 * D; gf.lda = 4 * D;
        gf.gate = reinterpret_cast<const float*>(m->hsq.p); gf.gate_stride = 6 * D;
        GemmArgs gq; gq.A = m->n.p; gq.W = m->blocks[0].qkv1.w; gq.C = m->qkv.p; gq.M = (int)M; gq.N = 3 * D; gq.K = D; gq.lda = D; gq.ldc = D;
        gq.c_seg_shift = __builtin_ctz((unsigned)D); gq.c_seg_stride = M * D; gq.rows_per_batch = S;
        gq.rs_sq = m->hsq.as<float>(); gq.rs_n = D / 128; gq.rs_D = D; gq.rs_eps = c.norm_eps; gq.cvec = m->hsq.as<float>(); gq.cvec_stride = 3 * D;
        GemmArgs g1 = gq; g1.W = m->blocks[0].ff1.w; g1.C = m->ff.p; g1.N = 4 * D; g1.ldc = 4 * D; g1.c_seg_shift = 0; g1.c_seg_stride = 0; g1.cvec_stride = 4 * D;
        nfold = ltx_opt().dense_qkv && m->blocks[0].qkv1.out == 3 * D && m->blocks[0].ff1.out == 4 * D &&
                ltx_gemm_fold_ok(gp, EPI_RESID) && ltx_gemm_fold_ok(gf, EPI_GATE_RESID) && ltx_gemm_fold_ok(gq, EPI_BIAS) && ltx_gemm_fold_ok(g1, EPI_GELU);
    }
    bool hs_valid = false;                                  // m->n holds h (.) (1 + scale) of the norm that comes next (written by the layer that wrote h)
    bool hsq_saved = false;                                 // m->orig_hsq holds the partials of m->orig (a layer some rows skip)
    // Few tokens (C1's 384: every linear layer is a latency-bound weight stream): ff2, the deepest one (K = 4 D), runs its K ranges as
    // separate blocks (the shape rule: four ranges from K = 8192 up) and leaves their f32 sums in m->parts; the row norm that follows
    // the block - the next block's norm1, or the final LayerNorm - adds them in part order, applies gate * y + h, writes h and goes on
    // normalising the row it has just finished (GemmArgs::defer_parts / RowNormArgs::parts).  Same K partition and order as the
    // in-launch reduction: the same bits.  ff2_defer=0: the in-launch reduction (A/B aid).
    bool defer_ff2 = false; int ff2_parts = 1;
    if (dt == LTX_DT_BF16 && !skip_layer_mask && ltx_opt().ff2_defer && !presum) {
        GemmArgs gf; gf.A = m->ff.p; gf.W = m->blocks[0].ff2.w; gf.C = m->h.p; gf.M = (int)M; gf.N = m->blocks[0].ff2.out; gf.K = m->blocks[0].ff2.in; gf.lda = 4 * D; gf.ldc = D;
        ff2_parts = ltx_gemm_split_factor(gf);
        defer_ff2 = M <= 512 && ff2_parts > 1 && gf.N == D && ltx_gemm_defer_ok(gf, EPI_GATE_RESID);
    }
    if (defer_ff2) LTX_TRY(m->parts.ensure((size_t)ff2_parts * M * D * sizeof(float)));
    const float* pend_gate = nullptr; const void* pend_bias = nullptr; bool pending = false;      // h's rows are still K-range sums in m->parts
    auto take_pending = [&](RowNormArgs& rn) {
        if (!pending) return;
        rn.parts = m->parts.as<float>(); rn.nparts = ff2_parts; rn.part_stride = M * D; rn.d_bias = pend_bias; rn.d_gate = pend_gate; rn.d_gate_stride = 6 * D;
        rn.x_out = m->h.p; pending = false;
    };
    if (skip_layer_mask) LTX_TRY(m->orig.ensure(M * D * esz));

    // inputs -> model dtype (:1045-1047)
    LTX_TRY(ltx_launch_cast(hidden, iodt, m->xin.p, dt, M * c.in_channels, s));
    LTX_TRY(ltx_linear(m->proj_in, m->xin.p, c.in_channels, m->h.p, D, (int)M, dt, EPI_BIAS, s));

    // AdaLayerNormSingle (:262-267): sinusoid(256) -> Linear -> SiLU -> Linear = embedded_timestep ; SiLU -> Linear(6D) = temb
    TimeVec tv; tv.n = B; for (int i = 0; i < 8; ++i) tv.t[i] = i < B ? timestep[i] : 0.f;
    DitTimeEntry* te = nullptr;
    for (auto& e : m->tcache) if (e.valid && e.B == B && e.stream == s && !memcmp(e.t, tv.t, sizeof(float) * B)) te = &e;
    if (!te) {
        if ((int)m->tcache.size() < kDitTimeEntries) { m->tcache.emplace_back(); te = &m->tcache.back(); }
        else { te = &m->tcache.front(); for (auto& e : m->tcache) if (e.used < te->used) te = &e; }
        te->valid = false; te->cfold_valid = false;
        LTX_TRY(te->ada.ensure((size_t)L * B * 6 * D * sizeof(float))); LTX_TRY(te->adaf.ensure((size_t)2 * B * D * sizeof(float)));
        LTX_TRY(ltx_launch_sinusoid(m->tproj.p, dt, tv, m->inv_freq, 128, /*round_t=*/dt == LTX_DT_BF16, 1.0f, s));
        LTX_TRY(ltx_linear(m->te1, m->tproj.p, 256, m->e1.p, D, B, dt, EPI_BIAS, s));
        LTX_TRY(ltx_launch_silu(m->e1.p, m->e1.p, (int64_t)B * D, dt, s));
        LTX_TRY(ltx_linear(m->te2, m->e1.p, D, m->emb.p, D, B, dt, EPI_BIAS, s));
        LTX_TRY(ltx_launch_silu(m->emb.p, m->embs.p, (int64_t)B * D, dt, s));
        LTX_TRY(ltx_linear(m->te_lin, m->embs.p, D, m->temb.p, 6 * D, B, dt, EPI_BIAS, s));
        LTX_TRY(ltx_launch_ada(te->ada.as<float>(), m->sst_blocks, m->temb.p, L, B, 6 * D, dt, s));
        LTX_TRY(ltx_launch_ada(te->adaf.as<float>(), m->sst_final, m->emb.p, 2, B, D, dt, s));
        memcpy(te->t, tv.t, sizeof(te->t)); te->B = B; te->stream = s; te->valid = true;
    }
    te->used = ++m->tclock;
    const float* ada_all = te->ada.as<float>();
    const float* adaf = te->adaf.as<float>();
    if (nfold && !te->cfold_valid) {                        // once per distinct timestep vector: streams the q|k|v and ff1 weights of every layer once
        LTX_TRY(te->cfold.ensure((size_t)L * B * 7 * D * sizeof(float)));
        for (int l = 0; l < L; ++l) {
            const float* ada = ada_all + (size_t)l * B * 6 * D;
            float* cq = te->cfold.as<float>() + (size_t)l * B * 7 * D;
            LTX_TRY(ltx_launch_shift_gemv(m->blocks[l].qkv1.w, m->blocks[l].qkv1.b, ada, 6 * D, B, 3 * D, D, cq, 3 * D, s));
            LTX_TRY(ltx_launch_shift_gemv(m->blocks[l].ff1.w, m->blocks[l].ff1.b, ada + 3 * D, 6 * D, B, 4 * D, D, cq + (size_t)B * 3 * D, 4 * D, s));
        }
        te->cfold_valid = true;
    }
    // norm_fold=2: the (1 + scale) factor rides on the CONSUMER's weights instead of on a second output of the producer:
    // (h (.) (1 + sc)) W^T = h (W (.) (1 + sc))^T.  One scaled copy of the q|k|v and ff1 weights per distinct timestep (1.6 GB at 2B:
    // read + written once, then cached like the modulation they are made from - a distilled schedule has 7), all batch rows at one
    // timestep (what LtxPipeline::call passes, t2v_pipeline.rs:868); otherwise, or when the schedule has more distinct timesteps
    // than copies (norm_fold_copies; the 40-step presets), the second-output form serves.  bf16 rounding moves from h (1 + sc) to W (1 + sc).
    bool wf = nfold && ltx_opt().norm_fold == 2 && !m->wfold_off;
    for (int i = 1; i < B; ++i) wf = wf && tv.t[i] == tv.t[0];
    DitWfold* we = nullptr;
    if (wf) {
        for (auto& e : m->wcache) if (e.valid && e.t == tv.t[0] && e.stream == s) we = &e;
        if (!we) {
            const int cap = ltx_opt().norm_fold_copies > 0 ? ltx_opt().norm_fold_copies : 1;
            int live = 0; DitWfold* victim = nullptr;
            for (auto& e : m->wcache) if (e.valid) { ++live; if (!victim || e.used < victim->used) victim = &e; }
            if (live >= cap) {
                if (m->tclock - victim->used < (uint64_t)4 * cap) {      // its timestep ran a moment ago: the schedule cycles through more timesteps than copies
                    m->wfold_off = true; wf = false;
                    HIP_TRY(hipStreamSynchronize(s));               // (earlier forwards on this stream may still read the copies)
                    for (auto& e : m->wcache) { e.w.release(); e.valid = false; }
                } else {                                                // its buffer is re-used (same size)
                    if (victim->stream != s) HIP_TRY(hipStreamSynchronize(victim->stream));      // (forwards on ITS stream may still read it)
                    victim->valid = false; we = victim;
                }
            }
            if (wf) {
                if (!we) { for (auto& e : m->wcache) if (!e.valid) { we = &e; break; } }
                if (!we) { m->wcache.emplace_back(); we = &m->wcache.back(); }
                const size_t per_layer = (size_t)7 * D * D * esz;
                if (we->w.ensure((size_t)L * per_layer) != LTX_OK) {      // no room for another copy: the second-output form from here on (speed only)
                    (void)hipGetLastError();
                    m->wfold_off = true; wf = false; we = nullptr;
                }
                if (wf)
                for (int l = 0; l < L; ++l) {
                    const float* ada = ada_all + (size_t)l * B * 6 * D;
                    char* wl = (char*)we->w.p + (size_t)l * per_layer;
                    LTX_TRY(ltx_launch_scale_cols(m->blocks[l].qkv1.w, ada + D, wl, 3 * D, D, dt, s));
                    LTX_TRY(ltx_launch_scale_cols(m->blocks[l].ff1.w, ada + 4 * D, wl + (size_t)3 * D * D * esz, 4 * D, D, dt, s));
                }
                if (wf) { we->t = tv.t[0]; we->stream = s; we->valid = true; }
            }
        }
        if (wf) we->used = m->tclock;
    }

    // Text context: caption projection (:186-190), mask bias (:1059-1070) and, for every layer, the cross-attention
    // K/V projections + k-RMSNorm (:667-672).  None of it depends on the timestep or the latents, so inside a
    // caching scope (ltx_dit_context_cache) it is computed once per (enc, mask) pair instead of once per forward.
    DitCtx* ctx = nullptr;
    for (auto& e : m->ctxs) if (e.valid && e.enc == enc && e.mask == enc_mask && e.B == B && e.K == K && e.iodt == iodt && e.fold_q2 == fold_q2) ctx = &e;
    if (!ctx) {
        if (m->ctxs.size() >= 4 || !m->ctx_mode) { for (auto& e : m->ctxs) e.valid = false; }
        for (auto& e : m->ctxs) if (!e.valid) { ctx = &e; break; }
        if (!ctx) { m->ctxs.emplace_back(); ctx = &m->ctxs.back(); }
        ctx->enc = enc; ctx->mask = enc_mask; ctx->B = B; ctx->K = K; ctx->iodt = iodt; ctx->fold_q2 = fold_q2;
        LTX_TRY(ctx->kv.ensure((size_t)L * MK * 2 * D * esz)); LTX_TRY(ctx->bias.ensure(MK * sizeof(float)));
        LTX_TRY(ltx_launch_cast(enc, iodt, m->encin.p, dt, MK * c.caption_channels, s));
        LTX_TRY(ltx_linear(m->cap1, m->encin.p, c.caption_channels, m->c1.p, D, (int)MK, dt, EPI_GELU, s));
        LTX_TRY(ltx_linear(m->cap2, m->c1.p, D, m->encp.p, D, (int)MK, dt, EPI_BIAS, s));
        if (enc_mask) LTX_TRY(ltx_launch_mask_bias(ctx->bias.as<float>(), enc_mask, MK, s));
        for (int l = 0; l < L; ++l) {
            void* kvl = (char*)ctx->kv.p + (size_t)l * MK * 2 * D * esz;
            LTX_TRY(ltx_linear(m->blocks[l].kv2, m->encp.p, D, kvl, 2 * D, (int)MK, dt, EPI_BIAS, s));
            QkNormRopeArgs k2; k2.x = kvl; k2.rows = MK; k2.D = D; k2.ld = 2 * D; k2.nseg = 1; k2.w0 = m->blocks[l].nk2; k2.eps = 1e-5f;
            if (fold_q2) k2.w0b = m->blocks[l].nq2;
            LTX_TRY(ltx_launch_qknorm_rope(k2, dt, s));
        }
        // Masked text tokens (bias -10000, :1059-1070) get softmax weight exp(s - 10000 - max) = +0.0f: exactly nothing.  Where the
        // short-key-set kernel serves the layer, the keys that are left are moved to the front of their batch row once per context and
        // the kernel sizes its work by their number (device-side count: no host synchronisation).
        ctx->compact = enc_mask && dt == LTX_DT_BF16 && ltx_attention_cross64_ok(hd, K) && ltx_opt().xattn_compact;     // xattn_compact=0: every layer multiplies all K keys (A/B aid)
        if (ctx->compact) {
            LTX_TRY(ctx->kvc.ensure((size_t)L * MK * 2 * D * esz)); LTX_TRY(ctx->biasc.ensure(MK * sizeof(float)));
            LTX_TRY(ctx->kidx.ensure(MK * sizeof(int))); LTX_TRY(ctx->kcount.ensure((size_t)B * sizeof(int)));
            LTX_TRY(ltx_launch_key_compact(ctx->bias.as<float>(), B, K, ctx->kidx.as<int>(), ctx->kcount.as<int>(), ctx->biasc.as<float>(), s));
            LTX_TRY(ltx_launch_gather_rows(ctx->kv.p, ctx->kvc.p, ctx->kidx.as<int>(), ctx->kcount.as<int>(), L, B, K, (int)(2 * D * esz), s));
        }
        ctx->valid = true;     // outside a caching scope the entry is invalidated again at the end of this forward
    }
    const float* bias = enc_mask ? ctx->bias.as<float>() : nullptr;

    // RoPE tables (:436-524); inside a caching scope the tables of the previous forward are kept when coords / geometry are the same
    {
        auto& rk = m->rope_key;
        const bool same = m->ctx_mode && rk.valid && rk.coords == video_coords && rk.B == B && rk.S == S && rk.F == num_frames && rk.H == height && rk.W == width &&
                          rk.stream == s && rk.has_rs == (rope_scale != nullptr) && (!rope_scale || !memcmp(rk.rs, rope_scale, sizeof(rk.rs)));
        if (!same) {
            rk.valid = false;
            RopeTableArgs r;
            r.cos = m->cosb.as<float>(); r.sin = m->sinb.as<float>(); r.freqs = m->rope_freqs;
            r.B = B; r.D = D;
            if (video_coords) {
                r.use_coords = 1; r.coords = video_coords; r.F = 1; r.H = 1; r.W = S;
                r.gscale[0] = (float)(1.0 / 20.0); r.gscale[1] = (float)(1.0 / 2048.0); r.gscale[2] = (float)(1.0 / 2048.0);
            } else {
                r.F = num_frames; r.H = height; r.W = width;
                if (rope_scale) {
                    r.gscale[0] = (float)((double)rope_scale[0] * c.patch_size_t / 20.0);
                    r.gscale[1] = (float)((double)rope_scale[1] * c.patch_size / 2048.0);
                    r.gscale[2] = (float)((double)rope_scale[2] * c.patch_size / 2048.0);
                }
            }
            LTX_TRY(ltx_launch_rope_table(r, s));
            rk.coords = video_coords; rk.B = B; rk.S = S; rk.F = num_frames; rk.H = height; rk.W = width; rk.stream = s;
            rk.has_rs = rope_scale != nullptr; if (rope_scale) memcpy(rk.rs, rope_scale, sizeof(rk.rs));
            rk.valid = m->ctx_mode;
        }
    }

    const float attn_scale = 1.0f / std::sqrt((float)hd);
    auto next_block = [&](int l) {                          // the next block that runs (not in the skip list, not skipped by every row), -1: none
        for (int n = l + 1; n < L; ++n) {
            bool skip = false;
            for (int sb : m->skip_blocks) if (sb == n) skip = true;
            if (skip) continue;
            if (skip_layer_mask) { bool all = true; for (int bb = 0; bb < B; ++bb) all &= skip_layer_mask[(size_t)n * B + bb] == 1.f; if (all) continue; }
            return n;
        }
        return -1;
    };
    for (int l = 0; l < L; ++l) {
        bool skip = false;
        for (int sb : m->skip_blocks) if (sb == l) skip = true;
        if (skip) continue;                                        // :1094-1096
        // skip_layer_mask (:1098-1123): all-ones rows make the block an exact identity
        TimeVec mv; mv.n = B; bool any = false, all = true;
        for (int i = 0; i < 8; ++i) mv.t[i] = 0.f;
        if (skip_layer_mask) {
            for (int b = 0; b < B; ++b) { mv.t[b] = skip_layer_mask[(size_t)l * B + b]; any |= mv.t[b] != 0.f; all &= mv.t[b] == 1.f; }
            if (all) continue;
            if (any) {
                HIP_TRY(hipMemcpyAsync(m->orig.p, m->h.p, M * D * esz, hipMemcpyDeviceToDevice, s));
                // the row partials of the kept rows travel with them (restored after the blend): a batch whose rows skip different
                // layers - the guidance branches of a step in one forward - then returns, row for row, the bits of separate forwards
                hsq_saved = presum && hsq_valid;
                if (hsq_saved) { LTX_TRY(m->orig_hsq.ensure(M * (D / 128) * sizeof(float))); HIP_TRY(hipMemcpyAsync(m->orig_hsq.p, m->hsq.p, M * (D / 128) * sizeof(float), hipMemcpyDeviceToDevice, s)); }
            }
        }
        const DitBlock& b = m->blocks[l];
        const float* ada = ada_all + (size_t)l * B * 6 * D;
        // norm1 + AdaLN (shift_msa = row 0, scale_msa = row 1)
        RowNormArgs rn; rn.x = m->h.p; rn.y = m->n.p; rn.rows = M; rn.D = D; rn.ldx = D; rn.ldy = D;
        rn.kind = 0; rn.eps = c.norm_eps; rn.shift = ada; rn.scale = ada + D; rn.rows_per_batch = S; rn.mod_stride = 6 * D;
        if (presum && hsq_valid) { rn.presum = m->hsq.as<float>(); rn.presum_n = D / 128; }
        const bool fold1 = nfold && hsq_valid && (wf || hs_valid);      // the layer that wrote h left its row partials and h (.) (1 + scale_msa) in m->n (or the factor is in the weights): no pass
        const float* cfold_l = nfold ? te->cfold.as<float>() + (size_t)l * B * 7 * D : nullptr;
        const char* wfold_l = wf ? (const char*)we->w.p + (size_t)l * 7 * D * D * esz : nullptr;
        if (!fold1) {
        take_pending(rn);
        LTX_TRY(ltx_launch_rownorm(rn, dt, s));
        }
        hs_valid = false;
        rn.parts = nullptr; rn.nparts = 0; rn.x_out = nullptr; rn.d_bias = nullptr; rn.d_gate = nullptr;
        // self attention
        // q, k, v leave the fused projection as three DENSE [M, D] matrices (segmented GEMM output) when D is a power
        // of two: the attention kernel reads K/V rows of a dense matrix 7-11 % faster than column slices of [M, 3D]
        const bool dense_qkv = (D & (D - 1)) == 0 && ltx_opt().dense_qkv;      // dense_qkv=0: column slices of [M, 3D] (A/B aid, and the path of a D that is not a power of two)
        const int64_t seg = dense_qkv ? M * D : D;           // elements from q to k to v
        const int ldqkv = dense_qkv ? D : 3 * D;
        {
            GemmArgs g;
            g.A = m->n.p; g.W = b.qkv1.w; g.C = m->qkv.p; g.bias = b.qkv1.b;
            g.M = (int)M; g.N = b.qkv1.out; g.K = b.qkv1.in; g.lda = D; g.ldc = ldqkv;
            if (dense_qkv) { g.c_seg_shift = __builtin_ctz((unsigned)D); g.c_seg_stride = seg; }
            if (fold1) {
                g.bias = nullptr; g.rows_per_batch = S; g.rs_sq = m->hsq.as<float>(); g.rs_n = D / 128; g.rs_D = D; g.rs_eps = c.norm_eps; g.cvec = cfold_l; g.cvec_stride = 3 * D;
                if (wf) { g.A = m->h.p; g.W = wfold_l; }
            }
            LTX_TRY(ltx_launch_gemm(g, dt, EPI_BIAS, s));
        }
        QkNormRopeArgs qa; qa.x = m->qkv.p; qa.rows = M; qa.D = D; qa.ld = ldqkv; qa.seg_stride = seg; qa.nseg = 2; qa.w0 = b.nq1; qa.w1 = b.nk1;
        qa.eps = 1e-5f; qa.cos = m->cosb.as<float>(); qa.sin = m->sinb.as<float>();
        // bf16: q leaves the norm already multiplied by scale*log2(e) (ONE bf16 rounding, of the product), so the
        // attention kernel's exponent is exp2(S - m) with no per-score multiply
        const bool fold_q = dt == LTX_DT_BF16 && ltx_attention_prescale_ok(hd);
        if (fold_q) qa.out_scale0 = attn_scale * 1.4426950408889634f;
        LTX_TRY(ltx_launch_qknorm_rope(qa, dt, s));
        AttnArgs at; at.q = m->qkv.p; at.k = (char*)m->qkv.p + (size_t)seg * esz; at.v = (char*)m->qkv.p + (size_t)2 * seg * esz; at.o = m->attn.p;
        at.ldq = at.ldk = at.ldv = ldqkv; at.ldo = D; at.B = B; at.Sq = S; at.Sk = S; at.heads = H; at.hd = hd; at.scale = attn_scale;
        at.q_prescaled = fold_q ? 1 : 0;
        LTX_TRY(ltx_launch_attention(at, dt, s));
        // h = h + gate_msa * to_out(attn)     (gate_msa = row 2)
        LTX_TRY(ltx_linear(b.o1, m->attn.p, D, m->h.p, D, (int)M, dt, EPI_GATE_RESID, s, m->h.p, D, ada + 2 * D, 6 * D, S));
        hsq_valid = false;
        // cross attention (no pre-norm, no RoPE, q/k RMSNorm, additive key bias)
        const char* kvl = (const char*)(ctx->compact ? ctx->kvc.p : ctx->kv.p) + (size_t)l * MK * 2 * D * esz;
        AttnArgs ax; ax.q = m->qkv.p; ax.k = kvl; ax.v = kvl + (size_t)D * esz; ax.o = m->attn.p;
        ax.ldq = D; ax.ldk = ax.ldv = 2 * D; ax.ldo = D; ax.B = B; ax.Sq = S; ax.Sk = K; ax.heads = H; ax.hd = hd; ax.scale = attn_scale; ax.bias = bias;
        if (ctx->compact) { ax.bias = ctx->biasc.as<float>(); ax.k_count = ctx->kcount.as<int>(); }
        if (fold_q2) {
            GemmArgs g;
            g.A = m->h.p; g.W = b.q2.w; g.C = m->qkv.p; g.bias = b.q2.b; g.M = (int)M; g.N = b.q2.out; g.K = b.q2.in; g.lda = D; g.ldc = D;
            g.rowsq = m->rsq.as<float>();
            LTX_TRY(ltx_launch_gemm(g, dt, EPI_BIAS, s));
            ax.q_rowsq = m->rsq.as<float>(); ax.q_rowsq_n = D / 128; ax.q_rowsq_D = D; ax.q_rowsq_eps = 1e-5f;
        } else {
            LTX_TRY(ltx_linear(b.q2, m->h.p, D, m->qkv.p, D, (int)M, dt, EPI_BIAS, s));
            QkNormRopeArgs q2; q2.x = m->qkv.p; q2.rows = M; q2.D = D; q2.ld = D; q2.nseg = 1; q2.w0 = b.nq2; q2.eps = 1e-5f;
            LTX_TRY(ltx_launch_qknorm_rope(q2, dt, s));
        }
        LTX_TRY(ltx_launch_attention(ax, dt, s));
        if (nfold && !wf) {                                    // + h (.) (1 + scale_mlp) into m->n for ff1
            GemmArgs g; g.A = m->attn.p; g.W = b.o2.w; g.C = m->h.p; g.bias = b.o2.b; g.resid = m->h.p; g.M = (int)M; g.N = b.o2.out; g.K = b.o2.in;
            g.lda = D; g.ldc = D; g.ldr = D; g.rows_per_batch = S; g.rowsq = m->hsq.as<float>();
            g.C2 = m->n.p; g.scale2 = ada + 4 * D; g.scale2_stride = 6 * D;
            LTX_TRY(ltx_launch_gemm(g, dt, EPI_RESID, s));
            hs_valid = true;
        } else
        LTX_TRY(ltx_linear(b.o2, m->attn.p, D, m->h.p, D, (int)M, dt, EPI_RESID, s, m->h.p, D, nullptr, 0, 1, presum ? m->hsq.as<float>() : nullptr));
        hsq_valid = presum;
        // MLP (shift_mlp = row 3, scale_mlp = row 4, gate_mlp = row 5)
        rn.shift = ada + 3 * D; rn.scale = ada + 4 * D;
        rn.presum = nullptr; rn.presum_n = 0;
        if (presum && hsq_valid) { rn.presum = m->hsq.as<float>(); rn.presum_n = D / 128; }
        if (nfold && hsq_valid && (wf || hs_valid)) {
            GemmArgs g; g.A = m->n.p; g.W = b.ff1.w; g.C = m->ff.p; g.M = (int)M; g.N = b.ff1.out; g.K = b.ff1.in; g.lda = D; g.ldc = 4 * D; g.rows_per_batch = S;
            g.rs_sq = m->hsq.as<float>(); g.rs_n = D / 128; g.rs_D = D; g.rs_eps = c.norm_eps; g.cvec = cfold_l + (size_t)B * 3 * D; g.cvec_stride = 4 * D;
            if (wf) { g.A = m->h.p; g.W = wfold_l + (size_t)3 * D * D * esz; }
            LTX_TRY(ltx_launch_gemm(g, dt, EPI_GELU, s));
        } else {
        LTX_TRY(ltx_launch_rownorm(rn, dt, s));
        LTX_TRY(ltx_linear(b.ff1, m->n.p, D, m->ff.p, 4 * D, (int)M, dt, EPI_GELU, s));
        }
        hs_valid = false;
        if (defer_ff2) {
            GemmArgs g; g.A = m->ff.p; g.W = b.ff2.w; g.C = m->h.p; g.M = (int)M; g.N = b.ff2.out; g.K = b.ff2.in; g.lda = 4 * D; g.ldc = D;
            g.defer_parts = m->parts.as<float>();
            LTX_TRY(ltx_launch_gemm(g, dt, EPI_BIAS, s));
            pending = true; pend_gate = ada + 5 * D; pend_bias = b.ff2.b;
        } else if (int ln = (nfold && !wf) ? next_block(l) : -1; ln >= 0) {      // + h (.) (1 + scale_msa of the next block that runs) into m->n for its q|k|v projection
            GemmArgs g; g.A = m->ff.p; g.W = b.ff2.w; g.C = m->h.p; g.bias = b.ff2.b; g.resid = m->h.p; g.M = (int)M; g.N = b.ff2.out; g.K = b.ff2.in;
            g.lda = 4 * D; g.ldc = D; g.ldr = D; g.gate = ada + 5 * D; g.gate_stride = 6 * D; g.rows_per_batch = S; g.rowsq = m->hsq.as<float>();
            g.C2 = m->n.p; g.scale2 = ada_all + (size_t)ln * B * 6 * D + D; g.scale2_stride = 6 * D;
            LTX_TRY(ltx_launch_gemm(g, dt, EPI_GATE_RESID, s));
            hs_valid = true;
        } else
        LTX_TRY(ltx_linear(b.ff2, m->ff.p, 4 * D, m->h.p, D, (int)M, dt, EPI_GATE_RESID, s, m->h.p, D, ada + 5 * D, 6 * D, S, presum ? m->hsq.as<float>() : nullptr));
        hsq_valid = presum;
        if (skip_layer_mask && any) {
            LTX_TRY(ltx_launch_skip_blend(m->h.p, m->orig.p, mv, S, D, dt, s));
            bool binary = true;
            for (int bb = 0; bb < B; ++bb) binary &= mv.t[bb] == 0.f || mv.t[bb] == 1.f;
            if (hsq_valid && hsq_saved && binary) {            // rows with mask 1 are the block's input again: so are their partials
                const size_t rowb = (size_t)S * (D / 128) * sizeof(float);
                for (int bb = 0; bb < B; ++bb)
                    if (mv.t[bb] == 1.f) HIP_TRY(hipMemcpyAsync((char*)m->hsq.p + bb * rowb, (const char*)m->orig_hsq.p + bb * rowb, rowb, hipMemcpyDeviceToDevice, s));
            } else hsq_valid = false;
            // m->n was formed from the un-blended rows, for the block after this one; the restored rows need theirs: the producer's
            // expression on the rows as they stand now (for the rows that kept the block: the bits the epilogue wrote)
            hs_valid = false;
            if (nfold && !wf && hsq_valid) {
                if (const int ln = next_block(l); ln >= 0) {
                    LTX_TRY(ltx_launch_mod_scale(m->h.p, ada_all + (size_t)ln * B * 6 * D + D, 6 * D, m->n.p, B, S, D, dt, s));
                    hs_valid = true;
                }
            }
        }
    }

    // final LayerNorm (no affine) + modulation (:1126-1161), proj_out (:1163)
    {
        RowNormArgs rn; rn.x = m->h.p; rn.y = m->n.p; rn.rows = M; rn.D = D; rn.ldx = D; rn.ldy = D;
        rn.kind = 1; rn.eps = 1e-6f; rn.shift = adaf; rn.scale = adaf + (size_t)B * D;
        rn.rows_per_batch = S; rn.mod_stride = D;
        take_pending(rn);
        LTX_TRY(ltx_launch_rownorm(rn, dt, s));
        void* dst = iodt == dt ? out : m->outT.p;
        LTX_TRY(ltx_linear(m->proj_out, m->n.p, D, dst, c.out_channels, (int)M, dt, EPI_BIAS, s));
        if (iodt != dt) LTX_TRY(ltx_launch_cast(m->outT.p, dt, out, iodt, M * c.out_channels, s));
    }
    if (!m->ctx_mode) ctx->valid = false;
    return LTX_OK;
}

// The trait puts no bound on the batch (t2v_pipeline.rs:68-80).  Batch rows never interact in the forward
// (ltx_transformer.rs:1029-1172: every op is per row or per (row, token)), so a larger batch runs as chunks of 8 rows with the
// same results as one call would give.
extern "C" int ltx_dit_forward(ltx_dit* m, const void* hidden, const void* enc, const float* timestep,
                               const float* enc_mask, int B, int S, int K, int num_frames, int height, int width,
                               const float* rope_scale, const float* video_coords, const float* skip_layer_mask,
                               ltx_dtype io_dtype, void* out, ltx_stream stream) {
    if (!m || !hidden || !enc || !timestep || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_forward: null argument");
    if (B < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_forward: batch must be at least 1");
    if (B <= 8) return dit_forward_b8(m, hidden, enc, timestep, enc_mask, B, S, K, num_frames, height, width, rope_scale, video_coords, skip_layer_mask, io_dtype, out, stream);
    const size_t esz = io_dtype == LTX_BF16 ? 2 : 4;
    const int L = m->cfg.num_layers;
    std::vector<float> mask_chunk;
    for (int b0 = 0; b0 < B; b0 += 8) {
        const int bc = B - b0 < 8 ? B - b0 : 8;
        const float* slm = nullptr;
        if (skip_layer_mask) {                              // [L, B] -> [L, bc]
            mask_chunk.resize((size_t)L * bc);
            for (int l = 0; l < L; ++l) for (int b = 0; b < bc; ++b) mask_chunk[(size_t)l * bc + b] = skip_layer_mask[(size_t)l * B + b0 + b];
            slm = mask_chunk.data();
        }
        LTX_TRY(dit_forward_b8(m, (const char*)hidden + (size_t)b0 * S * m->cfg.in_channels * esz, (const char*)enc + (size_t)b0 * K * m->cfg.caption_channels * esz,
                               timestep + b0, enc_mask ? enc_mask + (size_t)b0 * K : nullptr, bc, S, K, num_frames, height, width, rope_scale,
                               video_coords ? video_coords + (size_t)b0 * S * 3 : nullptr, slm, io_dtype,
                               (char*)out + (size_t)b0 * S * m->cfg.out_channels * esz, stream));
    }
    return LTX_OK;
}

extern "C" int ltx_dit_context_cache(ltx_dit* m, int enable) {
    if (!m) LTX_FAIL(LTX_ERR_ARG, "ltx_dit_context_cache: null handle");
    for (auto& e : m->ctxs) e.valid = false;
    m->rope_key.valid = false;
    m->ctx_mode = enable != 0;
    return LTX_OK;
}
