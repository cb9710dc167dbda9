// T5 v1.1 encoder on the path's kernels (include/ltxhip_t5.h).  Per layer: T5LayerNorm (rownorm RMS + weight) -> fused
// q|k|v GEMM -> attention with the shared [H,S,S] relative-position bias (unscaled scores, f32 softmax) -> o GEMM + h ->
// T5LayerNorm -> fused wi_0|wi_1 GEMM -> NewGelu(a)*b -> wo GEMM + h.  The prompt is <= 512 tokens, so the attention is a
// small exact-f32 kernel (one query per lane); the GEMMs are the DiT's.
#include <memory>
#include <string>
#include <vector>
#include "model_util.h"
#include "options.h"
#include "../../include/ltxhip_t5.h"
#include "../../include/ltxhip_weights.h"
#include <cstring>

struct ltx_t5 {
    ltx_t5_config cfg;
    int dtype = LTX_DT_BF16, device = 0;
    std::vector<void*> owned;
    void* embed = nullptr;              // [vocab, d_model]
    float* rel_table = nullptr;         // f32 [buckets, H]
    struct Layer { LinearW qkv, o, wi, wo; void *ln0 = nullptr, *ln1 = nullptr; };
    std::vector<Layer> layers;
    void* final_ln = nullptr;
    DevBuf ids, h, n, qkv, att, ff, act, bias, kmask;
    ~ltx_t5() { for (void* p : owned) (void)hipFree(p); ids.release(); h.release(); n.release(); qkv.release(); att.release(); ff.release(); act.release(); bias.release(); kmask.release(); }
};

namespace {

template <typename T>
__global__ void t5_embed_kernel(const int* ids, const T* table, T* out, int64_t rows, int D, int vocab) {
    const int64_t n = rows * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D; int id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        out[i] = table[(int64_t)id * D + (i - r * D)];
    }
}

// position_bias[h][i][j] = table[bucket(j - i)][h]   (T5Attention._relative_position_bucket, bidirectional)
__global__ void t5_bias_kernel(const float* table, float* bias, int H, int S, int num_buckets, int max_distance) {
    const int64_t n = (int64_t)H * S * S;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % S); const int i = (int)((idx / S) % S); const int h = (int)(idx / ((int64_t)S * S));
        const int rel = j - i;
        const int nb = num_buckets / 2, max_exact = nb / 2;
        int b = rel > 0 ? nb : 0;
        const int a = rel < 0 ? -rel : rel;
        if (a < max_exact) b += a;
        else {
            int large = max_exact + (int)(logf((float)a / (float)max_exact) / logf((float)max_distance / (float)max_exact) * (float)(nb - max_exact));
            b += large < nb - 1 ? large : nb - 1;
        }
        bias[idx] = table[(int64_t)b * H + h];
    }
}

// gated NewGelu: act[m][j] = gelu_tanh(ff[m][j]) * ff[m][d_ff + j]
template <typename T>
__global__ void t5_gate_kernel(const T* ff, T* act, int64_t rows, int dff) {
    const int64_t n = rows * dff;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dff; const int64_t j = i - r * dff;
        const float a = (float)ff[r * 2 * dff + j], b = (float)ff[r * 2 * dff + dff + j];
        act[i] = (T)(gelu_tanh_f(a) * b);
    }
}

// exact-f32 attention with a per-head [S,S] bias (+ an additive per-key mask [B,S] or null: the quantised encoder's
// (1 - mask) * -1e9, quantized_t5_encoder.rs:624-634).  One WAVE per (query, head, batch); S <= 512.
//   scores : lane j of chunk c owns key 64 c + j: x = q . K[key] + bias[head][query][key] (+ mask[key]), q broadcast to every lane
//            (no 1/sqrt(d) in T5; (scores + position_bias) + mask is the reference's order);
//   softmax: exact maximum and sum over the <= 8 chunks by wave reductions (two passes over registers, no running rescale);
//   P V    : lane d owns output dim d: out[d] = sum_key p[key] V[key][d], p broadcast through LDS, V rows read coalesced.
// (The first version gave a query to each LANE and walked the keys serially: 166 us per layer at S = 128, 40 % of a T5-XXL
// forward.)
template <typename T, int HD>
__global__ __launch_bounds__(256) void t5_attn_kernel(const T* qkv, const float* bias, const float* kmask, T* out, int S, int H) {
    constexpr int MAXC = 8;                                        // 512 keys
    __shared__ float ps[4][MAXC * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + wave, head = blockIdx.y, b = blockIdx.z;
    if (qi >= S) return;                                           // whole wave: no barrier is used below
    const int inner = H * HD, ld = 3 * inner;
    const T* base = qkv + (int64_t)b * S * ld + head * HD;
    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = (float)base[(int64_t)qi * ld + d];
    const float* brow = bias + ((int64_t)head * S + qi) * S;
    const float* mrow = kmask ? kmask + (int64_t)b * S : nullptr;
    const int nc = (S + 63) >> 6;
    float x[MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        x[c] = -INFINITY;
        if (c >= nc) continue;
        const int key = c * 64 + lane;
        if (key < S) {
            const T* kr = base + (int64_t)key * ld + inner;
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc += q[d] * (float)kr[d];
            float v = acc + brow[key];
            if (mrow) v += mrow[key];
            x[c] = v;
        }
        m = fmaxf(m, x[c]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c >= nc) continue;
        const float p = (c * 64 + lane < S) ? __expf(x[c] - m) : 0.f;
        ps[wave][c * 64 + lane] = p;
        l += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    __builtin_amdgcn_wave_barrier();                               // this wave's p values are in LDS (same-wave visibility)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (HD <= 64) {
        if (lane < HD) {
            const T* vr = base + 2 * inner + lane;
            float o = 0.f;
            for (int key = 0; key < S; ++key) o += ps[wave][key] * (float)vr[(int64_t)key * ld];
            out[((int64_t)b * S + qi) * inner + head * HD + lane] = (T)(o / l);
        }
    }
}

int own_tensor(ltx_t5* m, const WeightMap& wm, const std::string& name, int64_t numel, int dtype, void** out) {
    LTX_TRY(ltx_load_tensor(wm, name, numel, dtype, out));
    m->owned.push_back(*out);
    return LTX_OK;
}
int own_fused(ltx_t5* m, const WeightMap& wm, const std::vector<std::string>& names, int in, int out_each, LinearW* l) {
    const size_t esz = ltx_dt_size(m->dtype);
    l->in = in; l->out = out_each * (int)names.size(); l->b = nullptr;
    HIP_TRY(hipMalloc(&l->w, (size_t)l->out * in * esz)); m->owned.push_back(l->w);
    for (size_t i = 0; i < names.size(); ++i) {
        const ltx_weight* w = wm.find(names[i]);
        if (!w) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + names[i] + "'");
        LTX_TRY(ltx_upload_cast(w, (char*)l->w + i * (size_t)out_each * in * esz, m->dtype, (int64_t)out_each * in, names[i]));
    }
    return LTX_OK;
}

template <typename T>
int run_attn(const ltx_t5* m, const void* qkv, const float* bias, const float* kmask, void* out, int B, int S, hipStream_t s) {
    // The pipeline's case (bf16, d_kv 64, at most 128 tokens): the short-key-set MFMA kernel of the DiT's cross attention with
    // the relative position bias as its [heads, S, S] table - K and V of a head in LDS once, a 32-query unit per wave: 44.6 -> ~9 us
    // per layer at S = 128 (the kernel below gives a query to each WAVE and a key to each lane).  Option t5_attn_mfma=0: the kernel below.
    if constexpr (sizeof(T) == 2) {
        if (m->cfg.d_kv == 64 && ltx_attention_cross64_ok(64, S) && S % 4 == 0 && ltx_opt().t5_attn_mfma) {     // the launcher's own predicate
            const int inner = m->cfg.num_heads * 64;
            AttnArgs a;
            a.q = qkv; a.k = reinterpret_cast<const T*>(qkv) + inner; a.v = reinterpret_cast<const T*>(qkv) + 2 * inner; a.o = out;
            a.ldq = a.ldk = a.ldv = 3 * inner; a.ldo = inner;
            a.B = B; a.Sq = S; a.Sk = S; a.heads = m->cfg.num_heads; a.hd = 64; a.scale = 1.0f;       // T5 does not scale its scores
            a.bias = kmask; a.bias2d = bias;
            return ltx_launch_attention(a, LTX_DT_BF16, s);
        }
    }
    dim3 grid((unsigned)cdiv(S, 4), (unsigned)m->cfg.num_heads, (unsigned)B), block(256);
    switch (m->cfg.d_kv) {
        case 32: hipLaunchKernelGGL((t5_attn_kernel<T, 32>), grid, block, 0, s, (const T*)qkv, bias, kmask, (T*)out, S, m->cfg.num_heads); break;
        case 64: hipLaunchKernelGGL((t5_attn_kernel<T, 64>), grid, block, 0, s, (const T*)qkv, bias, kmask, (T*)out, S, m->cfg.num_heads); break;
        case 8: hipLaunchKernelGGL((t5_attn_kernel<T, 8>), grid, block, 0, s, (const T*)qkv, bias, kmask, (T*)out, S, m->cfg.num_heads); break;
        case 16: hipLaunchKernelGGL((t5_attn_kernel<T, 16>), grid, block, 0, s, (const T*)qkv, bias, kmask, (T*)out, S, m->cfg.num_heads); break;
        default: LTX_FAIL(LTX_ERR_UNSUPPORTED, "t5: d_kv must be 8, 16, 32 or 64");
    }
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

}  // namespace

extern "C" void ltx_t5_config_default(ltx_t5_config* c) {
    if (!c) return;
    c->vocab_size = 32128; c->d_model = 4096; c->d_kv = 64; c->d_ff = 10240; c->num_layers = 24; c->num_heads = 64;
    c->relative_attention_num_buckets = 32; c->relative_attention_max_distance = 128; c->layer_norm_epsilon = 1e-6f;
}

namespace {
// construction in three steps, so that a loader can hand the weights over group by group (ltx_t5_create_from_gguf streams
// one layer at a time through a small staging buffer) instead of holding the whole model twice
int t5_begin(const ltx_t5_config* cfg, ltx_dtype model_dtype, int device, std::unique_ptr<ltx_t5>* out) {
    const int D = cfg->d_model, inner = cfg->num_heads * cfg->d_kv;
    if (cfg->num_layers < 1 || D % 8 || inner % 8 || cfg->d_ff % 8 || cfg->relative_attention_num_buckets < 4)
        LTX_FAIL(LTX_ERR_ARG, "ltx_t5_create: d_model, heads*d_kv and d_ff must be multiples of 8");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<ltx_t5> m(new ltx_t5());
    m->cfg = *cfg; m->device = device; m->dtype = model_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32;
    m->layers.resize(cfg->num_layers);
    *out = std::move(m);
    return LTX_OK;
}
int t5_globals(ltx_t5* m, const WeightMap& wm) {
    const ltx_t5_config& c = m->cfg;
    LTX_TRY(own_tensor(m, wm, "shared.weight", (int64_t)c.vocab_size * c.d_model, m->dtype, &m->embed));
    void* rt = nullptr;
    LTX_TRY(own_tensor(m, wm, "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
                       (int64_t)c.relative_attention_num_buckets * c.num_heads, LTX_DT_F32, &rt));
    m->rel_table = reinterpret_cast<float*>(rt);
    LTX_TRY(own_tensor(m, wm, "encoder.final_layer_norm.weight", c.d_model, m->dtype, &m->final_ln));
    return LTX_OK;
}
int t5_layer(ltx_t5* m, int i, const WeightMap& wm) {
    const ltx_t5_config& c = m->cfg;
    const int D = c.d_model, inner = c.num_heads * c.d_kv;
    const std::string p = "encoder.block." + std::to_string(i) + ".layer.";
    ltx_t5::Layer& L = m->layers[i];
    LTX_TRY(own_fused(m, wm, {p + "0.SelfAttention.q.weight", p + "0.SelfAttention.k.weight", p + "0.SelfAttention.v.weight"}, D, inner, &L.qkv));
    LTX_TRY(own_fused(m, wm, {p + "0.SelfAttention.o.weight"}, inner, D, &L.o));
    LTX_TRY(own_tensor(m, wm, p + "0.layer_norm.weight", D, m->dtype, &L.ln0));
    LTX_TRY(own_fused(m, wm, {p + "1.DenseReluDense.wi_0.weight", p + "1.DenseReluDense.wi_1.weight"}, D, c.d_ff, &L.wi));
    LTX_TRY(own_fused(m, wm, {p + "1.DenseReluDense.wo.weight"}, c.d_ff, D, &L.wo));
    LTX_TRY(own_tensor(m, wm, p + "1.layer_norm.weight", D, m->dtype, &L.ln1));
    return LTX_OK;
}
}  // namespace

extern "C" int ltx_t5_create(const ltx_t5_config* cfg, const ltx_weight* weights, size_t n_weights,
                             ltx_dtype model_dtype, int device, ltx_t5** out) {
    if (!cfg || !weights || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_t5_create: null argument");
    *out = nullptr;
    std::unique_ptr<ltx_t5> m;
    LTX_TRY(t5_begin(cfg, model_dtype, device, &m));
    WeightMap wm(weights, n_weights);
    LTX_TRY(t5_globals(m.get(), wm));
    for (int i = 0; i < cfg->num_layers; ++i) LTX_TRY(t5_layer(m.get(), i, wm));
    *out = m.release();
    return LTX_OK;
}

extern "C" void ltx_t5_destroy(ltx_t5* m) { delete m; }

extern "C" int ltx_t5_forward(ltx_t5* m, const int32_t* input_ids, int B, int S, ltx_dtype out_dtype, void* out, ltx_stream stream) {
    return ltx_t5_forward_masked(m, input_ids, nullptr, B, S, out_dtype, out, stream);
}

// QuantizedT5EncoderModel::load_with_config (quantized_t5_encoder.rs:575-603): every tensor is dequantised to f32 on the device
// (QTensor::dequantize) and handed to the model under the Hugging Face name of the same weight.  The tensors stream through
// ONE staging buffer a layer at a time (the largest group, one T5-XXL layer, is 0.77 GB of f32) instead of the whole model
// sitting in f32 beside its final copy (19 GB in ~220 allocations).
extern "C" int ltx_t5_create_from_gguf(const ltx_t5_config* cfg, const char* gguf_path, ltx_dtype model_dtype, int device, ltx_t5** out) {
    if (!cfg || !gguf_path || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_t5_create_from_gguf: null argument");
    *out = nullptr;
    std::unique_ptr<ltx_t5> m;
    LTX_TRY(t5_begin(cfg, model_dtype, device, &m));
    ltx_gguf* g = nullptr;
    LTX_TRY(ltx_gguf_open(gguf_path, &g));
    struct Guard { ltx_gguf* g; void* stage = nullptr; size_t cap = 0; ~Guard() { if (stage) (void)hipFree(stage); ltx_gguf_close(g); } } guard{g};
    const int D = cfg->d_model, inner = cfg->num_heads * cfg->d_kv;
    // one group = the tensors of one construction step: described first (sizes), then dequantised into the staging buffer
    struct Item { std::string gguf_name, hf_name; int64_t d0, d1; int type; const void* data; int64_t numel; };
    auto build_group = [&](std::vector<Item>& items, int layer) -> int {
        size_t need = 0;
        for (Item& it : items) {
            const int idx = ltx_gguf_find(g, it.gguf_name.c_str());
            if (idx < 0) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "GGUF tensor '" + it.gguf_name + "' not found in '" + gguf_path + "'");
            const char* nm; int nd; const int64_t* shp; size_t nbytes;
            LTX_TRY(ltx_gguf_tensor(g, (size_t)idx, &nm, &it.type, &nd, &shp, &it.data, &nbytes));
            it.numel = 1; for (int i = 0; i < nd; ++i) it.numel *= shp[i];
            const bool ok = it.d1 ? (nd == 2 && shp[0] == it.d0 && shp[1] == it.d1) : (it.numel == it.d0);
            if (!ok) LTX_FAIL(LTX_ERR_ARG, "GGUF tensor '" + it.gguf_name + "': unexpected shape");
            need += ((size_t)it.numel * sizeof(float) + 255) & ~(size_t)255;
        }
        if (need > guard.cap) {
            if (guard.stage) { (void)hipFree(guard.stage); guard.stage = nullptr; guard.cap = 0; }
            HIP_TRY(hipMalloc(&guard.stage, need)); guard.cap = need;
        }
        std::vector<ltx_weight> ws(items.size());
        size_t off = 0;
        for (size_t k = 0; k < items.size(); ++k) {
            const Item& it = items[k];
            void* dev = (char*)guard.stage + off;
            off += ((size_t)it.numel * sizeof(float) + 255) & ~(size_t)255;
            LTX_TRY(ltx_gguf_dequantize(it.type, it.data, 0, it.numel, LTX_F32, dev, nullptr));
            ltx_weight w; memset(&w, 0, sizeof(w));
            w.name = it.hf_name.c_str(); w.data = dev; w.dtype = LTX_F32; w.ndim = it.d1 ? 2 : 1; w.shape[0] = it.d0; w.shape[1] = it.d1; w.on_device = 1;
            ws[k] = w;
        }
        WeightMap wm(ws.data(), ws.size());
        LTX_TRY(layer < 0 ? t5_globals(m.get(), wm) : t5_layer(m.get(), layer, wm));
        HIP_TRY(hipDeviceSynchronize());                     // the model's own copies are complete: the staging buffer is free again
        return LTX_OK;
    };
    {
        std::vector<Item> items = {
            {"token_embd.weight", "shared.weight", cfg->vocab_size, D, 0, nullptr, 0},
            {"enc.blk.0.attn_rel_b.weight", "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", cfg->relative_attention_num_buckets, cfg->num_heads, 0, nullptr, 0},
            {"enc.output_norm.weight", "encoder.final_layer_norm.weight", D, 0, 0, nullptr, 0}};
        LTX_TRY(build_group(items, -1));
    }
    for (int i = 0; i < cfg->num_layers; ++i) {
        const std::string gp = "enc.blk." + std::to_string(i) + ".", hp = "encoder.block." + std::to_string(i) + ".layer.";
        std::vector<Item> items = {
            {gp + "attn_q.weight", hp + "0.SelfAttention.q.weight", inner, D, 0, nullptr, 0},
            {gp + "attn_k.weight", hp + "0.SelfAttention.k.weight", inner, D, 0, nullptr, 0},
            {gp + "attn_v.weight", hp + "0.SelfAttention.v.weight", inner, D, 0, nullptr, 0},
            {gp + "attn_o.weight", hp + "0.SelfAttention.o.weight", D, inner, 0, nullptr, 0},
            {gp + "attn_norm.weight", hp + "0.layer_norm.weight", D, 0, 0, nullptr, 0},
            {gp + "ffn_gate.weight", hp + "1.DenseReluDense.wi_0.weight", cfg->d_ff, D, 0, nullptr, 0},      // gelu_new(gate(x)) * up(x), :437-449
            {gp + "ffn_up.weight", hp + "1.DenseReluDense.wi_1.weight", cfg->d_ff, D, 0, nullptr, 0},
            {gp + "ffn_down.weight", hp + "1.DenseReluDense.wo.weight", D, cfg->d_ff, 0, nullptr, 0},
            {gp + "ffn_norm.weight", hp + "1.layer_norm.weight", D, 0, 0, nullptr, 0}};
        LTX_TRY(build_group(items, i));
    }
    *out = m.release();
    return LTX_OK;
}

extern "C" int ltx_t5_forward_masked(ltx_t5* m, const int32_t* input_ids, const float* attention_mask, int B, int S,
                                     ltx_dtype out_dtype, void* out, ltx_stream stream) {
    if (!m || !input_ids || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_t5_forward: null argument");
    if (B < 1 || S < 1 || S > 512) LTX_FAIL(LTX_ERR_ARG, "ltx_t5_forward: need B >= 1 and 1 <= S <= 512");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const ltx_t5_config& c = m->cfg;
    const int D = c.d_model, H = c.num_heads, inner = H * c.d_kv, dt = m->dtype;
    const int64_t M = (int64_t)B * S; const size_t esz = ltx_dt_size(dt);
    LTX_TRY(m->ids.ensure(M * sizeof(int))); LTX_TRY(m->h.ensure(M * D * esz)); LTX_TRY(m->n.ensure(M * D * esz));
    LTX_TRY(m->qkv.ensure(M * 3 * inner * esz)); LTX_TRY(m->att.ensure(M * inner * esz));
    LTX_TRY(m->ff.ensure(M * 2 * c.d_ff * esz)); LTX_TRY(m->act.ensure(M * c.d_ff * esz));
    LTX_TRY(m->bias.ensure((size_t)H * S * S * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(m->ids.p, input_ids, M * sizeof(int), hipMemcpyHostToDevice, s));
    const float* kmask = nullptr;
    std::vector<float> kb;
    if (attention_mask) {                                            // (1 - mask) * -1e9 in f32, as the reference builds it (:624-634)
        kb.resize((size_t)M);
        for (int64_t i = 0; i < M; ++i) kb[(size_t)i] = (1.0f - attention_mask[i]) * -1e9f;
        LTX_TRY(m->kmask.ensure(M * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(m->kmask.p, kb.data(), M * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                            // kb is a stack-lifetime staging buffer
        kmask = m->kmask.as<float>();
    }
    const int blocks = (int)((M * D + 255) / 256 > 4096 ? 4096 : (M * D + 255) / 256);
    if (dt == LTX_DT_BF16) hipLaunchKernelGGL((t5_embed_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, m->ids.as<int>(), (const bf16_t*)m->embed, m->h.as<bf16_t>(), M, D, c.vocab_size);
    else hipLaunchKernelGGL((t5_embed_kernel<float>), dim3(blocks), dim3(256), 0, s, m->ids.as<int>(), (const float*)m->embed, m->h.as<float>(), M, D, c.vocab_size);
    LTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(t5_bias_kernel, dim3((unsigned)(((int64_t)H * S * S + 255) / 256)), dim3(256), 0, s, m->rel_table, m->bias.as<float>(), H, S,
                       c.relative_attention_num_buckets, c.relative_attention_max_distance);
    LTX_CHECK_LAUNCH();
    RowNormArgs rn; rn.rows = M; rn.D = D; rn.ldx = D; rn.ldy = D; rn.kind = 0; rn.eps = c.layer_norm_epsilon; rn.rows_per_batch = M;
    for (int i = 0; i < c.num_layers; ++i) {
        const ltx_t5::Layer& L = m->layers[i];
        rn.x = m->h.p; rn.y = m->n.p; rn.weight = L.ln0;
        LTX_TRY(ltx_launch_rownorm(rn, dt, s));
        LTX_TRY(ltx_linear(L.qkv, m->n.p, D, m->qkv.p, 3 * inner, (int)M, dt, EPI_BIAS, s));
        if (dt == LTX_DT_BF16) LTX_TRY(run_attn<bf16_t>(m, m->qkv.p, m->bias.as<float>(), kmask, m->att.p, B, S, s));
        else LTX_TRY(run_attn<float>(m, m->qkv.p, m->bias.as<float>(), kmask, m->att.p, B, S, s));
        LTX_TRY(ltx_linear(L.o, m->att.p, inner, m->h.p, D, (int)M, dt, EPI_RESID, s, m->h.p, D));
        rn.x = m->h.p; rn.y = m->n.p; rn.weight = L.ln1;
        LTX_TRY(ltx_launch_rownorm(rn, dt, s));
        LTX_TRY(ltx_linear(L.wi, m->n.p, D, m->ff.p, 2 * c.d_ff, (int)M, dt, EPI_BIAS, s));
        const int gb = (int)((M * c.d_ff + 255) / 256 > 8192 ? 8192 : (M * c.d_ff + 255) / 256);
        if (dt == LTX_DT_BF16) hipLaunchKernelGGL((t5_gate_kernel<bf16_t>), dim3(gb), dim3(256), 0, s, m->ff.as<bf16_t>(), m->act.as<bf16_t>(), M, c.d_ff);
        else hipLaunchKernelGGL((t5_gate_kernel<float>), dim3(gb), dim3(256), 0, s, m->ff.as<float>(), m->act.as<float>(), M, c.d_ff);
        LTX_CHECK_LAUNCH();
        LTX_TRY(ltx_linear(L.wo, m->act.p, c.d_ff, m->h.p, D, (int)M, dt, EPI_RESID, s, m->h.p, D));
    }
    rn.x = m->h.p; rn.y = m->n.p; rn.weight = m->final_ln;
    LTX_TRY(ltx_launch_rownorm(rn, dt, s));
    return ltx_launch_cast(m->n.p, dt, out, out_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32, M * D, s);
}
