// AutoencoderKLLtxVideo::decode on MI355X (decoder side only; T2V never encodes).
// Reference: src/models/ltx_video/vae.rs  (:1488-1727 decoder, :755-821 resnet, :1090-1169 upsampler,
// :415-464 conv, :2037-2066 decode_z, :2225-2290 / :2358-2434 tiling, :1927-2006 blends).
//
// HBM layout: activations are CHANNELS-LAST [B,T,H,W,C] in the model dtype, so that
//   * the conv3d is an implicit GEMM whose K dimension (C_in) is contiguous (16-B coalesced loads,
//     MFMA operands straight from LDS rows), 27 taps accumulated in the same accumulators;
//   * the per-voxel RMS norm over C is a contiguous row reduction;
//   * depth-to-space (+ first-frame drop + tiled residual) and unpatchify are pure index math in
//     the conv epilogues (weights' output channels are re-ordered once at load time so the 4
//     accumulator values a lane owns are contiguous in the destination).
// The packed token layout the denoise loop carries ([B, F*H*W, 128]) IS channels-last, so
// unpack_latents is free on this path.
#include <cstring>
#include <deque>
#include "model_util.h"
#include "options.h"
#include "../../include/ltxhip_frames.h"

extern "C" int ltx_pcg32_randn(uint64_t seed, uint64_t inc, size_t n, float* out_host);   // host/pipeline.hip (utils/deterministic_rng.rs)

struct ConvW {
    void* w = nullptr;   // [27][N][Cin] model dtype (N possibly permuted)
    void* b = nullptr;   // [N]
    int cin = 0, cout = 0;
    int nsub = 8;        // depth-to-space convs: output channels per final channel, 8 = (2, 2, 2), 4 = (1, 2, 2)
};
struct TimeEmbW { LinearW l1, l2; int dim = 0; };
struct ResnetW { ConvW c1, c2; void* sst = nullptr; void *pcs1 = nullptr, *pcs2 = nullptr; };   // pcs: per_channel_scale{1,2} [C] where the block injects noise (vae.rs:676-689)
struct UpBlockW { ConvW up; TimeEmbW te; std::vector<ResnetW> res; int ch = 0, cin = 0; bool residual = true, temporal = true; };

struct ltx_vae {
    ltx_vae_config cfg{};
    int dtype = LTX_DT_BF16, device = 0;
    ConvW conv_in, conv_out;
    TimeEmbW mid_te, out_te;
    std::vector<ResnetW> mid;
    std::vector<UpBlockW> ups;
    void* sst_out = nullptr;
    float tsm = 1.0f; bool has_tsm = false;
    float *mean = nullptr, *std_ = nullptr, *vtab = nullptr;
    int mid_ch = 0, last_ch = 0;
    std::vector<void*> owned;
    DevBuf zin, X, Y, N, C, tproj, e1, te, mod, tiles[2], tile_lat, stats, predec, rgbtmp;
    std::deque<DevBuf> tilebufs;   // deque: growing it must not move DevBufs that Tile.buf points at
    // CombinedTimestepEmbedder outputs (+ scale_shift_table) per (table, timestep vector, stream): functions of the weights and the
    // timestep alone (a pipeline decodes every video at the same decode_timestep), 5 launches per resnet when recomputed
    struct ModEntry { const void* sst = nullptr; float t[LTX_MAX_BATCH] = {0}; int n = 0; hipStream_t stream = nullptr; uint64_t used = 0; DevBuf buf; };
    std::deque<ModEntry> mods; uint64_t mod_clock = 0;
    // noise injection (vae.rs:741-753): the reference draws an [H, W] plane from the device RNG per injection; here plane k of the
    // handle's life is Pcg32::new(noise_seed, k).randn(H * W) (utils/deterministic_rng.rs), k restarted by ltx_vae_set_noise_seed
    uint64_t noise_seed = 0, noise_ctr = 0; bool any_inject = false;
    DevBuf noise_plane; std::vector<float> noise_host;
    size_t act_bytes_reserved() const { return X.bytes + Y.bytes + N.bytes + C.bytes; }   // activation buffers a decoder call re-uses
    void free_all() {
        for (void* p : owned) if (p) (void)hipFree(p);
        owned.clear();
        DevBuf* bs[] = {&zin, &X, &Y, &N, &C, &tproj, &e1, &te, &mod, &tiles[0], &tiles[1], &tile_lat, &stats, &predec, &noise_plane};
        for (DevBuf* b : bs) b->release();
        for (auto& b : tilebufs) b.release();
        for (auto& e : mods) e.buf.release();
        mods.clear();
    }
};

namespace {

enum { PERM_NONE = LTX_PERM_NONE, PERM_D2S = LTX_PERM_D2S, PERM_UNPATCH = LTX_PERM_UNPATCH };

// dst[tap][n'][i] = src[o][i][tap] ; bias'[n'] = bias[o]
__global__ void pack_conv_kernel(const void* src, int sdt, void* dst, int ddt, int O, int I, int ntaps, int mode, int Cf, int nsub) {
    const int64_t n = (int64_t)O * I * ntaps;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(idx % I); int64_t r = idx / I; int np = (int)(r % O); int tap = (int)(r / O);
        int o = np;
        if (mode == PERM_D2S) { int s = np / Cf, c = np - s * Cf; o = c * nsub + s; }                 // n' = s*Cf + c' (nsub = 8: (2,2,2), 4: (1,2,2))
        else if (mode == PERM_UNPATCH) { int c = np >> 4, oh = (np >> 2) & 3, ow = np & 3; o = (c * 4 + ow) * 4 + oh; }  // n' = (c*4+oh)*4+ow
        float v = sdt == LTX_DT_BF16 ? (float)reinterpret_cast<const bf16_t*>(src)[((int64_t)o * I + i) * ntaps + tap]
                                     : reinterpret_cast<const float*>(src)[((int64_t)o * I + i) * ntaps + tap];
        if (ddt == LTX_DT_BF16) reinterpret_cast<bf16_t*>(dst)[idx] = (bf16_t)v; else reinterpret_cast<float*>(dst)[idx] = v;
    }
}

// crop a [t0:t1, h0:h1, w0:w1] window of a channels-last tensor (elements of esz bytes)
__global__ void cl_window_kernel(const unsigned char* src, unsigned char* dst, int esz, int B, int T, int H, int W, int C,
                                 int t0, int t1, int h0, int h1, int w0, int w1) {
    const int nt = t1 - t0, nh = h1 - h0, nw = w1 - w0;
    const int64_t rowb = (int64_t)C * esz;
    const int64_t n = (int64_t)B * nt * nh * nw * rowb;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t bb = idx % rowb; int64_t r = idx / rowb;
        int w = (int)(r % nw); r /= nw; int h = (int)(r % nh); r /= nh; int t = (int)(r % nt); int b = (int)(r / nt);
        dst[idx] = src[((((int64_t)b * T + t0 + t) * H + h0 + h) * W + w0 + w) * rowb + bb];
    }
}

}  // namespace
int ltx_pack_conv(const void* src_dev, int sdt, void* dst, int ddt, int O, int I, int ntaps, int mode, int Cf, hipStream_t s) {
    const int nsub = (mode == PERM_D2S && Cf > 0) ? O / Cf : 8;
    int64_t n = (int64_t)O * I * ntaps; int64_t blocks = cdiv64(n, 256); if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src_dev, sdt, dst, ddt, O, I, ntaps, mode, Cf, nsub);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
namespace {
int load_conv(ltx_vae* v, const WeightMap& wm, const std::string& prefix, int cin, int cout, int mode, int Cf, ConvW* c) {
    const int dt = v->dtype; const size_t esz = ltx_dt_size(dt);
    c->cin = cin; c->cout = cout;
    const int nsub = (mode == PERM_D2S && Cf > 0) ? cout / Cf : 8;
    c->nsub = nsub;
    const ltx_weight* w = wm.find(prefix + ".conv.weight");
    const ltx_weight* b = wm.find(prefix + ".conv.bias");
    if (!w) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + prefix + ".conv.weight'");
    if (!b) LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + prefix + ".conv.bias'");
    if (ltx_numel(w) != (int64_t)cout * cin * 27) LTX_FAIL(LTX_ERR_ARG, "weight '" + prefix + ".conv.weight': wrong size");
    if (ltx_numel(b) != cout) LTX_FAIL(LTX_ERR_ARG, "weight '" + prefix + ".conv.bias': wrong size");
    HIP_TRY(hipMalloc(&c->w, (size_t)cout * cin * 27 * esz)); v->owned.push_back(c->w);
    HIP_TRY(hipMalloc(&c->b, (size_t)cout * esz + 16)); v->owned.push_back(c->b);
    const void* src = nullptr; void* tmp = nullptr;
    LTX_TRY(ltx_stage_src(w, &src, &tmp));
    int64_t n = (int64_t)cout * cin * 27; int64_t blocks = cdiv64(n, 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, src, w->dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32,
                       c->w, dt, cout, cin, 27, mode, Cf, nsub);
    hipError_t e = hipDeviceSynchronize(); if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) { ltx_set_error(std::string("pack conv: ") + hipGetErrorString(e)); return LTX_ERR_HIP; }
    LTX_TRY(ltx_stage_src(b, &src, &tmp));
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)cdiv(cout, 256)), dim3(256), 0, 0, src, b->dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32,
                       c->b, dt, cout, 1, 1, mode, Cf, nsub);
    e = hipDeviceSynchronize(); if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) { ltx_set_error(std::string("pack bias: ") + hipGetErrorString(e)); return LTX_ERR_HIP; }
    return LTX_OK;
}

int load_temb(ltx_vae* v, const WeightMap& wm, const std::string& prefix, int dim, TimeEmbW* t) {
    t->dim = dim;
    LTX_TRY(ltx_load_linear(wm, prefix + ".timestep_embedder.linear_1", 256, dim, v->dtype, &t->l1));
    v->owned.push_back(t->l1.w); if (t->l1.b) v->owned.push_back(t->l1.b);
    LTX_TRY(ltx_load_linear(wm, prefix + ".timestep_embedder.linear_2", dim, dim, v->dtype, &t->l2));
    v->owned.push_back(t->l2.w); if (t->l2.b) v->owned.push_back(t->l2.b);
    return LTX_OK;
}

int load_resnet(ltx_vae* v, const WeightMap& wm, const std::string& prefix, int ch, bool inject, ResnetW* r) {
    LTX_TRY(load_conv(v, wm, prefix + ".conv1", ch, ch, PERM_NONE, 0, &r->c1));
    LTX_TRY(load_conv(v, wm, prefix + ".conv2", ch, ch, PERM_NONE, 0, &r->c2));
    if (inject) {
        // vae.rs:676-689: vb.pp("per_channel_scaleN").get((C, 1, 1), "weight").ok() - a scale that is absent from the checkpoint
        // under exactly that name is no injection, not an error
        for (int k = 0; k < 2; ++k) {
            const std::string nm = prefix + ".per_channel_scale" + std::to_string(k + 1) + ".weight";
            if (!wm.find(nm)) continue;
            void** dst = k ? &r->pcs2 : &r->pcs1;
            LTX_TRY(ltx_load_tensor(wm, nm, ch, v->dtype, dst)); v->owned.push_back(*dst);
            v->any_inject = true;
        }
    }
    if (v->cfg.timestep_conditioning) {
        LTX_TRY(ltx_load_tensor(wm, prefix + ".scale_shift_table", 4 * (int64_t)ch, v->dtype, &r->sst));
        v->owned.push_back(r->sst);
    }
    return LTX_OK;
}

int build(ltx_vae* v, const ltx_weight* weights, size_t n_weights) {
    const ltx_vae_config& c = v->cfg;
    WeightMap wm0(weights, n_weights);
    // accept both "decoder.xxx" and bare "xxx" names
    std::vector<ltx_weight> renamed; std::vector<std::string> names;
    names.reserve(n_weights); renamed.reserve(n_weights);
    for (size_t i = 0; i < n_weights; ++i) {
        std::string nm = weights[i].name ? weights[i].name : "";
        if (nm.rfind("decoder.", 0) == 0) nm = nm.substr(8);
        names.push_back(nm);
    }
    for (size_t i = 0; i < n_weights; ++i) { ltx_weight w = weights[i]; w.name = names[i].c_str(); renamed.push_back(w); }
    WeightMap wm(renamed.data(), renamed.size());

    const int nb = c.n_blocks;
    std::vector<int> boc(nb), upf(nb), lpb(nb + 1);
    for (int i = 0; i < nb; ++i) { boc[i] = c.decoder_block_out_channels[nb - 1 - i]; upf[i] = c.decoder_upsample_factor[nb - 1 - i]; }
    for (int i = 0; i <= nb; ++i) lpb[i] = c.decoder_layers_per_block[nb - i];
    v->mid_ch = boc[0];
    LTX_TRY(load_conv(v, wm, "conv_in", c.latent_channels, v->mid_ch, PERM_NONE, 0, &v->conv_in));
    if (c.timestep_conditioning) LTX_TRY(load_temb(v, wm, "mid_block.time_embedder", 4 * v->mid_ch, &v->mid_te));
    v->mid.resize(lpb[0]);
    for (int i = 0; i < lpb[0]; ++i) LTX_TRY(load_resnet(v, wm, "mid_block.resnets." + std::to_string(i), v->mid_ch, c.decoder_inject_noise[nb] != 0, &v->mid[i]));   // inject list reversed: entry nb is the mid block (vae.rs:1514-1515, 1541)
    v->ups.resize(nb);
    int cur = v->mid_ch;
    for (int bi = 0; bi < nb; ++bi) {
        UpBlockW& u = v->ups[bi];
        const std::string p = "up_blocks." + std::to_string(bi);
        u.residual = c.decoder_upsample_residual[nb - 1 - bi] != 0;   // reversed like the other lists (vae.rs:1516-1517)
        u.ch = boc[bi] / upf[bi];                       // vae.rs:1548
        u.cin = u.ch * upf[bi];                         // upsampler in-channels (vae.rs:1215)
        if (u.cin != cur) LTX_FAIL(LTX_ERR_UNSUPPORTED, "decoder up-block channel chain mismatch");
        u.temporal = c.decoder_spatiotemporal_scaling[nb - 1 - bi] != 0;   // reversed too (vae.rs:1509-1510); false: the (1, 2, 2) upsampler (:1225-1236)
        const int nsub = u.temporal ? 8 : 4;
        if ((u.ch * nsub) % u.cin != 0 || u.cin % nsub != 0) LTX_FAIL(LTX_ERR_UNSUPPORTED, "upsampler residual repeat must be integral");
        LTX_TRY(load_conv(v, wm, p + ".upsamplers.0.conv", u.cin, u.ch * nsub, PERM_D2S, u.ch, &u.up));
        if (c.timestep_conditioning) LTX_TRY(load_temb(v, wm, p + ".time_embedder", 4 * u.ch, &u.te));
        u.res.resize(lpb[bi + 1]);
        for (int i = 0; i < lpb[bi + 1]; ++i) LTX_TRY(load_resnet(v, wm, p + ".resnets." + std::to_string(i), u.ch, c.decoder_inject_noise[nb - 1 - bi] != 0, &u.res[i]));   // inj[bi + 1] of the reversed list (vae.rs:1561)
        cur = u.ch;
    }
    v->last_ch = cur;
    const int nout = c.out_channels * c.patch_size * c.patch_size;
    LTX_TRY(load_conv(v, wm, "conv_out", cur, nout, PERM_UNPATCH, 0, &v->conv_out));
    if (c.timestep_conditioning) {
        LTX_TRY(load_temb(v, wm, "time_embedder", 2 * cur, &v->out_te));
        LTX_TRY(ltx_load_tensor(wm, "scale_shift_table", 2 * (int64_t)cur, v->dtype, &v->sst_out)); v->owned.push_back(v->sst_out);
        const ltx_weight* t = wm.find("timestep_scale_multiplier");
        if (t) {
            void* d = nullptr;
            LTX_TRY(ltx_load_tensor(wm, "timestep_scale_multiplier", 1, LTX_DT_F32, &d));
            HIP_TRY(hipMemcpy(&v->tsm, d, sizeof(float), hipMemcpyDeviceToHost)); (void)hipFree(d);
            v->has_tsm = true;
        }
    }
    // latents_mean / latents_std (vae.rs:1827-1838): from weights when present, else 0 / 1
    {
        const int C = c.latent_channels;
        std::vector<float> z(C, 0.f), o(C, 1.f);
        HIP_TRY(hipMalloc((void**)&v->mean, sizeof(float) * C)); v->owned.push_back(v->mean);
        HIP_TRY(hipMalloc((void**)&v->std_, sizeof(float) * C)); v->owned.push_back(v->std_);
        const ltx_weight* lm = wm0.find("latents_mean"); const ltx_weight* ls = wm0.find("latents_std");
        if (lm) LTX_TRY(ltx_upload_cast(lm, v->mean, LTX_DT_F32, C, "latents_mean")); else HIP_TRY(hipMemcpy(v->mean, z.data(), sizeof(float) * C, hipMemcpyHostToDevice));
        if (ls) LTX_TRY(ltx_upload_cast(ls, v->std_, LTX_DT_F32, C, "latents_std")); else HIP_TRY(hipMemcpy(v->std_, o.data(), sizeof(float) * C, hipMemcpyHostToDevice));
    }
    // sinusoid table exp(-ln(1e4) * i / 128) with the reference's f32 roundings (vae.rs:172-198)
    {
        std::vector<float> tab(128);
        ltx_sinusoid_table(1, tab.data());
        HIP_TRY(hipMalloc((void**)&v->vtab, sizeof(float) * 128)); v->owned.push_back(v->vtab);
        HIP_TRY(hipMemcpy(v->vtab, tab.data(), sizeof(float) * 128, hipMemcpyHostToDevice));
    }
    return LTX_OK;
}

struct Dims { int B, T, H, W; int64_t vox() const { return (int64_t)B * T * H * W; } };

struct PostNorm { int on = 0; float eps = 0.f; int act = 0; int mod_stride = 0; const float* scale = nullptr; const float* shift = nullptr; };

GemmArgs conv_args(ltx_vae* v, const ConvW& cw, const Dims& d) {
    GemmArgs g;
    g.W = cw.w; g.bias = cw.b;
    g.M = (int)d.vox(); g.N = cw.cout; g.K = cw.cin; g.ldc = cw.cout; g.ldr = cw.cout;
    g.conv = 1; g.B = d.B; g.T = d.T; g.H = d.H; g.Wd = d.W; g.Cin = cw.cin;
    g.ntaps = 27; g.kh = 3; g.kw = 3;
    g.pad_t = v->cfg.decoder_causal ? 2 : 1;
    return g;
}

// Samples per conv launch: the fast conv kernels address their operands with 32-bit byte offsets (< 2 GiB per operand).
// Samples never interact in a conv, so a batch whose activations pass that limit (the batched leaf tiles of a tiled decode
// at the last stages) runs as sample groups that fit; a single sample beyond the limit goes to the launcher as it is.
int conv_chunk(ltx_vae* v, const ConvW& cw, const Dims& d) {
    if (d.B <= 1) return d.B;
    {   // the large-tile conv kernels address a window around each tile, not the tensor: whole batch in one launch where they apply
        GemmArgs g; g.M = (int)std::min<int64_t>(d.vox(), 2147483647); g.N = cw.cout; g.K = cw.cin; g.conv = 1; g.B = d.B; g.T = d.T; g.H = d.H; g.Wd = d.W;
        g.Cin = cw.cin; g.ntaps = 27; g.kh = 3; g.kw = 3;
        if (v->dtype == LTX_DT_BF16 && d.vox() < 2147483647 && ltx_gemm_big_eligible(g, v->dtype)) return d.B;
    }
    const int64_t per = (int64_t)d.T * d.H * d.W;
    const int64_t lim = (2147483648LL - (1 << 20)) / (per * std::max(cw.cin, cw.cout) * (int64_t)ltx_dt_size(v->dtype));
    return (lim >= 1 && lim < d.B) ? (int)lim : d.B;
}

// whether conv1 of a resnet can carry norm2 in its epilogue: bf16, the halo-staged kernel with BN == channels
bool fuse_norm2(ltx_vae* v, const ConvW& cw, const Dims& d_all, int ch) {
    Dims d = d_all; d.B = conv_chunk(v, cw, d_all);
    const LtxOptions& o = ltx_opt();
    if (!o.vae_fuse_norm || !o.gemm_wide_epi || (o.gemm_off & (LTX_FAM_HALO | LTX_FAM_BIG))) return false;
    if (v->dtype != LTX_DT_BF16 || (ch != 128 && ch != 256) || cw.cout != ch) return false;
    const GemmArgs g = conv_args(v, cw, d);
    // the fused epilogue needs the whole channel row in one tile (BN == channels), i.e. a grid of M / 256 blocks: below about
    // one round of the chip the unfused conv on a plan with more, smaller tiles + the stand-alone norm is faster (C1's
    // 256-channel stage, 78 tiles: 168 us fused vs 108 + 13; decode 5.7 -> 5.45 ms, profiles/r5k_c1_vae_fused_norm_ab.jsonl)
    if ((g.M + 255) / 256 < 192 && o.vae_fuse_norm < 2) return false;
    return ltx_gemm_big_eligible(g, v->dtype) && ltx_conv_halo_eligible(g, EPI_BIAS, ch);
}

int conv3d(ltx_vae* v, const ConvW& cw, const void* x, void* y, const Dims& d, int epi, const void* resid, int post, hipStream_t s, const PostNorm* pn = nullptr) {
    const size_t esz = ltx_dt_size(v->dtype);
    const int64_t per = (int64_t)d.T * d.H * d.W;
    const int nb = conv_chunk(v, cw, d);
    size_t out_b, res_b = 0;                              // bytes per sample of the output / residual tensor
    const int To = cw.nsub == 4 ? d.T : 2 * d.T - 1;      // frames a depth-to-space conv writes
    if (epi == EPI_D2S) { out_b = (size_t)To * (2 * d.H) * (2 * d.W) * (cw.cout / cw.nsub) * esz; res_b = (size_t)per * cw.cin * esz; }
    else if (epi == EPI_UNPATCH) out_b = (size_t)(cw.cout / 16) * d.T * (4 * d.H) * (4 * d.W) * sizeof(float);
    else { out_b = (size_t)per * cw.cout * esz; res_b = out_b; }
    for (int b0 = 0; b0 < d.B; b0 += nb) {
        const int bc = std::min(nb, d.B - b0);
        GemmArgs g;
        if (pn && pn->on) {
            g.pn_on = 1; g.pn_eps = pn->eps; g.pn_act = pn->act; g.pn_mod_stride = pn->mod_stride;
            g.pn_scale = pn->scale ? pn->scale + (size_t)b0 * pn->mod_stride : nullptr; g.pn_shift = pn->shift ? pn->shift + (size_t)b0 * pn->mod_stride : nullptr;
        }
        g.A = (const char*)x + (size_t)b0 * per * cw.cin * esz; g.W = cw.w; g.C = (char*)y + (size_t)b0 * out_b; g.bias = cw.b;
        g.resid = resid ? (const char*)resid + (size_t)b0 * res_b : nullptr;
        g.M = (int)(bc * per); g.N = cw.cout; g.K = cw.cin; g.ldc = cw.cout; g.ldr = cw.cout;
        g.conv = 1; g.B = bc; g.T = d.T; g.H = d.H; g.Wd = d.W; g.Cin = cw.cin;
        g.ntaps = 27; g.kh = 3; g.kw = 3;
        g.pad_t = v->cfg.decoder_causal ? 2 : 1;          // vae.rs:383-412
        g.post = post;
        if (epi == EPI_D2S) { g.Cf = cw.cout / cw.nsub; g.Cr = cw.cin / cw.nsub; g.To = To; g.Ho = 2 * d.H; g.Wo = 2 * d.W; g.d2s_sp = cw.nsub == 4; }
        LTX_TRY(ltx_launch_gemm(g, v->dtype, epi, s));
    }
    return LTX_OK;
}

// CombinedTimestepEmbedder (vae.rs:236-265) + "+ scale_shift_table" -> f32 [B][rows][C]; *out points at the (cached) result
int time_mod(ltx_vae* v, const TimeEmbW& te, const void* sst, const TimeVec& tv, const float** out, hipStream_t s) {
    const int dt = v->dtype; const int B = tv.n;
    ltx_vae::ModEntry* e = nullptr;
    for (auto& c : v->mods) if (c.sst == sst && c.n == B && c.stream == s && !memcmp(c.t, tv.t, sizeof(float) * B)) e = &c;
    if (!e) {
        constexpr size_t kMax = 256;                       // entries (a decoder has ~21 tables; tiled decodes add leaf-batch sizes)
        if (v->mods.size() < kMax) { v->mods.emplace_back(); e = &v->mods.back(); }
        else { e = &v->mods.front(); for (auto& c : v->mods) if (c.used < e->used) e = &c; }
        e->sst = nullptr;
        LTX_TRY(e->buf.ensure((size_t)B * te.dim * sizeof(float)));
        LTX_TRY(ltx_launch_sinusoid(v->tproj.p, dt, tv, v->vtab, 128, dt == LTX_DT_BF16, v->has_tsm ? v->tsm : 1.0f, s));
        LTX_TRY(ltx_linear(te.l1, v->tproj.p, 256, v->e1.p, te.dim, B, dt, EPI_BIAS, s));
        LTX_TRY(ltx_launch_silu(v->e1.p, v->e1.p, (int64_t)B * te.dim, dt, s));
        LTX_TRY(ltx_linear(te.l2, v->e1.p, te.dim, v->te.p, te.dim, B, dt, EPI_BIAS, s));
        LTX_TRY(ltx_launch_ada(e->buf.as<float>(), sst, v->te.p, 1, B, te.dim, dt, s));
        e->sst = sst; e->n = B; e->stream = s; memcpy(e->t, tv.t, sizeof(float) * B);
    }
    e->used = ++v->mod_clock;
    *out = e->buf.as<float>();
    return LTX_OK;
}

// one injection: the next plane of the handle's noise stream, y = x + plane[h, w] * scale[c] (+ shortcut)
int inject_noise(ltx_vae* v, const void* x, void* y, const void* scale, const void* shortcut, const Dims& d, int ch, hipStream_t s) {
    const int64_t hw = (int64_t)d.H * d.W;
    HIP_TRY(hipStreamSynchronize(s));                       // the previous injection's copy may still be reading the (pageable) host image, its kernel the plane
    v->noise_host.resize((size_t)hw);
    LTX_TRY(ltx_pcg32_randn(v->noise_seed, v->noise_ctr++, (size_t)hw, v->noise_host.data()));
    LTX_TRY(v->noise_plane.ensure((size_t)hw * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(v->noise_plane.p, v->noise_host.data(), (size_t)hw * sizeof(float), hipMemcpyHostToDevice, s));
    return ltx_launch_noise_inject(x, y, v->noise_plane.as<float>(), scale, shortcut, d.vox(), ch, hw, v->dtype, s);
}

int resnet(ltx_vae* v, const ResnetW& r, const TimeEmbW& te, int ch, const Dims& d, const TimeVec* tv, hipStream_t s) {
    const int dt = v->dtype;
    const float* mod = nullptr;
    if (tv && r.sst) LTX_TRY(time_mod(v, te, r.sst, *tv, &mod, s));
    RowNormArgs rn; rn.x = v->X.p; rn.y = v->N.p; rn.rows = d.vox(); rn.D = ch; rn.ldx = ch; rn.ldy = ch;
    rn.kind = 0; rn.eps = 1e-8f; rn.act = 1; rn.rows_per_batch = (int64_t)d.T * d.H * d.W; rn.mod_stride = 4 * ch;
    if (mod) { rn.shift = mod; rn.scale = mod + ch; }
    LTX_TRY(ltx_launch_rownorm(rn, dt, s));
    // norm2 + modulation + SiLU inside conv1's epilogue where one conv tile spans all channels (128 / 256): no pass of
    // its own over the stage's largest tensor
    if (!r.pcs1 && !r.pcs2 && fuse_norm2(v, r.c1, d, ch)) {
        PostNorm pn; pn.on = 1; pn.eps = rn.eps; pn.act = 1; pn.mod_stride = 4 * ch;
        if (mod) { pn.shift = mod + 2 * ch; pn.scale = mod + 3 * ch; }
        LTX_TRY(conv3d(v, r.c1, v->N.p, v->C.p, d, EPI_BIAS, nullptr, 0, s, &pn));
        LTX_TRY(conv3d(v, r.c2, v->C.p, v->X.p, d, EPI_RESID, v->X.p, 0, s));   // X = conv2(..) + X, in place
        return LTX_OK;
    }
    LTX_TRY(conv3d(v, r.c1, v->N.p, v->C.p, d, EPI_BIAS, nullptr, 0, s));
    if (r.pcs1) LTX_TRY(inject_noise(v, v->C.p, v->C.p, r.pcs1, nullptr, d, ch, s));        // vae.rs:784
    rn.x = v->C.p;
    if (mod) { rn.shift = mod + 2 * ch; rn.scale = mod + 3 * ch; }
    LTX_TRY(ltx_launch_rownorm(rn, dt, s));
    if (r.pcs2) {                                                                           // vae.rs:807-819: h = conv2(h); h += noise * scale; h + x
        LTX_TRY(conv3d(v, r.c2, v->N.p, v->C.p, d, EPI_BIAS, nullptr, 0, s));
        return inject_noise(v, v->C.p, v->X.p, r.pcs2, v->X.p, d, ch, s);
    }
    LTX_TRY(conv3d(v, r.c2, v->N.p, v->X.p, d, EPI_RESID, v->X.p, 0, s));   // X = conv2(..) + X, in place
    return LTX_OK;
}

// LtxVideoDecoder3d::forward on a channels-last latent `z` [B,F,H,W,Clat] (model dtype) -> f32 NCTHW
int decoder_forward(ltx_vae* v, const void* z, int B, int F, int H, int W, const TimeVec* tv, int post, float* out, hipStream_t s) {
    const ltx_vae_config& c = v->cfg;
    const int dt = v->dtype; const size_t esz = ltx_dt_size(dt);
    // workspace sizing: the largest activation is the last stage
    Dims d{B, F, H, W};
    int64_t max_elems = d.vox() * v->mid_ch;
    {
        Dims q = d;
        for (auto& u : v->ups) { if (u.temporal) q.T = 2 * q.T - 1; q.H *= 2; q.W *= 2; int64_t e = q.vox() * u.ch; if (e > max_elems) max_elems = e; }
    }
    LTX_TRY(v->X.ensure(max_elems * esz)); LTX_TRY(v->Y.ensure(max_elems * esz));
    LTX_TRY(v->N.ensure(max_elems * esz)); LTX_TRY(v->C.ensure(max_elems * esz));
    const int maxdim = 4 * v->mid_ch;
    LTX_TRY(v->tproj.ensure((size_t)B * 256 * esz)); LTX_TRY(v->e1.ensure((size_t)B * maxdim * esz));
    LTX_TRY(v->te.ensure((size_t)B * maxdim * esz)); LTX_TRY(v->mod.ensure((size_t)B * maxdim * sizeof(float)));
    const TimeVec* tvc = (tv && c.timestep_conditioning) ? tv : nullptr;

    LTX_TRY(conv3d(v, v->conv_in, z, v->X.p, d, EPI_BIAS, nullptr, 0, s));
    for (auto& r : v->mid) LTX_TRY(resnet(v, r, v->mid_te, v->mid_ch, d, tvc, s));
    for (auto& u : v->ups) {
        LTX_TRY(conv3d(v, u.up, v->X.p, v->Y.p, d, EPI_D2S, u.residual ? v->X.p : nullptr, 0, s));   // vae.rs:1164-1168
        std::swap(v->X, v->Y);
        if (u.temporal) d.T = 2 * d.T - 1;
        d.H *= 2; d.W *= 2;
        for (auto& r : u.res) LTX_TRY(resnet(v, r, u.te, u.ch, d, tvc, s));
    }
    // norm_out + global scale/shift + SiLU (vae.rs:1687-1723), conv_out + unpatchify
    const int ch = v->last_ch;
    const float* mod = nullptr;
    if (tvc && v->sst_out) LTX_TRY(time_mod(v, v->out_te, v->sst_out, *tvc, &mod, s));
    RowNormArgs rn; rn.x = v->X.p; rn.y = v->N.p; rn.rows = d.vox(); rn.D = ch; rn.ldx = ch; rn.ldy = ch;
    rn.kind = 0; rn.eps = 1e-8f; rn.act = 1; rn.rows_per_batch = (int64_t)d.T * d.H * d.W; rn.mod_stride = 2 * ch;
    if (mod) { rn.shift = mod; rn.scale = mod + ch; }
    LTX_TRY(ltx_launch_rownorm(rn, dt, s));
    LTX_TRY(conv3d(v, v->conv_out, v->N.p, out, d, EPI_UNPATCH, nullptr, post, s));
    return LTX_OK;
}

int crop_cl(const void* src, void* dst, size_t esz, int B, int T, int H, int W, int C, int t0, int t1, int h0, int h1, int w0, int w1, hipStream_t s) {
    int64_t n = (int64_t)B * (t1 - t0) * (h1 - h0) * (w1 - w0) * C * (int64_t)esz;
    int64_t blocks = cdiv64(n, 256); if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cl_window_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, (int)esz,
                       B, T, H, W, C, t0, t1, h0, h1, w0, w1);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

struct Tile { float* p; int t, h, w; };   // decoded f32 NCTHW tile [B*3, t, h, w]

// tiled_decode (vae.rs:2225-2290) of a channels-last latent window; result into `out` (dims oT,oH,oW given)
// `pre` (optional): the leaf tiles already decoded, in this loop's order (batched_leaf_decode) - then nothing is decoded here.
int tiled_decode(ltx_vae* v, const void* z, int B, int F, int H, int W, const TimeVec* tv, const ltx_tiling& tl,
                 float* out, std::deque<DevBuf>& pool, size_t& pool_used, hipStream_t s, const std::vector<float*>* pre = nullptr) {
    const ltx_vae_config& c = v->cfg;
    const int r = c.spatial_compression_ratio, tr = c.temporal_compression_ratio;
    const size_t esz = ltx_dt_size(v->dtype);
    const int tmin_h = tl.tile_sample_min_height / r, tmin_w = tl.tile_sample_min_width / r;
    const int ts_h = tl.tile_sample_stride_height / r, ts_w = tl.tile_sample_stride_width / r;
    if (ts_h < 1 || ts_w < 1) LTX_FAIL(LTX_ERR_ARG, "tiling: stride must be >= one latent");
    const int blend_h = std::max(tl.tile_sample_min_height - tl.tile_sample_stride_height, 0);
    const int blend_w = std::max(tl.tile_sample_min_width - tl.tile_sample_stride_width, 0);
    const int oT = (F - 1) * tr + 1, oH = H * r, oW = W * r, BC = B * c.out_channels;
    auto take = [&](size_t bytes, DevBuf** b) -> int {
        if (pool_used >= pool.size()) pool.emplace_back();
        *b = &pool[pool_used++];
        return (*b)->ensure(bytes);
    };
    std::vector<Tile> prev, cur;
    int oy = 0; size_t leaf = 0;
    for (int i = 0; i < H; i += ts_h) {
        cur.clear();
        int ox = 0; int row_h = 0;
        for (int j = 0; j < W; j += ts_w) {
            const int h1 = std::min(i + tmin_h, H), w1 = std::min(j + tmin_w, W);
            const int th = h1 - i, tw = w1 - j;
            Tile t; t.t = oT; t.h = th * r; t.w = tw * r;
            if (pre) {
                if (leaf >= pre->size()) LTX_FAIL(LTX_ERR_ARG, "tiled decode: fewer pre-decoded leaves than tiles");
                t.p = (*pre)[leaf++];
            } else {
                LTX_TRY(v->tile_lat.ensure((size_t)B * F * th * tw * c.latent_channels * esz));
                LTX_TRY(crop_cl(z, v->tile_lat.p, esz, B, F, H, W, c.latent_channels, 0, F, i, h1, j, w1, s));
                DevBuf* tb = nullptr;
                LTX_TRY(take((size_t)BC * t.t * t.h * t.w * sizeof(float), &tb));
                t.p = tb->as<float>();
                LTX_TRY(decoder_forward(v, v->tile_lat.p, B, F, th, tw, tv, 0, t.p, s));
            }
            const size_t ci = cur.size();
            if (!prev.empty()) {        // blend_v with the (already blended) tile above
                BlendArgs ba; ba.a = prev[ci].p; ba.b = t.p; ba.dst = t.p; ba.BC = BC;
                ba.at = prev[ci].t; ba.ah = prev[ci].h; ba.aw = prev[ci].w; ba.a_len = prev[ci].h;
                ba.bt = ba.dt = t.t; ba.bh = ba.dh = t.h; ba.bw = ba.dw = t.w;
                ba.dim = 3; ba.blend = std::min(blend_h, std::min(prev[ci].h, t.h));
                ba.et = t.t; ba.eh = ba.blend; ba.ew = std::min(t.w, prev[ci].w);
                LTX_TRY(ltx_launch_blend(ba, s));
            }
            if (ci > 0) {               // blend_h with the (already blended) tile to the left
                BlendArgs ba; ba.a = cur[ci - 1].p; ba.b = t.p; ba.dst = t.p; ba.BC = BC;
                ba.at = cur[ci - 1].t; ba.ah = cur[ci - 1].h; ba.aw = cur[ci - 1].w; ba.a_len = cur[ci - 1].w;
                ba.bt = ba.dt = t.t; ba.bh = ba.dh = t.h; ba.bw = ba.dw = t.w;
                ba.dim = 4; ba.blend = std::min(blend_w, std::min(cur[ci - 1].w, t.w));
                ba.et = t.t; ba.eh = std::min(t.h, cur[ci - 1].h); ba.ew = ba.blend;
                LTX_TRY(ltx_launch_blend(ba, s));
            }
            cur.push_back(t);
            const int hs = std::min(tl.tile_sample_stride_height, t.h), ws = std::min(tl.tile_sample_stride_width, t.w);
            const int ch = std::min(hs, oH - oy), cw_ = std::min(ws, oW - ox);
            if (ch > 0 && cw_ > 0)
                LTX_TRY(ltx_launch_copy_window(t.p, t.t, t.h, t.w, out, oT, oH, oW, BC, oT, ch, cw_, 0, oy, ox, s));
            ox += ws; row_h = hs;
        }
        oy += row_h;
        prev = cur;
    }
    return LTX_OK;
}

// Leaf tiles of the framewise + spatially tiled decode, decoded up front in batches.  The reference's tiled decode is a serial
// loop of independent `decoder.forward` calls on small latents (vae.rs:2382-2408 around :2246-2257; C2: 52 calls of at most
// 3 x 16 x 16 latents), whose first stages cannot fill the chip one tile at a time (the mid block of one tile is 768
// voxels).  Leaves that share a latent shape - the same spatial tile of different temporal windows - are stacked along the
// batch axis (up to 16 samples per decoder call), and the blends then run over the decoded tiles in exactly the reference's
// order (tiled_decode with `pre`).  A decoder call on a batch gives each sample the bits it gets alone: no op crosses samples.
// per_window[li] = the window's leaves in tiled_decode's loop order.  LTX_VAE_TILE_BATCH=0: one call per leaf (A/B aid).
int batched_leaf_decode(ltx_vae* v, const void* z, int B, int F, int H, int W, const TimeVec* tv, const ltx_tiling& tl,
                        std::vector<std::vector<float*>>& per_window, hipStream_t s) {
    const ltx_vae_config& c = v->cfg;
    const int r = c.spatial_compression_ratio, tr = c.temporal_compression_ratio;
    const size_t esz = ltx_dt_size(v->dtype);
    const int tmin_t = tl.tile_sample_min_num_frames / tr, tstride_t = tl.tile_sample_stride_num_frames / tr;
    const int tmin_h = tl.tile_sample_min_height / r, tmin_w = tl.tile_sample_min_width / r;
    const int ts_h = tl.tile_sample_stride_height / r, ts_w = tl.tile_sample_stride_width / r;
    if (tstride_t < 1 || ts_h < 1 || ts_w < 1) LTX_FAIL(LTX_ERR_ARG, "tiling: stride must be >= one latent");
    struct Leaf { int li, slot, t0, t1, h0, h1, w0, w1; };
    std::vector<Leaf> leaves;
    int nwin = 0;
    for (int i = 0; i < F; i += tstride_t, ++nwin) {
        const int t1 = std::min(i + tmin_t + 1, F);
        int slot = 0;
        for (int y = 0; y < H; y += ts_h)
            for (int x = 0; x < W; x += ts_w) leaves.push_back({nwin, slot++, i, t1, y, std::min(y + tmin_h, H), x, std::min(x + tmin_w, W)});
    }
    per_window.assign(nwin, {});
    for (const Leaf& l : leaves) if ((int)per_window[l.li].size() <= l.slot) per_window[l.li].resize(l.slot + 1, nullptr);
    auto elems = [&](const Leaf& l) { return (size_t)B * c.out_channels * ((size_t)(l.t1 - l.t0 - 1) * tr + 1) * ((size_t)(l.h1 - l.h0) * r) * ((size_t)(l.w1 - l.w0) * r); };
    size_t total = 0;
    for (const Leaf& l : leaves) total += elems(l);
    LTX_TRY(v->predec.ensure(total * sizeof(float)));
    // Leaves per decoder call.  The reference tiles to CAP memory (vae.rs:2225-2290), so the batch must not undo that: it is
    // bounded by a share of the memory that is free now, against a generous estimate of one leaf's activations (six tensors
    // of the largest, 128-channel stage + its f32 output), and a call that still fails to allocate is retried with half as
    // many leaves, down to one.  Option vae_tile_batch=n: at most n leaves per call (-1: one).
    int max_n = LTX_MAX_BATCH / B; if (max_n < 1) max_n = 1;
    if (const int n_opt = ltx_opt().vae_tile_batch) max_n = n_opt < 0 ? 1 : std::min(max_n, n_opt);
    if (v->any_inject) max_n = 1;        // a decoder call draws ONE noise plane per injection for its whole batch (vae.rs:741-753): leaves keep their own draws
    std::vector<char> done(leaves.size(), 0);
    float* cursor = v->predec.as<float>();
    for (size_t a = 0; a < leaves.size(); ++a) {
        if (done[a]) continue;
        const Leaf& la = leaves[a];
        const int nf = la.t1 - la.t0, th = la.h1 - la.h0, tw = la.w1 - la.w0;
        int cap = max_n;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const double act = 6.0 * (double)B * ((double)(nf - 1) * tr + 1) * (th * r / 4.0) * (tw * r / 4.0) * 128.0 * (double)esz + (double)elems(la) * 4.0;
                const double fit = 0.5 * ((double)free_b + (double)v->act_bytes_reserved()) / (act > 1.0 ? act : 1.0);
                if (fit < (double)cap) cap = fit < 1.0 ? 1 : (int)fit;
            } else (void)hipGetLastError();
        }
        std::vector<size_t> grp;
        for (size_t b = a; b < leaves.size() && (int)grp.size() < cap; ++b) {
            const Leaf& lb = leaves[b];
            if (!done[b] && lb.t1 - lb.t0 == nf && lb.h1 - lb.h0 == th && lb.w1 - lb.w0 == tw) { grp.push_back(b); done[b] = 1; }
        }
        const int n = (int)grp.size();
        const size_t leaf_lat = (size_t)B * nf * th * tw * c.latent_channels * esz, leaf_out = elems(la);
        LTX_TRY(v->tile_lat.ensure(leaf_lat * n));
        TimeVec tvb; tvb.n = n * B;
        for (int k = 0; k < LTX_MAX_BATCH; ++k) tvb.t[k] = 0.f;
        for (int k = 0; k < n; ++k) {
            const Leaf& l = leaves[grp[k]];
            LTX_TRY(crop_cl(z, (char*)v->tile_lat.p + leaf_lat * k, esz, B, F, H, W, c.latent_channels, l.t0, l.t1, l.h0, l.h1, l.w0, l.w1, s));
            per_window[l.li][l.slot] = cursor + leaf_out * k;
            if (tv) for (int b = 0; b < B; ++b) tvb.t[k * B + b] = tv->t[b];
        }
        (void)ltx_take_oom();
        const int rc = decoder_forward(v, v->tile_lat.p, n * B, nf, th, tw, tv ? &tvb : nullptr, 0, cursor, s);
        if (rc != LTX_OK) {
            // only an allocation that did not fit is retried with fewer leaves per call; every other failure (argument, launch)
            // is deterministic and surfaces at once with its own message
            if (n == 1 || !ltx_take_oom()) return rc;
            (void)hipGetLastError();
            for (size_t k : grp) done[k] = 0;
            max_n = n / 2 < 1 ? 1 : n / 2;
            --a;
            continue;
        }
        cursor += leaf_out * n;
    }
    return LTX_OK;
}

int decode_cl(ltx_vae* v, const void* z, int B, int F, int H, int W, const TimeVec* tv, const ltx_tiling* tl, int post, float* out, hipStream_t s) {
    const ltx_vae_config& c = v->cfg;
    const int r = c.spatial_compression_ratio, tr = c.temporal_compression_ratio;
    const size_t esz = ltx_dt_size(v->dtype);
    const int BC = B * c.out_channels;
    const int oT = (F - 1) * tr + 1, oH = H * r, oW = W * r;
    const bool framewise = tl && tl->use_framewise_decoding && F > tl->tile_sample_min_num_frames / tr;
    const bool spatial = tl && tl->use_tiling && (W > tl->tile_sample_min_width / r || H > tl->tile_sample_min_height / r);
    if (!framewise && !spatial) return decoder_forward(v, z, B, F, H, W, tv, post, out, s);     // vae.rs:2065 (post == 2: RGB8 from conv_out's epilogue)
    // RGB8 output of a TILED decode: the blends run on f32 tiles - decode into a scratch video, convert with the frame kernel (same bits)
    if (post == 2) {
        LTX_TRY(v->rgbtmp.ensure((size_t)BC * oT * oH * oW * sizeof(float)));
        LTX_TRY(decode_cl(v, z, B, F, H, W, tv, tl, 1, v->rgbtmp.as<float>(), s));
        return ltx_video_to_rgb8(v->rgbtmp.as<float>(), B, oT, oH, oW, reinterpret_cast<uint8_t*>(out), (ltx_stream)s);
    }
    size_t pool_used = 0;
    if (!framewise) {
        LTX_TRY(tiled_decode(v, z, B, F, H, W, tv, *tl, out, v->tilebufs, pool_used, s));
    } else {
        // temporal_tiled_decode (vae.rs:2358-2434)
        const int tmin_t = tl->tile_sample_min_num_frames / tr, tstride_t = tl->tile_sample_stride_num_frames / tr;
        if (tstride_t < 1) LTX_FAIL(LTX_ERR_ARG, "tiling: temporal stride must be >= one latent frame");
        const int blend_t = std::max(tl->tile_sample_min_num_frames - tl->tile_sample_stride_num_frames, 0);
        const int tmin_h = tl->tile_sample_min_height / r, tmin_w = tl->tile_sample_min_width / r;
        DevBuf zt;                       // temporal latent window
        std::vector<std::vector<float*>> pre;                 // spatially tiled windows: every leaf decoded up front, in batches
        if (tl->use_tiling && (H > tmin_h || W > tmin_w)) LTX_TRY(batched_leaf_decode(v, z, B, F, H, W, tv, *tl, pre, s));
        int prev_t = 0, prev_stride = 0; int ot = 0; int li = 0;
        for (int i = 0; i < F; i += tstride_t, ++li) {
            const int t1 = std::min(i + tmin_t + 1, F), nf = t1 - i;
            int rc = zt.ensure((size_t)B * nf * H * W * c.latent_channels * esz);
            if (rc != LTX_OK) { zt.release(); return rc; }
            rc = crop_cl(z, zt.p, esz, B, F, H, W, c.latent_channels, i, t1, 0, H, 0, W, s);
            if (rc != LTX_OK) { zt.release(); return rc; }
            const int dT = (nf - 1) * tr + 1;
            DevBuf& cur = v->tiles[li & 1];
            rc = cur.ensure((size_t)BC * dT * oH * oW * sizeof(float));
            if (rc != LTX_OK) { zt.release(); return rc; }
            const bool sp = tl->use_tiling && (H > tmin_h || W > tmin_w);
            size_t pu = 0;
            rc = sp ? tiled_decode(v, zt.p, B, nf, H, W, tv, *tl, cur.as<float>(), v->tilebufs, pu, s, pre.empty() ? nullptr : &pre[li])
                    : decoder_forward(v, zt.p, B, nf, H, W, tv, 0, cur.as<float>(), s);
            if (rc != LTX_OK) { zt.release(); return rc; }
            // "if i > 0: decoded = decoded[:, :, :-1]" — keep the buffer, shrink the logical length
            int curT = dT; const int stride_full = dT;      // physical T stride of the buffer
            if (li > 0 && curT > 1) curT -= 1;
            // result_row: first tile keeps stride+1 frames, later tiles blend_t(row[idx-1], tile) then keep `stride`
            const int keep = li > 0 ? std::min(tl->tile_sample_stride_num_frames, curT) : std::min(tl->tile_sample_stride_num_frames + 1, curT);
            const int n = std::min(keep, oT - ot);
            if (n > 0) {
                rc = ltx_launch_copy_window(cur.as<float>(), stride_full, oH, oW, out, oT, oH, oW, BC, n, oH, oW, ot, 0, 0, s);
                if (rc != LTX_OK) { zt.release(); return rc; }
                if (li > 0) {       // blend against the RAW previous tile (row[idx-1]), written straight into `out`
                    BlendArgs ba; ba.a = v->tiles[(li - 1) & 1].as<float>(); ba.b = cur.as<float>(); ba.dst = out; ba.BC = BC;
                    ba.at = prev_stride; ba.ah = oH; ba.aw = oW; ba.a_len = prev_t;
                    ba.bt = stride_full; ba.bh = oH; ba.bw = oW; ba.dt = oT; ba.dh = oH; ba.dw = oW; ba.ot = ot;
                    ba.dim = 2; ba.blend = std::min(blend_t, std::min(prev_t, curT));
                    ba.et = std::min(ba.blend, n); ba.eh = oH; ba.ew = oW;
                    rc = ltx_launch_blend(ba, s);
                    if (rc != LTX_OK) { zt.release(); return rc; }
                }
            }
            ot += keep;
            prev_t = curT; prev_stride = stride_full;
        }
        HIP_TRY(hipStreamSynchronize(s));
        zt.release();
    }
    if (post) LTX_TRY(ltx_launch_postprocess(out, (int64_t)BC * oT * oH * oW, s));
    return LTX_OK;
}

}  // namespace

extern "C" int ltx_vae_create(const ltx_vae_config* cfg, const ltx_weight* weights, size_t n_weights,
                              ltx_dtype model_dtype, int device, ltx_vae** out) {
    if (!cfg || !weights || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_create: null argument");
    *out = nullptr;
    if (cfg->n_blocks < 1 || cfg->n_blocks > 4) LTX_FAIL(LTX_ERR_ARG, "n_blocks must be 1..4");
    if (cfg->patch_size != 4 || cfg->patch_size_t != 1) LTX_FAIL(LTX_ERR_UNSUPPORTED, "only patch_size=4, patch_size_t=1 are supported");
    if (cfg->latent_channels % 8 != 0) LTX_FAIL(LTX_ERR_UNSUPPORTED, "latent_channels must be a multiple of 8");
    for (int i = 0; i < cfg->n_blocks; ++i) {
        if (cfg->decoder_upsample_factor[i] < 1 || cfg->decoder_block_out_channels[i] % (8 * cfg->decoder_upsample_factor[i]) != 0)
            LTX_FAIL(LTX_ERR_UNSUPPORTED, "decoder channels must be multiples of 8*upsample_factor");
    }
    HIP_TRY(hipSetDevice(device));
    ltx_vae* v = new ltx_vae();
    v->cfg = *cfg; v->dtype = model_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32; v->device = device;
    int rc = build(v, weights, n_weights);
    if (rc != LTX_OK) { v->free_all(); delete v; return rc; }
    *out = v;
    return LTX_OK;
}
extern "C" void ltx_vae_destroy(ltx_vae* v) {
    if (!v) return;
    (void)hipSetDevice(v->device); (void)hipDeviceSynchronize();
    v->free_all(); delete v;
}
extern "C" int ltx_vae_set_noise_seed(ltx_vae* v, uint64_t seed) {
    if (!v) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_set_noise_seed: null handle");
    v->noise_seed = seed; v->noise_ctr = 0; return LTX_OK;
}
extern "C" int ltx_vae_injects_noise(const ltx_vae* v) { return v && v->any_inject ? 1 : 0; }
extern "C" int ltx_vae_get_config(const ltx_vae* v, ltx_vae_config* out) {
    if (!v || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_get_config: null argument");
    *out = v->cfg; return LTX_OK;
}
extern "C" const float* ltx_vae_latents_mean(const ltx_vae* v) { return v ? v->mean : nullptr; }
extern "C" const float* ltx_vae_latents_std(const ltx_vae* v) { return v ? v->std_ : nullptr; }

static int check_decode_args(ltx_vae* v, int B, int F, int H, int W) {
    if (!v) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_decode: null handle");
    if (B < 1 || F < 1 || H < 1 || W < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_decode: bad shape");
    return LTX_OK;
}
// The trait has no batch bound (t2v_pipeline.rs:102); samples never interact in the decoder (vae.rs:2107-2121 even decodes them
// one at a time under use_slicing), so batches beyond the 8 per-sample scalars a launch carries run as chunks of 8.
static size_t out_elems_per_sample(const ltx_vae* v, int F, int H, int W) {
    const ltx_vae_config& c = v->cfg;
    return (size_t)c.out_channels * ((size_t)(F - 1) * c.temporal_compression_ratio + 1) * ((size_t)H * c.spatial_compression_ratio) * ((size_t)W * c.spatial_compression_ratio);
}

extern "C" int ltx_vae_decode(ltx_vae* v, const void* latents, ltx_dtype io_dtype, const float* timestep,
                              int B, int F, int H, int W, const ltx_tiling* tiling, int postprocess,
                              float* out, ltx_stream stream) {
    LTX_TRY(check_decode_args(v, B, F, H, W));
    if (!latents || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_decode: null tensor");
    if (B > 8) {
        const size_t in_b = (size_t)v->cfg.latent_channels * F * H * W * (io_dtype == LTX_BF16 ? 2 : 4), out_b = out_elems_per_sample(v, F, H, W);
        for (int b0 = 0; b0 < B; b0 += 8)
            LTX_TRY(ltx_vae_decode(v, (const char*)latents + b0 * in_b, io_dtype, timestep ? timestep + b0 : nullptr, B - b0 < 8 ? B - b0 : 8, F, H, W, tiling,
                                   postprocess, postprocess == 2 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(out) + b0 * out_b) : out + b0 * out_b, stream));
        return LTX_OK;
    }
    HIP_TRY(hipSetDevice(v->device));
    hipStream_t s = (hipStream_t)stream;
    const int C = v->cfg.latent_channels; const int64_t S = (int64_t)F * H * W;
    LTX_TRY(v->zin.ensure((size_t)B * S * C * ltx_dt_size(v->dtype)));
    LTX_TRY(ltx_launch_ncthw_to_cl(latents, io_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32, v->zin.p, v->dtype, B, C, S, s));
    TimeVec tv; tv.n = B; for (int i = 0; i < 8; ++i) tv.t[i] = (timestep && i < B) ? timestep[i] : 0.f;
    return decode_cl(v, v->zin.p, B, F, H, W, timestep ? &tv : nullptr, tiling, postprocess, out, s);
}

extern "C" int ltx_vae_decode_tokens(ltx_vae* v, const float* tokens, const float* noise, const float* noise_scale,
                                     const float* timestep, int B, int F, int H, int W, const ltx_tiling* tiling,
                                     int postprocess, float* out, ltx_stream stream) {
    LTX_TRY(check_decode_args(v, B, F, H, W));
    if (!tokens || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_decode_tokens: null tensor");
    if (noise && !noise_scale) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_decode_tokens: noise needs noise_scale");
    if (B > 8) {
        const size_t in_e = (size_t)v->cfg.latent_channels * F * H * W, out_b = out_elems_per_sample(v, F, H, W);
        for (int b0 = 0; b0 < B; b0 += 8)
            LTX_TRY(ltx_vae_decode_tokens(v, tokens + b0 * in_e, noise ? noise + b0 * in_e : nullptr, noise ? noise_scale + b0 : nullptr, timestep ? timestep + b0 : nullptr,
                                          B - b0 < 8 ? B - b0 : 8, F, H, W, tiling, postprocess,
                                          postprocess == 2 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(out) + b0 * out_b) : out + b0 * out_b, stream));
        return LTX_OK;
    }
    HIP_TRY(hipSetDevice(v->device));
    hipStream_t s = (hipStream_t)stream;
    const int C = v->cfg.latent_channels; const int64_t S = (int64_t)F * H * W;
    LTX_TRY(v->zin.ensure((size_t)B * S * C * ltx_dt_size(v->dtype)));
    TimeVec ns; ns.n = B; for (int i = 0; i < 8; ++i) ns.t[i] = (noise && i < B) ? noise_scale[i] : 0.f;
    LTX_TRY(ltx_launch_denorm_mix(tokens, v->mean, v->std_, 1.0f / v->cfg.scaling_factor, noise, ns, v->zin.p, v->dtype, B, S, C, s));
    TimeVec tv; tv.n = B; for (int i = 0; i < 8; ++i) tv.t[i] = (timestep && i < B) ? timestep[i] : 0.f;
    return decode_cl(v, v->zin.p, B, F, H, W, timestep ? &tv : nullptr, tiling, postprocess, out, s);
}

extern "C" int ltx_vae_prepare_latents(ltx_vae* v, const float* tokens, const float* noise, const float* noise_scale,
                                       int B, int F, int H, int W, float* out_tokens, ltx_stream stream) {
    LTX_TRY(check_decode_args(v, B, F, H, W));
    if (!tokens || !out_tokens) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_prepare_latents: null tensor");
    if (noise && !noise_scale) LTX_FAIL(LTX_ERR_ARG, "ltx_vae_prepare_latents: noise needs noise_scale");
    if (B > 8) {
        const size_t in_e = (size_t)v->cfg.latent_channels * F * H * W;
        for (int b0 = 0; b0 < B; b0 += 8)
            LTX_TRY(ltx_vae_prepare_latents(v, tokens + b0 * in_e, noise ? noise + b0 * in_e : nullptr, noise ? noise_scale + b0 : nullptr, B - b0 < 8 ? B - b0 : 8, F, H, W,
                                            out_tokens + b0 * in_e, stream));
        return LTX_OK;
    }
    HIP_TRY(hipSetDevice(v->device));
    const int C = v->cfg.latent_channels; const int64_t S = (int64_t)F * H * W;
    TimeVec ns; ns.n = B; for (int i = 0; i < 8; ++i) ns.t[i] = (noise && i < B) ? noise_scale[i] : 0.f;
    return ltx_launch_denorm_mix(tokens, v->mean, v->std_, 1.0f / v->cfg.scaling_factor, noise, ns, out_tokens, LTX_DT_F32, B, S, C,
                                 (hipStream_t)stream);
}
