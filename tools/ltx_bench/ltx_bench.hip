// ltx_bench - the headline measurement without Python or torch in the process (SURVEY.md 8d "harness: C++ ltx_bench linking
// the HIP lib").  Everything it does goes through include/ltxhip*.h, the way a Rust host would over FFI
// (INTEGRATION.md): presets -> configs, named weights -> ltx_dit_create / ltx_vae_create, ltx_pipeline_call per video.  The
// only HIP code of its own is the generator of the synthetic weights (there are no checkpoints offline).
//
//   ltx_bench [--config c1|c2] [--preset NAME] [--steps K] [--warmup W] [--check 0|1]
//
// prints one JSON line: videos/s is NOT the contract metric of bench.py (that stays the driver's entry); this program exists
// to show that the rate does not depend on the Python plumbing above the C ABI and to give a host without Python a template.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ltxhip.h"
#include "ltxhip_presets.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } } while (0)
#define LTXCHECK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s:%d rc=%d %s\n", __FILE__, __LINE__, r_, ltx_last_error()); std::exit(2); } } while (0)

// counter-based N(mean, std): one SplitMix64 draw per element, Box-Muller on its two halves
__device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__device__ inline float normal_at(uint64_t seed, uint64_t i) {
    uint64_t r = mix64(seed ^ mix64(i));
    float u1 = ((uint32_t)(r >> 40) + 1) * (1.0f / 16777217.0f), u2 = (uint32_t)(r & 0xFFFFFF) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}
__global__ void fill_normal_bf16(uint16_t* p, size_t n, uint64_t seed, float mean, float std) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t b = __float_as_uint(mean + std * normal_at(seed, i));
        p[i] = (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
    }
}
__global__ void fill_normal_f32(float* p, size_t n, uint64_t seed, float mean, float std) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = mean + std * normal_at(seed, i);
}

struct Named { std::string name; std::vector<int64_t> shape; };

// LtxVideoTransformer3DModel::new (ltx_transformer.rs:957-1003): the names and shapes its VarBuilder asks for
static std::vector<Named> dit_names(const ltx_dit_config& c) {
    int64_t D = (int64_t)c.num_attention_heads * c.attention_head_dim;
    std::vector<Named> s;
    auto linear = [&](const std::string& n, int64_t in, int64_t out) { s.push_back({n + ".weight", {out, in}}); s.push_back({n + ".bias", {out}}); };
    linear("proj_in", c.in_channels, D);
    s.push_back({"scale_shift_table", {2, D}});
    linear("time_embed.emb.timestep_embedder.linear_1", 256, D);
    linear("time_embed.emb.timestep_embedder.linear_2", D, D);
    linear("time_embed.linear", D, 6 * D);
    linear("caption_projection.linear_1", c.caption_channels, D);
    linear("caption_projection.linear_2", D, D);
    for (int i = 0; i < c.num_layers; ++i) {
        std::string b = "transformer_blocks." + std::to_string(i) + ".";
        for (int a = 1; a <= 2; ++a) {
            std::string at = b + "attn" + std::to_string(a);
            int64_t kv = a == 1 ? D : c.cross_attention_dim;
            linear(at + ".to_q", D, D); linear(at + ".to_k", kv, D); linear(at + ".to_v", kv, D); linear(at + ".to_out.0", D, D);
            s.push_back({at + ".norm_q.weight", {D}}); s.push_back({at + ".norm_k.weight", {D}});
        }
        linear(b + "ff.net.0.proj", D, 4 * D); linear(b + "ff.net.2", 4 * D, D);
        s.push_back({b + "scale_shift_table", {6, D}});
    }
    linear("proj_out", D, c.out_channels);
    return s;
}

// LtxVideoDecoder3d::new (vae.rs:1521-1608): widest block first, each up block = upsampler (x8 channels for the 2x2x2
// depth-to-space) + its resnets; timestep conditioning adds the embedders and the scale/shift tables
static std::vector<Named> vae_names(const ltx_vae_config& c) {
    std::vector<Named> s;
    bool tc = c.timestep_conditioning != 0;
    const std::string P = "decoder.";
    auto conv = [&](const std::string& n, int64_t ci, int64_t co) { s.push_back({P + n + ".conv.weight", {co, ci, 3, 3, 3}}); s.push_back({P + n + ".conv.bias", {co}}); };
    auto embedder = [&](const std::string& n, int64_t d) {
        s.push_back({P + n + ".timestep_embedder.linear_1.weight", {d, 256}}); s.push_back({P + n + ".timestep_embedder.linear_1.bias", {d}});
        s.push_back({P + n + ".timestep_embedder.linear_2.weight", {d, d}});   s.push_back({P + n + ".timestep_embedder.linear_2.bias", {d}});
    };
    auto resnet = [&](const std::string& n, int64_t ch) { conv(n + ".conv1", ch, ch); conv(n + ".conv2", ch, ch); if (tc) s.push_back({P + n + ".scale_shift_table", {4, ch}}); };
    int nb = c.n_blocks;
    std::vector<int64_t> chans, layers, upf;
    for (int i = nb - 1; i >= 0; --i) { chans.push_back(c.decoder_block_out_channels[i]); upf.push_back(c.decoder_upsample_factor[i]); }
    for (int i = nb; i >= 0; --i) layers.push_back(c.decoder_layers_per_block[i]);
    int64_t mid = chans[0];
    conv("conv_in", c.latent_channels, mid);
    if (tc) embedder("mid_block.time_embedder", 4 * mid);
    for (int i = 0; i < layers[0]; ++i) resnet("mid_block.resnets." + std::to_string(i), mid);
    int64_t cur = mid;
    for (int bi = 0; bi < nb; ++bi) {
        int64_t ch = cur / upf[bi];
        std::string up = "up_blocks." + std::to_string(bi);
        conv(up + ".upsamplers.0.conv", cur, ch * 8);
        if (tc) embedder(up + ".time_embedder", 4 * ch);
        for (int i = 0; i < layers[bi + 1]; ++i) resnet(up + ".resnets." + std::to_string(i), ch);
        cur = ch;
    }
    conv("conv_out", cur, (int64_t)c.out_channels * c.patch_size * c.patch_size);
    if (tc) { embedder("time_embedder", 2 * cur); s.push_back({P + "scale_shift_table", {2, cur}}); s.push_back({P + "timestep_scale_multiplier", {}}); }
    s.push_back({"latents_mean", {c.latent_channels}}); s.push_back({"latents_std", {c.latent_channels}});
    return s;
}

static bool ends_with(const std::string& s, const char* t) { size_t n = std::strlen(t); return s.size() >= n && s.compare(s.size() - n, n, t) == 0; }

// Random-init weights of the real architecture directly in HBM, scaled like bench.py's synth_on_device so activations stay O(1):
// matrices / conv kernels N(0, 1/fan_in) bf16; biases 0.02 N; norm gains 1 + 0.1 N; tables N / sqrt(width).
struct Weights {
    std::vector<Named> names; std::vector<void*> bufs; std::vector<ltx_weight> w;
    void build(std::vector<Named> nm, uint64_t seed) {
        names = std::move(nm);
        for (size_t k = 0; k < names.size(); ++k) {
            const Named& n = names[k];
            size_t cnt = 1; for (int64_t d : n.shape) cnt *= (size_t)d;
            bool matrix = n.shape.size() >= 2 && !ends_with(n.name, "scale_shift_table");
            void* p = nullptr;
            HIPCHECK(hipMalloc(&p, cnt * (matrix ? 2 : 4)));
            unsigned grid = (unsigned)std::min<size_t>((cnt + 255) / 256, 8192);
            uint64_t sd = seed * 0x100000001B3ull + k;
            if (matrix) {
                size_t fan = cnt / (size_t)n.shape[0];
                fill_normal_bf16<<<grid, 256>>>((uint16_t*)p, cnt, sd, 0.f, 1.0f / std::sqrt((float)fan));
            } else {
                float mean = 0.f, sdv = 0.02f;
                if (ends_with(n.name, "timestep_scale_multiplier")) { mean = 1000.f; sdv = 0.f; }
                else if (n.name.find("norm_q") != std::string::npos || n.name.find("norm_k") != std::string::npos) { mean = 1.f; sdv = 0.1f; }
                else if (ends_with(n.name, "scale_shift_table")) sdv = 1.0f / std::sqrt((float)n.shape.back());
                else if (n.name == "latents_mean") sdv = 0.1f;
                else if (n.name == "latents_std") { mean = 1.f; sdv = 0.05f; }
                fill_normal_f32<<<grid, 256>>>((float*)p, cnt, sd, mean, sdv);
            }
            bufs.push_back(p);
            ltx_weight lw; std::memset(&lw, 0, sizeof(lw));
            lw.name = names[k].name.c_str(); lw.data = p; lw.dtype = matrix ? LTX_BF16 : LTX_F32; lw.ndim = (int)n.shape.size(); lw.on_device = 1;
            for (size_t d = 0; d < n.shape.size(); ++d) lw.shape[d] = n.shape[d];
            w.push_back(lw);
        }
        HIPCHECK(hipDeviceSynchronize());
    }
    void release() { for (void* p : bufs) (void)hipFree(p); bufs.clear(); w.clear(); }
};

int main(int argc, char** argv) {
    std::string config = "c2", preset_name;
    int steps = 10, warmup = 2, check = 1;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); } return argv[++i]; };
        if (a == "--config") config = val(); else if (a == "--preset") preset_name = val();
        else if (a == "--steps") steps = std::atoi(val()); else if (a == "--warmup") warmup = std::atoi(val());
        else if (a == "--check") check = std::atoi(val());
        else { std::fprintf(stderr, "usage: ltx_bench [--config c1|c2] [--preset NAME] [--steps K] [--warmup W] [--check 0|1]\n"); return 1; }
    }
    int height, width, frames;                                    // BASELINE.json configs[0] / configs[1]
    if (config == "c1") { height = 256; width = 384; frames = 25; }
    else if (config == "c2") { height = 512; width = 768; frames = 97; }
    else { std::fprintf(stderr, "unknown config %s\n", config.c_str()); return 1; }
    if (preset_name.empty()) preset_name = "0.9.8-2b-distilled";

    HIPCHECK(hipSetDevice(0));
    ltx_preset pre; LTXCHECK(ltx_preset_get(preset_name.c_str(), &pre));
    auto t_build = std::chrono::steady_clock::now();
    ltx_dit* dit = nullptr; ltx_vae* vae = nullptr;
    { Weights w; w.build(dit_names(pre.transformer), 1); LTXCHECK(ltx_dit_create(&pre.transformer, w.w.data(), w.w.size(), LTX_BF16, 0, &dit)); w.release(); }
    { Weights w; w.build(vae_names(pre.vae), 2); LTXCHECK(ltx_vae_create(&pre.vae, w.w.data(), w.w.size(), LTX_BF16, 0, &vae)); w.release(); }

    const int B = 1, K = 128, C = pre.transformer.in_channels;
    const int F = (frames - 1) / pre.vae.temporal_compression_ratio + 1, H = height / pre.vae.spatial_compression_ratio, W = width / pre.vae.spatial_compression_ratio;
    const size_t S = (size_t)F * H * W;
    // initial latents: the reference's PCG32 Box-Muller stream (deterministic_rng.rs) in [B,C,F,H,W] order, packed to [B,S,C]
    std::vector<float> z(C * S), packed(S * C);
    LTXCHECK(ltx_pcg32_randn(42, 1442695040888963407ull, z.size(), z.data()));
    for (int c = 0; c < C; ++c) for (size_t s = 0; s < S; ++s) packed[s * C + c] = z[(size_t)c * S + s];
    float *lat0, *lat, *pe, *pm, *ne, *nm, *noise, *video;
    const size_t video_n = (size_t)B * 3 * frames * height * width;
    HIPCHECK(hipMalloc(&lat0, packed.size() * 4)); HIPCHECK(hipMalloc(&lat, packed.size() * 4));
    HIPCHECK(hipMemcpy(lat0, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    const size_t en = (size_t)B * K * pre.transformer.caption_channels;
    HIPCHECK(hipMalloc(&pe, en * 4)); HIPCHECK(hipMalloc(&ne, en * 4)); HIPCHECK(hipMalloc(&pm, K * 4)); HIPCHECK(hipMalloc(&nm, K * 4));
    HIPCHECK(hipMalloc(&noise, z.size() * 4)); HIPCHECK(hipMalloc(&video, video_n * 4));
    fill_normal_f32<<<4096, 256>>>(pe, en, 42, 0.f, 1.f); fill_normal_f32<<<4096, 256>>>(ne, en, 43, 0.f, 1.f);
    fill_normal_f32<<<4096, 256>>>(noise, z.size(), 44, 0.f, 1.f);
    std::vector<float> m(K, 0.f);
    for (int i = 0; i < 32; ++i) m[i] = 1.f;
    HIPCHECK(hipMemcpy(pm, m.data(), K * 4, hipMemcpyHostToDevice));
    for (int i = 8; i < 32; ++i) m[i] = 0.f;
    HIPCHECK(hipMemcpy(nm, m.data(), K * 4, hipMemcpyHostToDevice));

    ltx_pipeline_params p; LTXCHECK(ltx_pipeline_params_from_preset(&pre, &p));
    p.height = height; p.width = width; p.num_frames = frames; p.frame_rate = 25; p.postprocess = 1;
    const bool cfg_on = p.guidance_scale > 1.0f;
    hipStream_t stream; HIPCHECK(hipStreamCreate(&stream));
    LTXCHECK(ltx_warmup(dit, vae, B, F, H, W, K, stream));
    double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();

    auto one_video = [&]() {
        HIPCHECK(hipMemcpyAsync(lat, lat0, packed.size() * 4, hipMemcpyDeviceToDevice, stream));
        LTXCHECK(ltx_pipeline_call(dit, vae, &p, lat, pe, pm, cfg_on ? ne : nullptr, cfg_on ? nm : nullptr, noise, B, K, video, stream));
    };
    for (int i = 0; i < warmup + 1; ++i) one_video();
    HIPCHECK(hipStreamSynchronize(stream));
    double dit_ms = 0, vae_ms = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) { one_video(); float t[4]; LTXCHECK(ltx_pipeline_last_timing(t)); dit_ms += t[0]; vae_ms += t[2]; }
    HIPCHECK(hipStreamSynchronize(stream));
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    int finite = -1; double mean = 0;
    if (check) {                                                  // the video is a picture: finite, inside [0, 255], not constant
        std::vector<float> v(video_n);
        HIPCHECK(hipMemcpy(v.data(), video, video_n * 4, hipMemcpyDeviceToHost));
        finite = 1; float lo = v[0], hi = v[0];
        for (float x : v) { if (!std::isfinite(x) || x < 0.f || x > 255.f) finite = 0; mean += x; lo = std::min(lo, x); hi = std::max(hi, x); }
        mean /= (double)video_n;
        if (!(hi > lo)) finite = 0;
    }
    std::printf("{\"program\": \"ltx_bench (C ABI only, no Python in the process)\", \"preset\": \"%s\", \"config\": \"%s\", \"height\": %d, \"width\": %d, "
                "\"num_frames\": %d, \"denoise_steps\": %d, \"videos\": %d, \"warmup\": %d, \"s_per_video\": %.5f, \"frames_per_s\": %.2f, "
                "\"dit_ms_per_video\": %.2f, \"vae_ms_per_video\": %.2f, \"build_and_warmup_s\": %.1f, \"video_ok\": %d, \"video_mean\": %.3f}\n",
                pre.version, config.c_str(), height, width, frames, p.num_inference_steps, steps, warmup, el / steps, frames * steps / el,
                dit_ms / steps, vae_ms / steps, build_s, finite, mean);
    ltx_dit_destroy(dit); ltx_vae_destroy(vae);
    return finite == 0 ? 3 : 0;
}
