// Host side of the drop-in: LtxPipeline::call control flow, FlowMatchEulerDiscreteScheduler scalar
// math, PCG32 latent generator and video_coords — C++ restatement above the kernel C ABI.
// Reference: src/models/ltx_video/t2v_pipeline.rs (:627-1073), scheduler.rs (:172-207, 274-412, 495-595,
// 646-668), src/utils/deterministic_rng.rs (:11-81), examples/ltx-video/main.rs (:567-646).
#include <cmath>
#include <cstring>
#include <vector>
#include "../csrc/model_util.h"
#include "../csrc/options.h"

// calculate_shift (t2v_pipeline.rs:159-169), f32 arithmetic
extern "C" float ltx_calculate_shift(int seq_len, int base_seq_len, int max_seq_len, float base_shift, float max_shift) {
    float m = (max_shift - base_shift) / (float)(max_seq_len - base_seq_len);
    float b = base_shift - m * (float)base_seq_len;
    return (float)seq_len * m + b;
}

// set_timesteps with explicit sigmas (scheduler.rs:320-412) + i64 truncation of the trait wrapper (:658-659)
extern "C" int ltx_sched_set_timesteps(const float* sigmas_in, int n, float mu, int use_mu, float shift,
                                       float shift_terminal, int use_shift_terminal, float* sigmas_out, int64_t* timesteps_out) {
    if (!sigmas_in || n < 1 || !sigmas_out) LTX_FAIL(LTX_ERR_ARG, "ltx_sched_set_timesteps: bad argument");
    std::vector<float> s(sigmas_in, sigmas_in + n);
    if (use_mu) {                                   // time_shift_scalar, exponential (:172-179), sigma = 1
        const float emu = std::exp(mu);
        for (float& v : s) { float base = std::pow(1.0f / v - 1.0f, 1.0f); v = emu / (emu + base); }
    } else {
        for (float& v : s) v = shift * v / (1.0f + (shift - 1.0f) * v);
    }
    if (use_shift_terminal) {                       // stretch_shift_to_terminal_vec (:188-207)
        const float one_minus_last = 1.0f - s[n - 1];
        const float denom = 1.0f - shift_terminal;
        if (std::fabs(denom) < 1e-12f) LTX_FAIL(LTX_ERR_ARG, "shift_terminal too close to 1.0");
        const float scale = one_minus_last / denom;
        for (float& v : s) v = 1.0f - ((1.0f - v) / scale);
    }
    for (int i = 0; i < n; ++i) {
        sigmas_out[i] = s[i];
        if (timesteps_out) timesteps_out[i] = (int64_t)(s[i] * 1000.0f);
    }
    sigmas_out[n] = 0.0f;                           // terminal sigma (:398)
    return LTX_OK;
}

namespace {
// Regularised incomplete beta function I_x(a, b) (continued fraction, modified Lentz) and its inverse by bisection + Newton, f64:
// what statrs' Beta::inverse_cdf / scipy.stats.beta.ppf compute (scheduler.rs:247-272 calls the former).
double betacf(double a, double b, double x) {
    const double tiny = 1e-300;
    double c = 1.0, d = 1.0 - (a + b) * x / (a + 1.0);
    if (std::fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((a + m2 - 1.0) * (a + m2));
        d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (a + b + m) * x / ((a + m2) * (a + m2 + 1.0));
        d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}
double betai(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double lbt = std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log1p(-x);
    if (x < (a + 1.0) / (a + b + 2.0)) return std::exp(lbt) * betacf(a, b, x) / a;
    return 1.0 - std::exp(lbt) * betacf(b, a, 1.0 - x) / b;
}
double beta_ppf(double p, double a, double b) {
    if (p <= 0.0) return 0.0;
    if (p >= 1.0) return 1.0;
    double lo = 0.0, hi = 1.0, x = 0.5;
    for (int it = 0; it < 200; ++it) {
        const double f = betai(a, b, x) - p;
        if (f > 0.0) hi = x; else lo = x;
        // Newton step from the density; kept only if it stays inside the bracket
        const double lpdf = std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + (a - 1.0) * std::log(x) + (b - 1.0) * std::log1p(-x);
        double xn = x - f / std::exp(lpdf);
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (std::fabs(xn - x) < 1e-15) { x = xn; break; }
        x = xn;
    }
    return x;
}
std::vector<float> linspace_f32(float start, float end, int steps) {     // scheduler.rs:209-220
    std::vector<float> v((size_t)(steps > 0 ? steps : 0));
    if (steps == 1) v[0] = start;
    else for (int i = 0; i < steps; ++i) v[i] = start + (end - start) * (float)i / (float)(steps - 1);
    return v;
}
}  // namespace

extern "C" int ltx_sched_set_timesteps_ex(const float* sigmas_in, int n, float mu, int use_mu, float shift,
                                          float shift_terminal, int use_shift_terminal, int sigma_kind, int invert_sigmas,
                                          float* sigmas_out, int64_t* timesteps_out) {
    if (sigma_kind < 0 || sigma_kind > 3) LTX_FAIL(LTX_ERR_ARG, "ltx_sched_set_timesteps_ex: sigma_kind must be 0 (none), 1 (karras), 2 (exponential) or 3 (beta)");
    LTX_TRY(ltx_sched_set_timesteps(sigmas_in, n, mu, use_mu, shift, shift_terminal, use_shift_terminal, sigmas_out, timesteps_out));
    std::vector<float> s(sigmas_out, sigmas_out + n);
    const float smin = s[n - 1], smax = s[0];
    if (sigma_kind == 1) {                           // convert_to_karras (:222-235)
        const float rho = 7.0f, mn = std::pow(smin, 1.0f / rho), mx = std::pow(smax, 1.0f / rho);
        const std::vector<float> ramp = linspace_f32(0.0f, 1.0f, n);
        for (int i = 0; i < n; ++i) s[i] = std::pow(mx + ramp[i] * (mn - mx), rho);
    } else if (sigma_kind == 2) {                    // convert_to_exponential (:237-245)
        const std::vector<float> logs = linspace_f32(std::log(smax), std::log(smin), n);
        for (int i = 0; i < n; ++i) s[i] = std::exp(logs[i]);
    } else if (sigma_kind == 3) {                    // convert_to_beta (:247-272), alpha = beta = 0.6 (:369)
        const std::vector<float> lin = linspace_f32(0.0f, 1.0f, n);
        for (int i = 0; i < n; ++i) {
            const double t = 1.0 - (double)lin[i];
            s[i] = (float)((double)smin + beta_ppf(t, 0.6, 0.6) * (double)(smax - smin));
        }
    }
    for (int i = 0; i < n; ++i) {
        if (invert_sigmas) s[i] = 1.0f - s[i];
        sigmas_out[i] = s[i];
        if (timesteps_out) timesteps_out[i] = (int64_t)(s[i] * 1000.0f);
    }
    sigmas_out[n] = invert_sigmas ? 1.0f : 0.0f;     // terminal sigma (:389-399)
    return LTX_OK;
}

// ---- PCG32 (deterministic_rng.rs) ----
namespace {
struct Pcg32 {
    uint64_t state, inc;
    Pcg32(uint64_t seed, uint64_t inc_) : state(0), inc((inc_ << 1) | 1) { next_u32(); state += seed; next_u32(); }
    uint32_t next_u32() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31));
    }
    float next_f32() { return (float)(next_u32() >> 8) * 5.9604645e-8f; }
    void next_gaussian(float& z0, float& z1) {
        float u1;
        do { u1 = next_f32(); } while (!(u1 > 1e-7f));
        float u2 = next_f32();
        float mag = std::sqrt(-2.0f * std::log(u1));
        const float two_pi_u2 = 2.0f * 3.14159265358979323846f * u2;
        z0 = mag * std::cos(two_pi_u2); z1 = mag * std::sin(two_pi_u2);
    }
};
}  // namespace

extern "C" int ltx_pcg32_randn(uint64_t seed, uint64_t inc, size_t n, float* out) {
    if (!out) LTX_FAIL(LTX_ERR_ARG, "ltx_pcg32_randn: null output");
    Pcg32 r(seed, inc);
    for (size_t i = 0; i < n; i += 2) {
        float a, b; r.next_gaussian(a, b);
        out[i] = a; if (i + 1 < n) out[i + 1] = b;
    }
    return LTX_OK;
}

// the integer stream itself (deterministic_rng.rs:23-35): bit-exact by construction, pinned to the published PCG32 vectors
extern "C" int ltx_pcg32_u32(uint64_t seed, uint64_t inc, size_t n, uint32_t* out) {
    if (!out) LTX_FAIL(LTX_ERR_ARG, "ltx_pcg32_u32: null output");
    Pcg32 r(seed, inc);
    for (size_t i = 0; i < n; ++i) out[i] = r.next_u32();
    return LTX_OK;
}

// ---- device memory for hosts that have no HIP bindings of their own (the Rust shim, rust/hip_backend.rs) ----
extern "C" int ltx_device_alloc(size_t bytes, int device, void** out) {
    if (!out) LTX_FAIL(LTX_ERR_ARG, "ltx_device_alloc: null output");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return LTX_OK;
}
extern "C" int ltx_device_free(void* p) { if (p) HIP_TRY(hipFree(p)); return LTX_OK; }
extern "C" int ltx_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes, ltx_stream stream) {
    HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return LTX_OK;
}
extern "C" int ltx_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes, ltx_stream stream) {
    HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return LTX_OK;
}
extern "C" int ltx_stream_synchronize(ltx_stream stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return LTX_OK; }

// video_coords (t2v_pipeline.rs:798-847): f' = clamp(8f-7, 0, 1000)/frame_rate ; h' = 32h ; w' = 32w
extern "C" int ltx_build_video_coords(int F, int H, int W, int frame_rate, int ts_ratio, int sp_ratio, float* out) {
    if (!out || F < 1 || H < 1 || W < 1 || frame_rate < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_build_video_coords: bad argument");
    const float inv_fr = (float)(1.0 / (double)frame_rate);
    size_t i = 0;
    for (int f = 0; f < F; ++f)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                float vf = (float)f * (float)ts_ratio + (1.0f - (float)ts_ratio);
                vf = std::fmin(std::fmax(vf, 0.0f), 1000.0f) * inv_fr;
                out[i++] = vf; out[i++] = (float)h * (float)sp_ratio; out[i++] = (float)w * (float)sp_ratio;
            }
    return LTX_OK;
}

extern "C" int ltx_guidance_step(const void* text, const void* uncond, const void* perturbed, ltx_dtype pred_dtype,
                                 float* latents, float* noise_pred_out, int B, int64_t n,
                                 float guidance_scale, float guidance_rescale, float stg_scale, float dt,
                                 void* stats_ws, ltx_stream stream) {
    GuidanceArgs a;
    a.text = text; a.uncond = uncond; a.pert = perturbed; a.pred_dtype = pred_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32;
    a.latents = latents; a.noise_out = noise_pred_out; a.B = B; a.n_per_batch = n;
    a.guidance_scale = guidance_scale; a.guidance_rescale = guidance_rescale; a.stg_scale = stg_scale; a.dt = dt;
    a.stats = reinterpret_cast<double*>(stats_ws);
    return ltx_launch_guidance_step(a, (hipStream_t)stream);
}

extern "C" int ltx_guidance_step_stochastic(const void* text, const void* uncond, const void* perturbed, ltx_dtype pred_dtype,
                                            float* latents, float* noise_pred_out, int B, int64_t n,
                                            float guidance_scale, float guidance_rescale, float stg_scale,
                                            float sigma, float sigma_next, const float* step_noise,
                                            void* stats_ws, ltx_stream stream) {
    if (!step_noise || !latents) LTX_FAIL(LTX_ERR_ARG, "ltx_guidance_step_stochastic: latents and step_noise are required");
    GuidanceArgs a;
    a.text = text; a.uncond = uncond; a.pert = perturbed; a.pred_dtype = pred_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32;
    a.latents = latents; a.noise_out = noise_pred_out; a.B = B; a.n_per_batch = n;
    a.guidance_scale = guidance_scale; a.guidance_rescale = guidance_rescale; a.stg_scale = stg_scale;
    a.sigma = sigma; a.sigma_next = sigma_next; a.step_noise = step_noise;
    a.stats = reinterpret_cast<double*>(stats_ws);
    return ltx_launch_guidance_step(a, (hipStream_t)stream);
}

// ---- LtxPipeline::call ----
static thread_local float g_timing[4] = {0, 0, 0, 0};
extern "C" int ltx_pipeline_last_timing(float ms[4]) {
    if (!ms) LTX_FAIL(LTX_ERR_ARG, "null output");
    for (int i = 0; i < 4; ++i) ms[i] = g_timing[i];
    return LTX_OK;
}

static thread_local int t_steps[2] = {0, 0};
extern "C" int ltx_pipeline_last_steps(int* executed, int* requested) {
    if (executed) *executed = t_steps[0];
    if (requested) *requested = t_steps[1];
    return LTX_OK;
}

namespace {
// Device scratch of the denoise loop, kept per (thread, device) across calls: per-call hipMalloc/hipFree (an implicit
// device sync each) and the coords rebuild + blocking upload cost ~1 ms per video for nothing.
struct PipeCache {
    DevBuf p_text, p_uncond, p_pert, stats, coords;
    DevBuf g_lat, g_emb, g_mask, g_pred, g_coords;      // the guidance branches of a step as one forward: inputs / predictions of all branches
    int coords_key[6] = {-1, -1, -1, -1, -1, -1};      // B, F, H, W, frame_rate, ratios packed
};
PipeCache& pipe_cache(int device) {
    thread_local std::map<int, PipeCache> caches;
    return caches[device];
}
struct PipeScratch {
    void *p_text = nullptr, *p_uncond = nullptr, *p_pert = nullptr, *coords = nullptr, *stats = nullptr;   // borrowed from PipeCache
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> step_ev;        // three per denoise step (start, after the forwards, after the update); every
                                            // event is owned exactly once and pushed here the moment it exists
    int new_event(hipEvent_t* out) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        step_ev.push_back(e); *out = e;
        return LTX_OK;
    }
    ~PipeScratch() {
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        for (auto e : step_ev) (void)hipEventDestroy(e);
    }
};
}  // namespace

extern "C" int ltx_pipeline_call(ltx_dit* dit, ltx_vae* vae, const ltx_pipeline_params* p,
                                 float* latents, const float* prompt_embeds, const float* prompt_mask,
                                 const float* neg_embeds, const float* neg_mask, const float* decode_noise,
                                 int B, int K, float* out_video, ltx_stream stream) {
    if (!dit || !p || !latents || !prompt_embeds || !prompt_mask) LTX_FAIL(LTX_ERR_ARG, "ltx_pipeline_call: null argument");
    if (!p->output_latent && (!vae || !out_video)) LTX_FAIL(LTX_ERR_ARG, "ltx_pipeline_call: decode requested without vae/out_video");
    // check_inputs (:313-365)
    if (p->height % 32 != 0 || p->width % 32 != 0) LTX_FAIL(LTX_ERR_ARG, "`height` and `width` must be divisible by 32");
    if (p->num_inference_steps < 1) LTX_FAIL(LTX_ERR_ARG, "num_inference_steps must be >= 1");
    const bool do_cfg = p->guidance_scale > 1.0f, do_stg = p->stg_scale > 0.0f;      // :304-310
    if (do_cfg && (!neg_embeds || !neg_mask)) LTX_FAIL(LTX_ERR_ARG, "classifier-free guidance needs negative embeddings and mask");
    if (p->stochastic_sampling && !p->step_noise) LTX_FAIL(LTX_ERR_ARG, "stochastic_sampling needs step_noise [steps,B,S*C]");
    hipStream_t s = (hipStream_t)stream;
    ltx_dit_config dc; LTX_TRY(ltx_dit_get_config(dit, &dc));
    int ts_ratio = 8, sp_ratio = 32;
    if (vae) { ltx_vae_config vc; LTX_TRY(ltx_vae_get_config(vae, &vc)); ts_ratio = vc.temporal_compression_ratio; sp_ratio = vc.spatial_compression_ratio; }
    const int F = (p->num_frames - 1) / ts_ratio + 1, H = p->height / sp_ratio, W = p->width / sp_ratio;   // :743-745
    const int S = F * H * W, C = dc.in_channels, L = dc.num_layers;
    const int64_t n = (int64_t)S * C;

    // skip blocks: permanent iff STG is off (:691-697)
    if (p->skip_block_list) {
        if (!do_stg) LTX_TRY(ltx_dit_set_skip_blocks(dit, p->skip_block_list, p->n_skip_blocks));
        else LTX_TRY(ltx_dit_set_skip_blocks(dit, nullptr, 0));
    }
    // sigmas / mu / timesteps (:750-791)
    const int N = p->num_inference_steps;
    std::vector<float> sig_in(N), sig(N + 1); std::vector<int64_t> ts(N);
    const bool custom = p->sigmas != nullptr;
    for (int i = 0; i < N; ++i)
        sig_in[i] = custom ? p->sigmas[i] : (N == 1 ? 1.0f : 1.0f + (1.0f / (float)N - 1.0f) * (float)i / (float)(N - 1));
    const float mu = custom ? 0.0f : ltx_calculate_shift(S, 256, 4096, 0.5f, 1.15f);      // static default cfg (scheduler.rs:638-639)
    LTX_TRY(ltx_sched_set_timesteps(sig_in.data(), N, mu, 1, 1.0f, p->shift_terminal, p->use_shift_terminal, sig.data(), ts.data()));

    PipeScratch sc;
    int cur_dev = 0; HIP_TRY(hipGetDevice(&cur_dev));
    PipeCache& pc = pipe_cache(cur_dev);
    const size_t pred_bytes = (size_t)B * n * sizeof(float);
    LTX_TRY(pc.p_text.ensure(pred_bytes)); sc.p_text = pc.p_text.p;
    if (do_cfg) { LTX_TRY(pc.p_uncond.ensure(pred_bytes)); sc.p_uncond = pc.p_uncond.p; }
    if (do_stg) { LTX_TRY(pc.p_pert.ensure(pred_bytes)); sc.p_pert = pc.p_pert.p; }
    LTX_TRY(pc.stats.ensure(64 * B)); sc.stats = pc.stats.p;
    // video_coords [B,S,3] (:798-847): rebuilt only when the geometry changes
    {
        const int key[6] = {B, F, H, W, p->frame_rate, ts_ratio * 1000 + sp_ratio};
        LTX_TRY(pc.coords.ensure((size_t)B * S * 3 * sizeof(float)));
        if (std::memcmp(key, pc.coords_key, sizeof(key)) != 0) {
            std::vector<float> vc((size_t)S * 3), all((size_t)B * S * 3);
            LTX_TRY(ltx_build_video_coords(F, H, W, p->frame_rate, ts_ratio, sp_ratio, vc.data()));
            for (int b = 0; b < B; ++b) std::memcpy(all.data() + (size_t)b * S * 3, vc.data(), sizeof(float) * S * 3);
            HIP_TRY(hipMemcpyAsync(pc.coords.p, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
            std::memcpy(pc.coords_key, key, sizeof(key));
        }
        sc.coords = pc.coords.p;
    }
    // The guidance branches of a step (:860-940: up to three B-row forwards on the same latents - negative prompt, prompt, prompt with
    // the STG blocks skipped) as ONE forward of nbr * B rows where that fits the DiT's 8-row pass: branch rows are independent, every
    // plan of a GEMM shape returns the same bits, and a row that skips a layer keeps its row partials (dit.hip) - so each branch's
    // prediction is, bit for bit, the separate forward's WHERE B * S and nbr * B * S rows take the same kernels and K partition
    // (above 512 rows per branch: C2 / C3 / C5; at a few hundred tokens the split-K factor, the one-row-per-block norms and the
    // deferred ff2 depend on the row count and the two forms agree to rounding only: tests/test_gpu_c3.py), at 2.5 instead of 3
    // rounds of the chip per attention launch and fewer one-round grids (one C3 step 61.8 -> 58.1 ms).  guidance_batch=0: the
    // reference's three calls.
    const int nbr = 1 + (do_cfg ? 1 : 0) + (do_stg ? 1 : 0);
    const bool gbatch = nbr > 1 && nbr * B <= 8 && ltx_opt().guidance_batch;
    const int Dt = dc.caption_channels;
    std::vector<float> g_slm;
    if (gbatch) {
        const int GB = nbr * B;
        LTX_TRY(pc.g_lat.ensure((size_t)GB * n * sizeof(float))); LTX_TRY(pc.g_pred.ensure((size_t)GB * n * sizeof(float)));
        LTX_TRY(pc.g_emb.ensure((size_t)GB * K * Dt * sizeof(float))); LTX_TRY(pc.g_mask.ensure((size_t)GB * K * sizeof(float)));
        LTX_TRY(pc.g_coords.ensure((size_t)GB * S * 3 * sizeof(float)));
        int r = 0;                                   // branch order: uncond, text, perturbed (the reference's call order)
        auto put = [&](const float* e, const float* mk) -> int {
            HIP_TRY(hipMemcpyAsync(pc.g_emb.as<float>() + (size_t)r * B * K * Dt, e, (size_t)B * K * Dt * sizeof(float), hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(pc.g_mask.as<float>() + (size_t)r * B * K, mk, (size_t)B * K * sizeof(float), hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(pc.g_coords.as<float>() + (size_t)r * B * S * 3, sc.coords, (size_t)B * S * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
            ++r; return LTX_OK;
        };
        if (do_cfg) { sc.p_uncond = pc.g_pred.as<float>() + (size_t)r * B * n; LTX_TRY(put(neg_embeds, neg_mask)); }
        sc.p_text = pc.g_pred.as<float>() + (size_t)r * B * n; LTX_TRY(put(prompt_embeds, prompt_mask));
        if (do_stg) {
            sc.p_pert = pc.g_pred.as<float>() + (size_t)r * B * n;
            g_slm.assign((size_t)L * GB, 0.0f);
            for (int i = 0; i < p->n_skip_blocks; ++i) { const int li = p->skip_block_list[i]; if (li >= 0 && li < L) for (int b = 0; b < B; ++b) g_slm[(size_t)li * GB + r * B + b] = 1.0f; }
            LTX_TRY(put(prompt_embeds, prompt_mask));
        }
    }
    for (auto& e : sc.ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(sc.ev[0], s));
    std::vector<float> stg_mask;
    if (do_stg) {                                    // :911-923
        stg_mask.assign((size_t)L * B, 0.0f);
        for (int i = 0; i < p->n_skip_blocks; ++i) { int li = p->skip_block_list[i]; if (li >= 0 && li < L) for (int b = 0; b < B; ++b) stg_mask[(size_t)li * B + b] = 1.0f; }
    }
    const float* coords = reinterpret_cast<const float*>(sc.coords);
    struct CtxScope { ltx_dit* d; ~CtxScope() { (void)ltx_dit_context_cache(d, 0); } } ctx_scope{dit};
    LTX_TRY(ltx_dit_context_cache(dit, 1));     // embeddings/masks are step-invariant inside one call
    // denoising loop (:860-994)
    const bool hooked = p->interrupt != nullptr || p->on_step != nullptr;
    bool stopped = false; int steps_run = 0;
    std::vector<hipEvent_t> done_ev;                        // e2 of every executed step (owned by sc.step_ev)
    for (int i = 0; i < N; ++i) {
        // interrupt / per-step hook (:861-865).  The loop only ENQUEUES work: the host stays at most one step ahead of the device
        // here, so that a flag raised while step k runs skips from step k + 2 on at the latest
        if (hooked) {
            if (done_ev.size() >= 2) HIP_TRY(hipEventSynchronize(done_ev[done_ev.size() - 2]));
            if (!stopped && p->on_step && p->on_step(p->on_step_user, i, N, ts[i]) != 0) stopped = true;
            if (stopped || (p->interrupt && *p->interrupt)) continue;
        }
        ++steps_run;
        float tvals[8]; for (int b = 0; b < 8; ++b) tvals[b] = (float)ts[i];       // Tensor::full(t as f32, (b,))
        hipEvent_t e0, e1, e2;
        LTX_TRY(sc.new_event(&e0)); LTX_TRY(sc.new_event(&e1)); LTX_TRY(sc.new_event(&e2));
        HIP_TRY(hipEventRecord(e0, s));
        if (gbatch) {
            for (int r = 0; r < nbr; ++r) HIP_TRY(hipMemcpyAsync(pc.g_lat.as<float>() + (size_t)r * B * n, latents, (size_t)B * n * sizeof(float), hipMemcpyDeviceToDevice, s));
            LTX_TRY(ltx_dit_forward(dit, pc.g_lat.p, pc.g_emb.p, tvals, pc.g_mask.as<float>(), nbr * B, S, K, F, H, W, nullptr, pc.g_coords.as<float>(),
                                    do_stg ? g_slm.data() : nullptr, LTX_F32, pc.g_pred.p, s));
        } else {
        if (do_cfg) LTX_TRY(ltx_dit_forward(dit, latents, neg_embeds, tvals, neg_mask, B, S, K, F, H, W, nullptr, coords, nullptr, LTX_F32, sc.p_uncond, s));
        LTX_TRY(ltx_dit_forward(dit, latents, prompt_embeds, tvals, prompt_mask, B, S, K, F, H, W, nullptr, coords, nullptr, LTX_F32, sc.p_text, s));
        if (do_stg) LTX_TRY(ltx_dit_forward(dit, latents, prompt_embeds, tvals, prompt_mask, B, S, K, F, H, W, nullptr, coords, stg_mask.data(), LTX_F32, sc.p_pert, s));
        }
        HIP_TRY(hipEventRecord(e1, s));
        // guidance mix (:941-962) + scheduler.step (:987; scheduler.rs:544-581): dt = sigma_next - sigma
        const float dts = sig[i + 1] - sig[i];
        if (p->stochastic_sampling)
            LTX_TRY(ltx_guidance_step_stochastic(sc.p_text, do_cfg ? sc.p_uncond : nullptr, do_stg ? sc.p_pert : nullptr, LTX_F32, latents, nullptr, B, n,
                                                 p->guidance_scale, p->guidance_rescale, p->stg_scale, sig[i], sig[i + 1],
                                                 p->step_noise + (size_t)i * B * n, sc.stats, s));
        else
            LTX_TRY(ltx_guidance_step(sc.p_text, do_cfg ? sc.p_uncond : nullptr, do_stg ? sc.p_pert : nullptr, LTX_F32, latents, nullptr, B, n,
                                      p->guidance_scale, p->guidance_rescale, p->stg_scale, dts, sc.stats, s));
        HIP_TRY(hipEventRecord(e2, s));
        done_ev.push_back(e2);
    }
    t_steps[0] = steps_run; t_steps[1] = N;
    HIP_TRY(hipEventRecord(sc.ev[1], s));
    if (!p->output_latent) {
        // unpack + denormalize + noise mix + decode + postprocess (:1002-1070)
        float tdec[8], nsc[8];
        for (int b = 0; b < 8; ++b) { tdec[b] = p->decode_timestep; nsc[b] = p->decode_noise_scale; }
        ltx_vae_config vc; LTX_TRY(ltx_vae_get_config(vae, &vc));
        const bool tc = vc.timestep_conditioning != 0;
        LTX_TRY(ltx_vae_decode_tokens(vae, latents, tc ? decode_noise : nullptr, nsc, tc ? tdec : nullptr, B, F, H, W, p->tiling, p->postprocess, out_video, s));
    }
    HIP_TRY(hipEventRecord(sc.ev[2], s));
    HIP_TRY(hipStreamSynchronize(s));
    float dit_ms = 0, gs_ms = 0, t = 0;
    for (size_t i = 0; i + 3 <= sc.step_ev.size(); i += 3) {
        HIP_TRY(hipEventElapsedTime(&t, sc.step_ev[i], sc.step_ev[i + 1])); dit_ms += t;
        HIP_TRY(hipEventElapsedTime(&t, sc.step_ev[i + 1], sc.step_ev[i + 2])); gs_ms += t;
    }
    g_timing[0] = dit_ms; g_timing[1] = gs_ms;
    HIP_TRY(hipEventElapsedTime(&g_timing[2], sc.ev[1], sc.ev[2]));
    HIP_TRY(hipEventElapsedTime(&g_timing[3], sc.ev[0], sc.ev[2]));
    return LTX_OK;
}

// ---- warm-up: everything a first call would otherwise do inside the caller's forward --------------------------------
// Runs one DiT forward and one decode of the given geometry on scratch buffers: GEMM plans are measured (or taken from a
// loaded plan file), workspaces are sized, code objects are loaded.  Afterwards calls of this geometry enqueue work only.
extern "C" int ltx_warmup(ltx_dit* dit, ltx_vae* vae, int B, int F, int H, int W, int K, ltx_stream stream) {
    if (B < 1 || B > 8 || F < 1 || H < 1 || W < 1 || K < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_warmup: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    const int S = F * H * W;
    float tvals[8] = {500.f, 500.f, 500.f, 500.f, 500.f, 500.f, 500.f, 500.f};
    if (dit) {
        ltx_dit_config dc; LTX_TRY(ltx_dit_get_config(dit, &dc));
        struct Tmp { DevBuf b; ~Tmp() { b.release(); } void* p() const { return b.p; } int ensure(size_t n) { return b.ensure(n); } } x, enc, out;
        LTX_TRY(x.ensure((size_t)B * S * dc.in_channels * sizeof(float)));
        LTX_TRY(enc.ensure((size_t)B * K * dc.caption_channels * sizeof(float)));
        LTX_TRY(out.ensure((size_t)B * S * dc.out_channels * sizeof(float)));
        HIP_TRY(hipMemsetAsync(x.p(), 0, (size_t)B * S * dc.in_channels * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(enc.p(), 0, (size_t)B * K * dc.caption_channels * sizeof(float), s));
        LTX_TRY(ltx_dit_forward(dit, x.p(), enc.p(), tvals, nullptr, B, S, K, F, H, W, nullptr, nullptr, nullptr, LTX_F32, out.p(), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (vae) {
        ltx_vae_config vc; LTX_TRY(ltx_vae_get_config(vae, &vc));
        const int To = (F - 1) * vc.temporal_compression_ratio + 1, Ho = H * vc.spatial_compression_ratio, Wo = W * vc.spatial_compression_ratio;
        struct Tmp { DevBuf b; ~Tmp() { b.release(); } void* p() const { return b.p; } int ensure(size_t n) { return b.ensure(n); } } z, vid;
        LTX_TRY(z.ensure((size_t)B * S * vc.latent_channels * sizeof(float)));
        LTX_TRY(vid.ensure((size_t)B * 3 * To * Ho * Wo * sizeof(float)));
        HIP_TRY(hipMemsetAsync(z.p(), 0, (size_t)B * S * vc.latent_channels * sizeof(float), s));
        LTX_TRY(ltx_vae_decode_tokens(vae, reinterpret_cast<const float*>(z.p()), nullptr, tvals, vc.timestep_conditioning ? tvals : nullptr, B, F, H, W, nullptr, 0,
                                      reinterpret_cast<float*>(vid.p()), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return LTX_OK;
}
