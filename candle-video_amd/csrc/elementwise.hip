// Small HBM-/latency-bound kernels around the GEMM/attention/conv cores.
#include "common.h"
#include "kernels.h"

namespace {

template <typename T> __device__ __forceinline__ float ld(const void* p, int64_t i) { return to_f32(reinterpret_cast<const T*>(p)[i]); }
__device__ __forceinline__ float ldd(const void* p, int dt, int64_t i) {
    return dt == LTX_DT_BF16 ? (float)reinterpret_cast<const bf16_t*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void std_(void* p, int dt, int64_t i, float v) {
    if (dt == LTX_DT_BF16) reinterpret_cast<bf16_t*>(p)[i] = (bf16_t)v; else reinterpret_cast<float*>(p)[i] = v;
}
__device__ __forceinline__ float round_dt(float v, int dt) { return dt == LTX_DT_BF16 ? (float)(bf16_t)v : v; }

#define GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// get_timestep_embedding (ltx_transformer.rs:271-309 / vae.rs:172-198): [cos | sin]
__global__ void sinusoid_kernel(void* out, int dt, TimeVec tv, const float* tab, int half, int round_t, float tmul) {
    const int n = tv.n * 2 * half;
    GRID_STRIDE(i, n) {
        int b = (int)(i / (2 * half)), j = (int)(i % (2 * half));
        float t = tv.t[b];
        if (round_t) t = round_dt(t, dt);                   // timestep.to_dtype(model_dtype) (:1051)
        if (tmul != 1.0f) t = round_dt(t * round_dt(tmul, round_t ? dt : LTX_DT_F32), round_t ? dt : LTX_DT_F32);
        float ang = t * tab[j < half ? j : j - half];
        std_(out, dt, i, j < half ? cosf(ang) : sinf(ang));
    }
}

__global__ void silu_kernel(const void* x, void* y, int64_t n, int dt) {
    GRID_STRIDE(i, n) std_(y, dt, i, silu_f(ldd(x, dt, i)));
}

__global__ void cast_kernel(const void* x, int xdt, void* y, int ydt, int64_t n) {
    GRID_STRIDE(i, n) std_(y, ydt, i, ldd(x, xdt, i));
}

__global__ void ada_kernel(float* out, const void* tables, const void* temb, int nl, int B, int width, int dt) {
    const int64_t n = (int64_t)nl * B * width;
    GRID_STRIDE(i, n) {
        int j = (int)(i % width); int64_t r = i / width; int b = (int)(r % B); int l = (int)(r / B);
        out[i] = ldd(tables, dt, (int64_t)l * width + j) + ldd(temb, dt, (int64_t)b * width + j);
    }
}

__global__ void mask_bias_kernel(float* out, const float* mask, int64_t n) {
    GRID_STRIDE(i, n) out[i] = (1.0f - mask[i]) * -10000.0f;      // ltx_transformer.rs:1063
}

// one wave per batch row: ordered compaction of the keys whose bias is above the drop threshold (kernels.h: k_count)
__global__ __launch_bounds__(64) void key_compact_kernel(const float* bias, int K, int* idx, int* count, float* bias_c) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* bb = bias + (int64_t)b * K;
    int n = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const float v = k < K ? bb[k] : -INFINITY;
        const bool keep = v > -5000.0f;
        const unsigned long long m = __ballot(keep);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) { idx[(int64_t)b * K + pos] = k; bias_c[(int64_t)b * K + pos] = v; }
        n += __popcll(m);
    }
    if (n == 0) {                                           // every key masked: the mask is a constant shift, keep them all
        for (int k = lane; k < K; k += 64) { idx[(int64_t)b * K + k] = k; bias_c[(int64_t)b * K + k] = bb[k]; }
        n = K;
    }
    for (int k = n + lane; k < K; k += 64) { idx[(int64_t)b * K + k] = 0; bias_c[(int64_t)b * K + k] = -INFINITY; }
    if (lane == 0) count[b] = n;
}

// 16 bytes per thread; one block per destination row
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* src, uint4* dst, const int* idx, const int* count, int B, int K, int row_vecs) {
    const int64_t row = blockIdx.x;                         // (l * B + b) * K + pos
    const int pos = (int)(row % K), b = (int)((row / K) % B);
    const int64_t base = row - pos;
    const bool live = pos < count[b];
    const uint4* sp = src + (base + (live ? idx[(int64_t)b * K + pos] : 0)) * row_vecs;
    uint4* dp = dst + row * row_vecs;
    for (int i = threadIdx.x; i < row_vecs; i += 256) dp[i] = live ? sp[i] : make_uint4(0u, 0u, 0u, 0u);
}

__global__ void skip_blend_kernel(void* h, const void* orig, TimeVec m, int64_t rows_per_batch, int D, int dt) {
    const int64_t n = (int64_t)m.n * rows_per_batch * D;
    GRID_STRIDE(i, n) {
        int b = (int)(i / (rows_per_batch * D));
        float mm = m.t[b];
        std_(h, dt, i, ldd(h, dt, i) * (1.0f - mm) + ldd(orig, dt, i) * mm);   // :1112-1123
    }
}

// y = h (.) (1 + scale[b]) in the model dtype: the second output of the norm fold's producers (GemmArgs::C2, gemm_asm.hip - the same
// expression on the rows as stored, so the same bits), for the one place a GEMM epilogue cannot write it: behind the skip-layer blend
__global__ void mod_scale_kernel(const void* h, const float* scale, int scale_stride, void* y, int B, int64_t rows_per_batch, int D, int dt) {
    const int64_t n = (int64_t)B * rows_per_batch * D;
    GRID_STRIDE(i, n) {
        const int b = (int)(i / (rows_per_batch * D)), col = (int)(i % D);
        std_(y, dt, i, ldd(h, dt, i) * (1.0f + scale[(int64_t)b * scale_stride + col]));
    }
}

// W'[n][k] = W[n][k] * (1 + scale[k]) in the model dtype (norm fold through the consumer's weights: dit.hip, DitTimeEntry::wfold)
__global__ void scale_cols_kernel(const void* W, const float* scale, void* out, int64_t N, int K, int dt) {
    const int64_t n = N * K;
    GRID_STRIDE(i, n) std_(out, dt, i, ldd(W, dt, i) * (1.0f + scale[(int)(i % K)]));
}

// ---- guidance + Euler (t2v_pipeline.rs:941-964, 227-243; scheduler.rs:576-581) ----
__device__ __forceinline__ float cfg_of(const GuidanceArgs& a, int64_t i, float& t) {
    t = ldd(a.text, a.pred_dtype, i);
    float c = t;
    if (a.uncond) { float u = ldd(a.uncond, a.pred_dtype, i); c = u + (t - u) * a.guidance_scale; }
    return c;
}
__global__ void guidance_stats_kernel(GuidanceArgs a) {
    // per-batch sums for the unbiased std of text and cfg (Tensor::var_keepdim)
    const int b = blockIdx.y;
    double st = 0, sst = 0, sc = 0, ssc = 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.n_per_batch; j += (int64_t)gridDim.x * blockDim.x) {
        float t; float c = cfg_of(a, (int64_t)b * a.n_per_batch + j, t);
        st += t; sst += (double)t * t; sc += c; ssc += (double)c * c;
    }
    __shared__ double sh[4][256];
    sh[0][threadIdx.x] = st; sh[1][threadIdx.x] = sst; sh[2][threadIdx.x] = sc; sh[3][threadIdx.x] = ssc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&a.stats[b * 4 + threadIdx.x], sh[threadIdx.x][0]);
}
__global__ void guidance_apply_kernel(GuidanceArgs a) {
    const int64_t n = (int64_t)a.B * a.n_per_batch;
    GRID_STRIDE(i, n) {
        float t; float c = cfg_of(a, i, t);
        if (a.uncond && a.guidance_rescale > 0.0f) {
            const int b = (int)(i / a.n_per_batch);
            const double N = (double)a.n_per_batch;
            const double* s = a.stats + b * 4;
            double vt = (s[1] - s[0] * s[0] / N) / (N - 1.0), vc = (s[3] - s[2] * s[2] / N) / (N - 1.0);
            float ratio = (float)sqrt(vt) / (float)sqrt(vc);
            c = (c * ratio) * a.guidance_rescale + c * (1.0f - a.guidance_rescale);
        }
        if (a.pert) c = c + (t - ldd(a.pert, a.pred_dtype, i)) * a.stg_scale;
        if (a.noise_out) a.noise_out[i] = c;
        if (a.latents) {
            if (a.step_noise) {
                const float x0 = a.latents[i] - a.sigma * c;
                a.latents[i] = (1.0f - a.sigma_next) * x0 + a.sigma_next * a.step_noise[i];
            } else {
                a.latents[i] = a.latents[i] + c * a.dt;
            }
        }
    }
}

// denormalize_latents (t2v_pipeline.rs:573-594) + decode-noise mix (:1049-1062); tokens are already channels-last
__global__ void denorm_mix_kernel(const float* lat, const float* mean, const float* sd, float inv_sf, const float* noise,
                                  TimeVec ns, void* out, int dt, int B, int64_t S, int C) {
    const int64_t n = (int64_t)B * S * C;
    GRID_STRIDE(i, n) {
        int c = (int)(i % C); int64_t r = i / C; int64_t s = r % S; int b = (int)(r / S);
        float x = lat[i] * sd[c] * inv_sf + mean[c];
        if (noise) {
            float sc = ns.t[b];
            x = x * (1.0f - sc) + noise[((int64_t)b * C + c) * S + s] * sc;
        }
        std_(out, dt, i, x);
    }
}

__global__ void ncthw_to_cl_kernel(const void* x, int xdt, void* y, int ydt, int B, int C, int64_t S) {
    const int64_t n = (int64_t)B * S * C;
    GRID_STRIDE(i, n) {
        int c = (int)(i % C); int64_t r = i / C; int64_t s = r % S; int b = (int)(r / S);
        std_(y, ydt, i, ldd(x, xdt, ((int64_t)b * C + c) * S + s));
    }
}

__global__ void blend_kernel(BlendArgs a) {
    // dst[..., x] = a[..., a_len - blend + x]*(1 - x/blend) + b[..., x]*(x/blend) along `dim`
    const int64_t n = (int64_t)a.BC * a.et * a.eh * a.ew;
    GRID_STRIDE(i, n) {
        int w = (int)(i % a.ew); int64_t r = i / a.ew; int h = (int)(r % a.eh); r /= a.eh; int t = (int)(r % a.et); int bc = (int)(r / a.et);
        const int off = a.a_len - a.blend;
        const int x = a.dim == 2 ? t : (a.dim == 3 ? h : w);
        const int ta = t + (a.dim == 2 ? off : 0), ha = h + (a.dim == 3 ? off : 0), wa = w + (a.dim == 4 ? off : 0);
        float wgt = (float)x * a.inv_blend;
        int64_t ib = (((int64_t)bc * a.bt + t) * a.bh + h) * a.bw + w;
        int64_t ia = (((int64_t)bc * a.at + ta) * a.ah + ha) * a.aw + wa;
        int64_t id = (((int64_t)bc * a.dt + a.ot + t) * a.dh + a.oh + h) * a.dw + a.ow + w;
        a.dst[id] = a.a[ia] * (1.0f - wgt) + a.b[ib] * wgt;
    }
}

__global__ void copy_window_kernel(const float* src, int t, int h, int w, float* dst, int T, int H, int W, int BC,
                                   int st, int sh, int sw, int ot, int oh, int ow) {
    const int64_t n = (int64_t)BC * st * sh * sw;
    GRID_STRIDE(i, n) {
        int x = (int)(i % sw); int64_t r = i / sw; int y = (int)(r % sh); r /= sh; int z = (int)(r % st); int bc = (int)(r / st);
        dst[(((int64_t)bc * T + ot + z) * H + oh + y) * W + ow + x] = src[(((int64_t)bc * t + z) * h + y) * w + x];
    }
}

__global__ void postprocess_kernel(float* x, int64_t n) {
    GRID_STRIDE(i, n) x[i] = fminf(fmaxf(x[i] * 0.5f + 0.5f, 0.0f), 1.0f) * 255.0f;   // t2v_pipeline.rs:146-155
}

// LtxVideoResnetBlock3d::maybe_inject_noise (vae.rs:741-753) on a channels-last tensor: y = x + noise[h, w] * scale[c], the
// reference's roundings in the model dtype (noise cast first, the product rounded, the sum rounded), then optionally the block's
// shortcut (vae.rs:811-819: result = h + x, one more rounding)
template <typename T>
__global__ void noise_inject_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ noise, const T* __restrict__ scale,
                                    const T* __restrict__ resid, int64_t n, int C, int64_t HW) {
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C); const int64_t hw = (i / C) % HW;
        const float p = to_f32((T)(to_f32((T)noise[hw]) * to_f32(scale[c])));
        T v = (T)(to_f32(x[i]) + p);
        if (resid) v = (T)(to_f32(v) + to_f32(resid[i]));
        y[i] = v;
    }
}

inline dim3 grid_for(int64_t n) { int64_t b = cdiv64(n, 256); if (b > 16384) b = 16384; if (b < 1) b = 1; return dim3((unsigned)b); }

}  // namespace

int ltx_launch_sinusoid(void* out, int dtype, const TimeVec& tv, const float* tab, int half, int round_t, float tmul, hipStream_t s) {
    if (tv.n < 1 || tv.n > LTX_MAX_BATCH) LTX_FAIL(LTX_ERR_ARG, "batch must be 1..16");
    hipLaunchKernelGGL(sinusoid_kernel, grid_for(tv.n * 2 * half), dim3(256), 0, s, out, dtype, tv, tab, half, round_t, tmul);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_noise_inject(const void* x, void* y, const float* noise, const void* scale, const void* resid, int64_t rows, int C, int64_t HW, int dtype, hipStream_t s) {
    if (!x || !y || !noise || !scale || rows < 1 || C < 1 || HW < 1 || rows % HW != 0) LTX_FAIL(LTX_ERR_ARG, "noise_inject: bad arguments");
    const int64_t n = rows * C;
    if (dtype == LTX_DT_BF16) hipLaunchKernelGGL(noise_inject_kernel<bf16_t>, grid_for(n), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, noise, (const bf16_t*)scale, (const bf16_t*)resid, n, C, HW);
    else hipLaunchKernelGGL(noise_inject_kernel<float>, grid_for(n), dim3(256), 0, s, (const float*)x, (float*)y, noise, (const float*)scale, (const float*)resid, n, C, HW);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_silu(const void* x, void* y, int64_t n, int dtype, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, grid_for(n), dim3(256), 0, s, x, y, n, dtype);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_cast(const void* x, int xdt, void* y, int ydt, int64_t n, hipStream_t s) {
    if (n <= 0) return LTX_OK;
    hipLaunchKernelGGL(cast_kernel, grid_for(n), dim3(256), 0, s, x, xdt, y, ydt, n);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_ada(float* out, const void* tables, const void* temb, int nl, int B, int width, int dtype, hipStream_t s) {
    hipLaunchKernelGGL(ada_kernel, grid_for((int64_t)nl * B * width), dim3(256), 0, s, out, tables, temb, nl, B, width, dtype);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_mask_bias(float* out, const float* mask, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(mask_bias_kernel, grid_for(n), dim3(256), 0, s, out, mask, n);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_key_compact(const float* bias, int B, int K, int* idx, int* count, float* bias_c, hipStream_t s) {
    if (!bias || !idx || !count || !bias_c || B < 1 || K < 1) LTX_FAIL(LTX_ERR_ARG, "key_compact: bad argument");
    hipLaunchKernelGGL(key_compact_kernel, dim3((unsigned)B), dim3(64), 0, s, bias, K, idx, count, bias_c);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_gather_rows(const void* src, void* dst, const int* idx, const int* count, int nl, int B, int K, int row_bytes, hipStream_t s) {
    if (!src || !dst || !idx || !count || nl < 1 || B < 1 || K < 1 || row_bytes < 16 || row_bytes % 16 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        LTX_FAIL(LTX_ERR_ARG, "gather_rows: bad argument");
    if ((int64_t)nl * B * K > 0x7fffffffLL) LTX_FAIL(LTX_ERR_ARG, "gather_rows: too many rows");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(nl * B * K)), dim3(256), 0, s, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), idx, count, B, K, row_bytes / 16);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_skip_blend(void* h, const void* orig, const TimeVec& m, int64_t rows_per_batch, int D, int dtype, hipStream_t s) {
    hipLaunchKernelGGL(skip_blend_kernel, grid_for((int64_t)m.n * rows_per_batch * D), dim3(256), 0, s, h, orig, m, rows_per_batch, D, dtype);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_mod_scale(const void* h, const float* scale, int scale_stride, void* y, int B, int64_t rows_per_batch, int D, int dtype, hipStream_t s) {
    hipLaunchKernelGGL(mod_scale_kernel, grid_for((int64_t)B * rows_per_batch * D), dim3(256), 0, s, h, scale, scale_stride, y, B, rows_per_batch, D, dtype);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_scale_cols(const void* W, const float* scale, void* out, int64_t N, int K, int dtype, hipStream_t s) {
    hipLaunchKernelGGL(scale_cols_kernel, grid_for(N * K), dim3(256), 0, s, W, scale, out, N, K, dtype);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_guidance_step(const GuidanceArgs& a, hipStream_t s) {
    if (!a.text || a.B < 1 || a.n_per_batch < 1) LTX_FAIL(LTX_ERR_ARG, "guidance: text prediction required");
    const bool rescale = a.uncond && a.guidance_rescale > 0.0f;
    if (rescale) {
        if (!a.stats) LTX_FAIL(LTX_ERR_ARG, "guidance: rescale needs a stats workspace");
        HIP_TRY(hipMemsetAsync(a.stats, 0, sizeof(double) * 4 * a.B, s));
        int64_t bx = cdiv64(a.n_per_batch, 256 * 8); if (bx > 1024) bx = 1024; if (bx < 1) bx = 1;
        hipLaunchKernelGGL(guidance_stats_kernel, dim3((unsigned)bx, (unsigned)a.B), dim3(256), 0, s, a);
        LTX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(guidance_apply_kernel, grid_for((int64_t)a.B * a.n_per_batch), dim3(256), 0, s, a);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_denorm_mix(const float* lat, const float* mean, const float* std_dev, float inv_sf, const float* noise,
                          const TimeVec& nscale, void* out, int dtype, int B, int64_t S, int C, hipStream_t s) {
    hipLaunchKernelGGL(denorm_mix_kernel, grid_for((int64_t)B * S * C), dim3(256), 0, s, lat, mean, std_dev, inv_sf, noise, nscale, out, dtype, B, S, C);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_ncthw_to_cl(const void* x, int xdt, void* y, int ydt, int B, int C, int64_t S, hipStream_t s) {
    hipLaunchKernelGGL(ncthw_to_cl_kernel, grid_for((int64_t)B * S * C), dim3(256), 0, s, x, xdt, y, ydt, B, C, S);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_blend(const BlendArgs& a, hipStream_t s) {
    if (a.blend <= 0 || a.et <= 0 || a.eh <= 0 || a.ew <= 0) return LTX_OK;
    BlendArgs b = a;
    b.inv_blend = 1.0f / (float)a.blend;
    hipLaunchKernelGGL(blend_kernel, grid_for((int64_t)a.BC * a.et * a.eh * a.ew), dim3(256), 0, s, b);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_copy_window(const float* src, int t, int h, int w, float* dst, int T, int H, int W, int BC,
                           int st, int sh, int sw, int ot, int oh, int ow, hipStream_t s) {
    hipLaunchKernelGGL(copy_window_kernel, grid_for((int64_t)BC * st * sh * sw), dim3(256), 0, s, src, t, h, w, dst, T, H, W, BC, st, sh, sw, ot, oh, ow);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
int ltx_launch_postprocess(float* x, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(postprocess_kernel, grid_for(n), dim3(256), 0, s, x, n);
    LTX_CHECK_LAUNCH(); return LTX_OK;
}
