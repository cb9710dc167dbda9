// Shared pieces of the GEMM / implicit-GEMM kernels: MFMA wrappers, 4-wide loads/stores and the fused
// epilogues (bias, GELU-tanh, gate*x+residual, residual, depth-to-space(+residual), unpatchify).
#pragma once
#include "common.h"
#include "kernels.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const Chunk16& w, const Chunk16& a, f32x4 acc) {
        bf16x8 wv = __builtin_bit_cast(bf16x8, w.u);
        bf16x8 av = __builtin_bit_cast(bf16x8, a.u);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x4 run(const Chunk16& w, const Chunk16& a, f32x4 acc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.f[j], a.f[j], acc, 0, 0, 0);
        return acc;
    }
};

template <typename T> __device__ __forceinline__ void load4(const T* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
    bf16x4 x = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
    f32x4 x = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = x;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) {
    bf16x4 x = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = x;
}

// Canonical sum of squares of one 4-column group (GemmArgs::rowsq) of the values AS STORED: for bf16 two v_dot2c_f32_bf16
// (columns 0,1 then 2,3 on top), for f32 a fixed fma chain.  A 128-column partial is the sequential sum, in ascending column
// order, of its 32 leaves.  Every producer (gemm_asm16's epilogue, rowsq_kernel) goes through these two functions.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ltx_rowsq_leaf(const bf16x4& p) {
    const bf16x2_t lo = {p[0], p[1]}, hi = {p[2], p[3]};
    float s = __builtin_amdgcn_fdot2_f32_bf16(lo, lo, 0.f, false);
    return __builtin_amdgcn_fdot2_f32_bf16(hi, hi, s, false);
}
__device__ __forceinline__ float ltx_rowsq_leaf(const f32x4& v) {
    float s = v[0] * v[0];
    s = __builtin_fmaf(v[1], v[1], s); s = __builtin_fmaf(v[2], v[2], s);
    return __builtin_fmaf(v[3], v[3], s);
}

template <typename T, int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& g, int m, int nb, float* v) {
    T* C = reinterpret_cast<T*>(g.C);
    if (g.bias) {
        float b[4];
        load4<T>(reinterpret_cast<const T*>(g.bias) + nb, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += b[i];
    }
    if constexpr (EPI == EPI_BIAS) {
        if (g.c_seg_shift) { const int sg = nb >> g.c_seg_shift; C += sg * g.c_seg_stride; nb -= sg << g.c_seg_shift; }
        store4<T>(C + (int64_t)m * g.ldc + nb, v);
    } else if constexpr (EPI == EPI_GELU) {
        gelu_tanh4(v);
        store4<T>(C + (int64_t)m * g.ldc + nb, v);
    } else if constexpr (EPI == EPI_GATE_RESID) {
        float r[4];
        load4<T>(reinterpret_cast<const T*>(g.resid) + (int64_t)m * g.ldr + nb, r);
        const float* gp = g.gate + (int64_t)(m / g.rows_per_batch) * g.gate_stride + nb;
        f32x4 gt = *reinterpret_cast<const f32x4*>(gp);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(gt[i], v[i], r[i]);       // ONE rounding, spelled out: every kernel that finishes these rows must agree
        store4<T>(C + (int64_t)m * g.ldc + nb, v);
    } else if constexpr (EPI == EPI_RESID) {
        float r[4];
        load4<T>(reinterpret_cast<const T*>(g.resid) + (int64_t)m * g.ldr + nb, r);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += r[i];
        store4<T>(C + (int64_t)m * g.ldc + nb, v);
    } else if constexpr (EPI == EPI_D2S) {
        // LtxVideoUpsampler3d (vae.rs:1090-1169): packed conv channel (c'*8 + st*4+sh*2+sw) was
        // re-ordered at weight-pack time to n' = s*Cf + c' so 4 consecutive n' are 4 consecutive
        // output channels of ONE output voxel.  Residual = d2s(x) tiled over channels (:1117-1121).
        int w = m % g.Wd; int t1 = m / g.Wd;
        int h = t1 % g.H; int t2 = t1 / g.H;
        int t = t2 % g.T; int b = t2 / g.T;
        int s = nb / g.Cf, co = nb - s * g.Cf;
        int st = s >> 2, sh = (s >> 1) & 1, sw = s & 1;
        int to = g.d2s_sp ? t : 2 * t + st - 1;  // (1, 2, 2): s = sh*2 + sw, frames kept as they are (:1225-1236)
        if (to < 0) return;                      // drop first frame (:1161)
        if (g.resid) {
            const T* x = reinterpret_cast<const T*>(g.resid) + (int64_t)m * g.Cin;
            const int nsub = g.d2s_sp ? 4 : 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += to_f32(x[((co + i) % g.Cr) * nsub + s]);
        }
        int64_t o = ((((int64_t)b * g.To + to) * g.Ho + (2 * h + sh)) * g.Wo + (2 * w + sw)) * g.Cf + co;
        store4<T>(C + o, v);
    } else if constexpr (EPI == EPI_UNPATCH) {
        // conv_out + unpatchify (vae.rs:1626-1654); channels re-ordered at pack time to
        // n' = (c*4 + off_h)*4 + off_w.  Output is f32 NCTHW [B, N/16, T, 4H, 4W].
        int w = m % g.Wd; int t1 = m / g.Wd;
        int h = t1 % g.H; int t2 = t1 / g.H;
        int t = t2 % g.T; int b = t2 / g.T;
        int c = nb >> 4, oh = (nb >> 2) & 3;
        int nc = g.N >> 4;
        if (g.post) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(v[i] * 0.5f + 0.5f, 0.0f), 1.0f) * 255.0f;
        }
        if (g.post == 2) {
            // RGB8 frames straight from the epilogue (main.rs:653-675: the CLI's frame conversion): u8 [B, T, 4H, 4W, 3], the
            // truncating cast of ltx_video_to_rgb8 on the same f32 values - 115 MB written at C2 instead of 458 MB + a 458 MB read
            unsigned char* o8 = reinterpret_cast<unsigned char*>(g.C) +
                                (((((int64_t)b * g.T + t) * (4 * g.H) + (4 * h + oh)) * (int64_t)(4 * g.Wd) + 4 * w) * nc + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) o8[i * nc] = (unsigned char)v[i];
            return;
        }
        int64_t o = ((((int64_t)b * nc + c) * g.T + t) * (4 * g.H) + (4 * h + oh)) * (int64_t)(4 * g.Wd) + 4 * w;
        store4<float>(reinterpret_cast<float*>(g.C) + o, v);
    }
}


}  // namespace

// hipFuncSetAttribute is per device: a per-kernel-instantiation bit mask of the devices already configured (several
// devices in one process).  The check and the call run under one lock and the bit is published only after the call
// succeeded: a second thread on the same device either waits for the attribute or finds it set, and a failed call is
// retried by the next launch instead of leaving the kernel permanently mis-configured.
#include <atomic>
#include <mutex>
static inline int ltx_set_max_dyn_smem(std::atomic<unsigned long long>& mask, const void* kern, int smem) {
    int dev = 0;
    const bool tracked = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    const unsigned long long bit = tracked ? 1ull << dev : 0ull;
    if (tracked && (mask.load(std::memory_order_acquire) & bit)) return LTX_OK;
    static std::mutex mu;                                   // one lock for all instantiations: taken once per (kernel, device)
    std::lock_guard<std::mutex> lock(mu);
    if (tracked && (mask.load(std::memory_order_acquire) & bit)) return LTX_OK;
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    if (tracked) mask.fetch_or(bit, std::memory_order_release);
    return LTX_OK;
}
