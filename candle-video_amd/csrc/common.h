// Shared device/host helpers for the ltxhip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LTX_WAVE 64

#include "errors.h"
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    ltx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return LTX_ERR_HIP; } } while (0)
#define LTX_CHECK_LAUNCH() HIP_TRY(hipGetLastError())

// ---- element traits: T in {float, bf16_t}; a "chunk" is 16 bytes ----
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int CHUNK = 4; };
template <> struct ElemTraits<bf16_t> { static constexpr int CHUNK = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// 16-byte vector of T viewed as raw words
union Chunk16 {
    u32x4 u;
    float f[4];
    bf16_t h[8];
};

template <typename T> __device__ __forceinline__ void chunk_to_f32(const Chunk16& c, float* out);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const Chunk16& c, float* out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c.f[i];
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const Chunk16& c, float* out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)c.h[i];
}
template <typename T> __device__ __forceinline__ void f32_to_chunk(const float* in, Chunk16& c);
template <> __device__ __forceinline__ void f32_to_chunk<float>(const float* in, Chunk16& c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c.f[i] = in[i];
}
template <> __device__ __forceinline__ void f32_to_chunk<bf16_t>(const float* in, Chunk16& c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c.h[i] = (bf16_t)in[i];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (~10 VALU):
// both activations sit in HBM-/MFMA-bound kernels' epilogues where the division was a visible VALU cost.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// 0.5 x (1 + tanh(u)),  u = sqrt(2/pi) (x + 0.044715 x^3)   (ltx_transformer.rs:214-226)
// 0.5 (1 + tanh(u)) = sigmoid(2u)  ->  x / (1 + exp(-2u));  exp overflow -> rcp(inf) = 0 (the x -> -inf limit).
// -2u log2(e) = x (A + B x^2): three multiply-adds in front of v_exp_f32 instead of six (the epilogue of ff1 spends 320 of these
// per lane and tile).  The scalar and the 4-wide form below run the same operations in the same order: the same bits.
constexpr float kGeluA = -2.0f * 0.7978845608028654f * 1.4426950408889634f, kGeluB = kGeluA * 0.044715f;
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float t = __builtin_fmaf(kGeluB, x * x, kGeluA);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * t));
}
// four values at once on packed f32 operations (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32; the transcendentals stay scalar)
__device__ __forceinline__ void gelu_tanh4(float* v) {
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2 x = {v[2 * h], v[2 * h + 1]};
        const f2 t = __builtin_elementwise_fma((f2){kGeluB, kGeluB}, x * x, (f2){kGeluA, kGeluA});
        const f2 a = x * t;
        f2 d = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        d = d + (f2){1.0f, 1.0f};
        const f2 y = x * (f2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        v[2 * h] = y[0]; v[2 * h + 1] = y[1];
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
