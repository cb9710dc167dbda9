// Host-side helpers shared by dit.hip / vae.hip / pipeline.hip: device buffers, weight lookup/upload.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <cmath>
#include "common.h"
#include "kernels.h"
#include "../../include/ltxhip.h"

// thread-local "the last failure on this thread was an out-of-memory hipMalloc" (DevBuf::ensure); cleared by the reader
void ltx_note_oom();
bool ltx_take_oom();

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return LTX_OK;
        if (p) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(p)); p = nullptr; bytes = 0; }
        const hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            if (e == hipErrorOutOfMemory) ltx_note_oom();     // callers that can shrink their request tell this failure from every other one
            (void)hipGetLastError();
            ltx_set_error(std::string("hipMalloc(") + std::to_string(n) + " bytes): " + hipGetErrorString(e)); return LTX_ERR_HIP;
        }
        bytes = n;
        return LTX_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct LinearW {
    void* w = nullptr;   // [out, in] model dtype
    void* b = nullptr;   // [out] model dtype or null
    int in = 0, out = 0;
    // tile-contiguous second copy for gemm_ring.hip (GemmArgs::Wp), made by ltx_linear on the first call with at most 512 rows
    // and freed with the last copy of this struct; wp_tried: an allocation that failed is not retried
    mutable std::shared_ptr<void> wp;
    mutable bool wp_tried = false;
};

struct WeightMap {
    std::map<std::string, const ltx_weight*> m;
    WeightMap(const ltx_weight* w, size_t n) { for (size_t i = 0; i < n; ++i) if (w[i].name) m[w[i].name] = &w[i]; }
    const ltx_weight* find(const std::string& k) const { auto it = m.find(k); return it == m.end() ? nullptr : it->second; }
};

static inline int64_t ltx_numel(const ltx_weight* w) {
    int64_t n = 1;
    for (int i = 0; i < w->ndim; ++i) n *= w->shape[i];
    return n;
}

// Copy weight `w` (host or device, any dtype) into device memory `dst` as `dst_dtype` (no layout change).
int ltx_upload_cast(const ltx_weight* w, void* dst, int dst_dtype, int64_t expect_numel, const std::string& name);
// Allocate + upload a tensor by name; fails with LTX_ERR_MISSING_WEIGHT when absent (unless optional).
int ltx_load_tensor(const WeightMap& wm, const std::string& name, int64_t numel, int dtype, void** out, bool optional = false);
int ltx_load_linear(const WeightMap& wm, const std::string& prefix, int in, int out, int dtype, LinearW* l);
// stage a source tensor on the device in its own dtype (returns temp pointer to free, or the original device ptr)
int ltx_stage_src(const ltx_weight* w, const void** dev_src, void** temp_to_free);

// Frequency tables of the two get_timestep_embedding flavours, with the reference's f32 roundings:
//   DiT (ltx_transformer.rs:288-290): 1 / 10000^(i/128);   VAE (vae.rs:172-198): exp(-ln(1e4) / 128 * i)
inline void ltx_sinusoid_table(int vae_flavour, float tab[128]) {
    if (!vae_flavour) {
        for (int i = 0; i < 128; ++i) { const float ex = (float)i / 128.0f; const float pw = (float)std::pow(10000.0, (double)ex); tab[i] = 1.0f / pw; }
    } else {
        const float coef = (float)(-std::log(10000.0) / 128.0);
        for (int i = 0; i < 128; ++i) { const float x = (float)i * coef; tab[i] = (float)std::exp((double)x); }
    }
}

// y[M,N] = epi(x[M,K] @ W^T + b)
int ltx_linear(const LinearW& l, const void* x, int lda, void* y, int ldc, int M, int dtype, int epi, hipStream_t s,
               const void* resid = nullptr, int ldr = 0, const float* gate = nullptr, int gate_stride = 0, int rows_per_batch = 1,
               float* rowsq = nullptr);

// conv weight repack [O,I,kt,kh,kw] -> [tap][n'][I] with output-channel permutation (vae.hip)
enum { LTX_PERM_NONE = 0, LTX_PERM_D2S = 1, LTX_PERM_UNPATCH = 2 };
int ltx_pack_conv(const void* src_dev, int sdt, void* dst, int ddt, int O, int I, int ntaps, int mode, int Cf, hipStream_t s);
