// ggml block formats -> dense tensors on the device (include/ltxhip_weights.h: ltx_gguf_dequantize).
// The reference's default text encoder dequantises every GGUF weight to f32 and multiplies in f32
// (quantized_t5_encoder.rs:53-72: QTensor::dequantize + matmul), so "dequantise once at load" IS its arithmetic.  The block
// layouts and the f32 operation order are those of ggml's reference dequantize_row_* functions (published format; candle's
// k_quants.rs port is not in the checkout): every product and difference below is a separate f32 rounding (no FMA).
#include "model_util.h"
#include "../../include/ltxhip_weights.h"

namespace {

__device__ __forceinline__ float h2f(const unsigned char* p) {            // little-endian IEEE half at any byte alignment
    const unsigned short b = (unsigned short)(p[0] | (p[1] << 8));
    _Float16 h; __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
__device__ __forceinline__ void scale_min_k4(int j, const unsigned char* q, int& sc, int& m) {
    if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
    else { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

template <typename T>
__global__ void gguf_dequant_kernel(const unsigned char* __restrict__ src, int type, int64_t numel, T* __restrict__ dst) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < numel; idx += (int64_t)gridDim.x * blockDim.x) {
        float y = 0.f;
        switch (type) {
            case 0: { float v; __builtin_memcpy(&v, src + idx * 4, 4); y = v; break; }
            case 1: y = h2f(src + idx * 2); break;
            case 30: { const unsigned int b = (unsigned int)(src[idx * 2] | (src[idx * 2 + 1] << 8)) << 16; y = __uint_as_float(b); break; }
            case 8: {                                             // Q8_0: half d; int8 qs[32]
                const unsigned char* b = src + (idx >> 5) * 34; const int i = (int)(idx & 31);
                y = __fmul_rn((float)(signed char)b[2 + i], h2f(b));
                break;
            }
            case 2: {                                             // Q4_0: half d; u8 qs[16]: low nibbles = elements 0..15, high = 16..31
                const unsigned char* b = src + (idx >> 5) * 18; const int i = (int)(idx & 31);
                const int q = i < 16 ? (b[2 + i] & 0xF) : (b[2 + i - 16] >> 4);
                y = __fmul_rn((float)(q - 8), h2f(b));
                break;
            }
            case 6: {                                             // Q5_0: half d; u8 qh[4]; u8 qs[16]
                const unsigned char* b = src + (idx >> 5) * 22; const int i = (int)(idx & 31);
                const unsigned int qh = (unsigned int)b[2] | ((unsigned int)b[3] << 8) | ((unsigned int)b[4] << 16) | ((unsigned int)b[5] << 24);
                const int j = i & 15;
                const int q = i < 16 ? ((b[6 + j] & 0xF) | (int)(((qh >> j) << 4) & 0x10)) : ((b[6 + j] >> 4) | (int)((qh >> (j + 12)) & 0x10));
                y = __fmul_rn((float)(q - 16), h2f(b));
                break;
            }
            case 12: {                                            // Q4_K: half d, dmin; u8 scales[12]; u8 qs[128]
                const unsigned char* b = src + (idx >> 8) * 144; const int i = (int)(idx & 255);
                const int j64 = i >> 6, half = (i >> 5) & 1, l = i & 31;
                int sc, m; scale_min_k4(2 * j64 + half, b + 4, sc, m);
                const int q = b[16 + j64 * 32 + l];
                const float d1 = __fmul_rn(h2f(b), (float)sc), m1 = __fmul_rn(h2f(b + 2), (float)m);
                y = __fsub_rn(__fmul_rn(d1, (float)(half ? q >> 4 : q & 0xF)), m1);
                break;
            }
            case 13: {                                            // Q5_K: half d, dmin; u8 scales[12]; u8 qh[32]; u8 qs[128]
                const unsigned char* b = src + (idx >> 8) * 176; const int i = (int)(idx & 255);
                const int j64 = i >> 6, half = (i >> 5) & 1, l = i & 31;
                int sc, m; scale_min_k4(2 * j64 + half, b + 4, sc, m);
                const int q = b[48 + j64 * 32 + l];
                const int v = (half ? q >> 4 : q & 0xF) + (((b[16 + l] >> (2 * j64 + half)) & 1) ? 16 : 0);
                const float d1 = __fmul_rn(h2f(b), (float)sc), m1 = __fmul_rn(h2f(b + 2), (float)m);
                y = __fsub_rn(__fmul_rn(d1, (float)v), m1);
                break;
            }
            case 14: {                                            // Q6_K: u8 ql[128]; u8 qh[64]; i8 scales[16]; half d
                const unsigned char* b = src + (idx >> 8) * 210; const int i = (int)(idx & 255);
                const int n = i >> 7, quarter = (i >> 5) & 3, l = i & 31;
                const unsigned char* ql = b + n * 64; const unsigned char h = b[128 + n * 32 + l];
                const int lo = (quarter & 1) ? ql[l + 32] : ql[l];
                const int q = ((quarter & 2) ? (lo >> 4) : (lo & 0xF)) | (((h >> (2 * quarter)) & 3) << 4);
                const int sc = (signed char)b[192 + n * 8 + (l >> 4) + 2 * quarter];
                y = __fmul_rn(__fmul_rn(h2f(b + 208), (float)sc), (float)(q - 32));
                break;
            }
        }
        dst[idx] = (T)y;
    }
}

}  // namespace

int ltx_launch_gguf_dequant(const void* blocks_dev, int ggml_type, int64_t numel, void* dst, int dst_dtype, hipStream_t s) {
    int be = 0, bb = 0;
    if (ltx_gguf_type_info(ggml_type, &be, &bb) != LTX_OK) LTX_FAIL(LTX_ERR_UNSUPPORTED, "gguf: ggml type " + std::to_string(ggml_type) + " is not read");
    if (numel <= 0 || numel % be) LTX_FAIL(LTX_ERR_ARG, "gguf: element count is not a multiple of the block size");
    int64_t blocks = cdiv64(numel, 256); if (blocks > 65536) blocks = 65536;
    if (dst_dtype == LTX_DT_BF16) hipLaunchKernelGGL((gguf_dequant_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char*)blocks_dev, ggml_type, numel, (bf16_t*)dst);
    else hipLaunchKernelGGL((gguf_dequant_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char*)blocks_dev, ggml_type, numel, (float*)dst);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

extern "C" int ltx_gguf_dequantize(int ggml_type, const void* blocks, int on_device, int64_t numel, ltx_dtype dst_dtype, void* dst, ltx_stream stream) {
    if (!blocks || !dst) LTX_FAIL(LTX_ERR_ARG, "ltx_gguf_dequantize: null argument");
    int be = 0, bb = 0;
    if (ltx_gguf_type_info(ggml_type, &be, &bb) != LTX_OK) LTX_FAIL(LTX_ERR_UNSUPPORTED, "ltx_gguf_dequantize: ggml type " + std::to_string(ggml_type) + " is not read");
    if (numel <= 0 || numel % be) LTX_FAIL(LTX_ERR_ARG, "ltx_gguf_dequantize: element count is not a multiple of the block size");
    const size_t nbytes = (size_t)(numel / be) * (size_t)bb;
    const void* src = blocks; void* tmp = nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (!on_device) {
        HIP_TRY(hipMalloc(&tmp, nbytes));
        if (hipMemcpyAsync(tmp, blocks, nbytes, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipFree(tmp); LTX_FAIL(LTX_ERR_HIP, "ltx_gguf_dequantize: upload failed"); }
        src = tmp;
    }
    int rc = ltx_launch_gguf_dequant(src, ggml_type, numel, dst, dst_dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32, s);
    if (tmp) {                                                    // the staging buffer must outlive the kernel
        if (hipStreamSynchronize(s) != hipSuccess && rc == LTX_OK) { ltx_set_error("ltx_gguf_dequantize: kernel failed"); rc = LTX_ERR_HIP; }
        (void)hipFree(tmp);
    }
    return rc;
}
