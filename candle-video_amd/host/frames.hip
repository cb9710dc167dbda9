// Frame output (include/ltxhip_frames.h), device side: RGB8 conversion on the GPU and the save-to-files wrappers.
// The file encoders themselves (PNG, GIF) are host-only code in host/frame_files.cpp.
#include <sys/stat.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ltxhip_frames.h"
#include "../csrc/common.h"

namespace {
__global__ void rgb8_kernel(const float* v, uint8_t* out, int B, int F, int H, int W) {
    const int64_t hw = (int64_t)H * W, n = (int64_t)B * F * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % hw; const int64_t bf = i / hw; const int f = (int)(bf % F); const int b = (int)(bf / F);
        const float* src = v + ((int64_t)b * 3 * F + f) * hw + p;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = src[(int64_t)c * F * hw];
            x = fminf(fmaxf(x, 0.0f), 255.0f);
            out[i * 3 + c] = (uint8_t)x;                         // truncating cast, like to_dtype(U8)
        }
    }
}
}  // namespace

extern "C" int ltx_video_to_rgb8(const float* video, int B, int F, int H, int W, uint8_t* rgb, ltx_stream stream) {
    if (!video || !rgb || B < 1 || F < 1 || H < 1 || W < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_video_to_rgb8: bad argument");
    const int64_t n = (int64_t)B * F * H * W;
    int64_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(rgb8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, video, rgb, B, F, H, W);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

extern "C" int ltx_save_frames_png(const float* video, int B, int F, int H, int W, const char* dir, int* n_written, ltx_stream stream) {
    if (!video || !dir) LTX_FAIL(LTX_ERR_ARG, "ltx_save_frames_png: null argument");
    if (n_written) *n_written = 0;
    struct stat sb;
    if (stat(dir, &sb) != 0 && mkdir(dir, 0755) != 0) LTX_FAIL(LTX_ERR_ARG, std::string("cannot create directory '") + dir + "'");
    const size_t bytes = (size_t)B * F * H * W * 3;
    uint8_t* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, bytes));
    int rc = ltx_video_to_rgb8(video, B, F, H, W, dev, stream);
    std::vector<uint8_t> host(rc == LTX_OK ? bytes : 0);
    if (rc == LTX_OK && hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) { ltx_set_error("frame copy failed"); rc = LTX_ERR_HIP; }
    if (rc == LTX_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { ltx_set_error("frame copy failed"); rc = LTX_ERR_HIP; }
    (void)hipFree(dev);
    if (rc != LTX_OK) return rc;
    for (int j = 0; j < B * F; ++j) {
        char name[64]; snprintf(name, sizeof(name), "/frame_%04d.png", j);
        LTX_TRY(ltx_write_png((std::string(dir) + name).c_str(), host.data() + (size_t)j * H * W * 3, W, H));
        if (n_written) *n_written = j + 1;
    }
    return LTX_OK;
}

extern "C" int ltx_save_video_gif(const float* video, int B, int F, int H, int W, const char* path, ltx_stream stream) {
    if (!video || !path) LTX_FAIL(LTX_ERR_ARG, "ltx_save_video_gif: null argument");
    const size_t bytes = (size_t)B * F * H * W * 3;
    uint8_t* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, bytes));
    int rc = ltx_video_to_rgb8(video, B, F, H, W, dev, stream);
    std::vector<uint8_t> host(rc == LTX_OK ? bytes : 0);
    if (rc == LTX_OK && hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) { ltx_set_error("frame copy failed"); rc = LTX_ERR_HIP; }
    if (rc == LTX_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { ltx_set_error("frame copy failed"); rc = LTX_ERR_HIP; }
    (void)hipFree(dev);
    if (rc != LTX_OK) return rc;
    return ltx_write_gif(path, host.data(), B * F, W, H, 4, 30);       // main.rs:697-698: speed 30, delay 4
}
