// Frame output (include/ltxhip_frames.h): device RGB8 conversion + host PNG writer (zlib deflate, CRC per chunk).
#include <sys/stat.h>
#include <zlib.h>

#include <cstdio>
#include <cstring>
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

void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<unsigned char>& png, const char* type, const unsigned char* data, size_t n) {
    put32(png, (uint32_t)n);
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(n + 4)));
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

extern "C" int ltx_write_png(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width < 1 || height < 1) LTX_FAIL(LTX_ERR_ARG, "ltx_write_png: bad argument");
    const size_t row = (size_t)width * 3;
    std::vector<unsigned char> raw((row + 1) * height);
    for (int y = 0; y < height; ++y) { raw[y * (row + 1)] = 0; memcpy(&raw[y * (row + 1) + 1], rgb + y * row, row); }   // filter 0
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) LTX_FAIL(LTX_ERR_ARG, "ltx_write_png: deflate failed");
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);    // 8-bit, colour type 2 (RGB)
    chunk(png, "IHDR", ihdr.data(), ihdr.size());
    chunk(png, "IDAT", z.data(), zn);
    chunk(png, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_write_png: cannot open '") + path + "'");
    const size_t w = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    if (w != png.size()) LTX_FAIL(LTX_ERR_ARG, std::string("ltx_write_png: short write to '") + path + "'");
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
