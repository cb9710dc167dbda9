// Does a wave that issues LDS-DMA pieces slow the MFMA stream of the OTHER wave on its SIMD?  (gfx950)
// 512-thread block per CU: waves 0-3 (one per SIMD) only issue LDS-DMA pieces (16 per round, then vmcnt(0)); waves 4-7
// (their SIMD partners) only issue independent MFMAs (64 per round).  Modes: mfma alone, dma alone, both.
// Prints the time per round of each role (s_memrealtime, 100 MHz) and the MFMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__global__ __launch_bounds__(512) void k(const unsigned char* src, float* sink, int rounds, int mode, int pieces) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * 65536;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, 65536, 0x00020000);
    if (wave < 4) {
        if (!(mode & 2)) return;
        for (int it = 0; it < rounds; ++it) {
            for (int j = 0; j < pieces; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + ((j & 15) * 4 + wave) * 1024), 16, lane * 16, ((j & 15) * 4 + wave) * 1024, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        if (!(mode & 1)) return;
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
        for (int it = 0; it < rounds; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 123.456f) sink[tid] = s;
    }
}

static float run(const unsigned char* src, float* sink, int mode, int pieces) {
    const int rounds = 4000;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, src, sink, 10, mode, pieces);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, src, sink, rounds, mode, pieces);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e6f / rounds;     // ns per round
}

int main() {
    unsigned char* src; float* sink;
    CK(hipMalloc(&src, 256 * 65536)); CK(hipMemset(src, 1, 256 * 65536)); CK(hipMalloc(&sink, 4096));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const float mfma = run(src, sink, 1, 16);
    printf("64 MFMA(16x16x32) per round, one wave per SIMD, alone : %7.1f ns per round (%.0f TF/s chip)\n", mfma, 64.0 * 16384 * 1024 / mfma / 1e3);
    for (int pieces : {4, 8, 16, 32}) {
        const float dma = run(src, sink, 2, pieces);
        const float both = run(src, sink, 3, pieces);
        printf("%2d LDS-DMA pieces per round per partner wave: dma alone %7.1f ns/round (%5.1f GB/s per CU); both %7.1f ns/round\n", pieces, dma, pieces * 4 * 1024.0 / dma, both);
    }
    return 0;
}
