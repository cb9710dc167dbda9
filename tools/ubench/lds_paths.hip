// Per-CU rates of the operand paths a GEMM K-step uses, from L2-resident data (gfx950):
//   dma   : buffer_load_dwordx4 ... lds (1 KiB per wave instruction) into LDS, nothing else
//   gload : global_load_dwordx4 into registers (16 B per lane)
//   read  : ds_read_b128 fragment-style reads (conflict-free), nothing else
//   dma+read : both at once (do LDS-DMA writes and reads share the LDS port?)
// One 512-thread block per CU, ITER iterations; prints bytes per clock per CU for each mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned* sink, int iters, long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * 65536;          // 64 KiB per CU, re-read every iteration (L2 hits)
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, 65536, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    long long t0 = 0;
    __syncthreads();
    if (tid == 0) t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j)                                        // 8 pieces per wave = 64 KiB per block
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + (j * 8 + wave) * 1024), 16, lane * 16, (j * 8 + wave) * 1024, 0, 0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u32x4 v = *reinterpret_cast<const u32x4*>(base + (j * 8 + wave) * 1024 + lane * 16);
                acc += v;
            }
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 24; ++j) {                                     // 24 KiB per wave = 192 KiB per block
                const int row = (j * 16 + (lane & 15)) & 511, c = (lane >> 4) + 4 * (j & 1);
                u32x4 v = *reinterpret_cast<const u32x4*>(smem + 65536 + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                acc += v;
            }
        }
        if (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (tid == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (acc[0] == 0x12345678u) sink[tid] = acc[1] + acc[2] + acc[3];
}

template <int MODE> void run(const char* name, const unsigned char* src, unsigned* sink, long long* cyc, double bytes_per_iter) {
    const int iters = 2000, blocks = 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 131072, 0, src, sink, 10, cyc);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 131072, 0, src, sink, iters, cyc);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> h(blocks); CK(hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
    double c = 0; for (auto v : h) c += (double)v; c /= blocks;
    // s_memtime counts at 100 MHz on gfx9: convert with the event time instead
    printf("%-10s %8.3f ms  %7.1f GB/s per CU  (%.1f B per ns per CU; at 2.1 GHz = %.1f B/clk)\n", name, ms, bytes_per_iter * iters / (ms * 1e6), bytes_per_iter * iters / (ms * 1e6), bytes_per_iter * iters / (ms * 1e6) / 2.1);
}

int main() {
    unsigned char* src; unsigned* sink; long long* cyc;
    CK(hipMalloc(&src, 256 * 65536)); CK(hipMemset(src, 1, 256 * 65536)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&cyc, 256 * 8));
    run<0>("dma", src, sink, cyc, 65536.0);
    run<1>("gload", src, sink, cyc, 65536.0);
    run<2>("read", src, sink, cyc, 196608.0);
    run<3>("dma+read", src, sink, cyc, 65536.0 + 196608.0);
    return 0;
}
