// Self-attention at head_dim 128, q prescaled by scale*log2(e), no key bias (the 13B DiT, BASELINE C5: S = 17556, 32 heads):
// one wave per SIMD, 64 queries per wave (256-query workgroups; 32 per wave in the 128-query form that serves the queries
// left over), the whole tile loop one generated asm statement (tools/gen_attn_q128_asm.py ->
// attn_q128_loop.inc).  Round 3: the structure of attn_q64.hip carried to head_dim 128, where it pays more - per 64-key
// tile and 32 queries a wave issues the same 32 MFMAs of 32x32x16 as the 256-query form at head_dim 64 but HALF the exponentials, so the
// matrix pipe, not the vector issue port, bounds the loop.  attn_bf16_kernel<128> (attention.hip) runs a tile's phases back
// to back in each wave and leaves the overlap to a second workgroup on the CU: 0.44-0.46 of the bf16 MFMA peak.
//
//   * workgroup = 4 waves = 128 queries of one head; K/V tiles of 64 keys (32 KiB) arrive by buffer LDS-DMA into a ring of
//     four slots (128 KiB), tile t+4 issued in iteration t, counted vmcnt, one barrier per tile; LDS images and fragment maps
//     are attention.hip's at KCPR = VCPR = 16 (K rows chunk-XOR-swizzled by row % 16, V read through ds_read_b64_tr_b16);
//   * FIXED max from the first key tile, row sums on the matrix pipe, exactly as attn_q64.hip; a key count that is not a
//     multiple of 64 (S = 17556 = 274 x 64 + 20) is handled INSIDE the pipeline: the scores of the last tile's missing keys are
//     set to -inf between the QK and PV gaps of the last-but-one iteration;
//   * overflow (a later score beyond the first tile's maximum by ~100 in log2 units; never seen on model activations): every
//     block checks its row sums and raises a device flag stamped with the launch's ticket; the launcher follows this kernel
//     with attn_bf16_kernel<128> gated on that flag (exact running max; it returns at once when the flag is not this launch's).
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "options.h"

namespace {

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
#ifndef Q128_LOOP_INC
#define Q128_LOOP_INC "attn_q128_loop.inc"
#endif
#include Q128_LOOP_INC

constexpr int BKV = 64, KROW = 256, VROW = 256, TILE_BYTES = BKV * (KROW + VROW), NSLOT = 4;

// One workgroup: queries [q_first, q_first + 128 QB) of (batch b, head) against all keys.  QB = 32-query blocks per wave.
template <int QB>
__device__ __forceinline__ void q128_block(const AttnArgs& a, unsigned char* smem, int b, int head, int q_first, int* flag, int ticket) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = q_first + wave * 32 * QB;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + head * 128;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Sk * a.ldk + head * 128;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Sk * a.ldv + head * 128;
    const int nt = (a.Sk + BKV - 1) / BKV;
    const int rem = a.Sk - (nt - 1) * BKV;                 // keys of the last tile (64: nothing to mask)

    // LDS-DMA geometry: pieces of 4 rows x 256 B; wave w issues pieces 4w .. 4w+3 of K and of V (rows 16w .. 16w+15); the
    // tile's bank swizzles are applied to the SOURCE chunk; rows past Sk are out of the buffer's range -> zeros
    u32x8 dma_u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 4 + (lane >> 4), pc = lane & 15;
        dma_u[j] = (uint32_t)row * (uint32_t)a.ldk * 2u + (uint32_t)(pc ^ (row & 15)) * 16u;                 // kswz<16>
        dma_u[4 + j] = (uint32_t)row * (uint32_t)a.ldv * 2u + (uint32_t)(pc ^ ((row & 3) << 2)) * 16u;      // vswz<16>
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    u32x8 kbase, kbase_hi;
    u32x4 trbase, trbase_hi;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                          // K fragment: row kb*32 + r, chunk 2 ks + h (rows r, r + 32 swizzle alike)
        kbase[ks] = smem_base + (uint32_t)(r * KROW + (((2 * ks + h) ^ (r & 15)) << 4));
        kbase_hi[ks] = kbase[ks] + 65536u;
    }
    {
        const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int row = 4 * h + trq, cv = d * 4 + trdh * 2 + (trp >> 1);
            trbase[d] = smem_base + (uint32_t)(row * VROW + ((cv ^ ((row & 3) << 2)) << 4) + (trp & 1) * 8);
            trbase_hi[d] = trbase[d] + 65536u;
        }
    }
    // constant A operand of the row-sum MFMA (attn_q64.hip): 1 where the parities of row and of k's group of 8 agree
    u32x4 ones_u;
    {
        const uint32_t one2 = ((lane & 1) == ((lane >> 4) & 1)) ? 0x3f803f80u : 0u;
        ones_u = (u32x4){one2, one2, one2, one2};
    }
    u32x2 qoff, ooff;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = q0 + 32 * (qb < QB ? qb : 0) + r;
        const int qc = qr > a.Sq - 1 ? a.Sq - 1 : qr;
        qoff[qb] = (uint32_t)qc * (uint32_t)a.ldq * 2u + 16u * h;
        ooff[qb] = qr < a.Sq ? (uint32_t)qr * (uint32_t)a.ldo * 2u + 16u * h : 0x80000000u;     // rows past Sq: out of range, dropped
    }
    auto words = [](const void* p, uint32_t bytes) {
        const uint64_t u = (uint64_t)(uintptr_t)p;
        return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) & 0xffffu,
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    };
    const u32x4 rk = words(K, (uint32_t)(a.Sk - 1) * (uint32_t)a.ldk * 2u + 256u);
    const u32x4 rv = words(V, (uint32_t)(a.Sk - 1) * (uint32_t)a.ldv * 2u + 256u);
    const u32x4 rq = words(Q, (uint32_t)(a.Sq - 1) * (uint32_t)a.ldq * 2u + 256u);
    const u32x4 ro = words(reinterpret_cast<bf16_t*>(a.o) + (int64_t)b * a.Sq * a.ldo + head * 128, (uint32_t)(a.Sq - 1) * (uint32_t)a.ldo * 2u + 256u);
    const uint32_t kstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldk * 2u));
    const uint32_t vstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldv * 2u));
    const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * 4096u));
    const uint32_t sel = (lane & 16) ? 1u : 0u, key0 = (uint32_t)(4 * h);
    const int nt_u = __builtin_amdgcn_readfirstlane(nt), rem_u = __builtin_amdgcn_readfirstlane(rem);
    bool bad;
    if constexpr (QB == 2) {
        f32x8 la;
        q128_full_qb2(la, ones_u, kbase, kbase_hi, trbase, trbase_hi, dma_u, qoff, ooff, sel, key0, rk, rv, rq, ro, nt_u, rem_u, kstep, vstep, ldsw);
        const float l0 = (lane & 16) ? la[1] : la[0], l1 = (lane & 16) ? la[5] : la[4];
        bad = !(l0 < 0x1p100f) || !(l1 < 0x1p100f);
    } else {
        f32x4 la;
        q128_full(la, ones_u, kbase, trbase, kbase_hi, trbase_hi, dma_u, qoff[0], ooff[0], sel, key0, rk, rv, rq, ro, nt_u, rem_u, kstep, vstep, ldsw);
        const float l0 = (lane & 16) ? la[1] : la[0];
        bad = !(l0 < 0x1p100f);
    }
    // l beyond 2^100 (or NaN): some p overflowed or came close - the stored rows are then not to be trusted; the gated exact
    // kernel that follows recomputes the whole launch
#ifndef Q128_NO_FALLBACK      // timing ablations (garbage results) must not start the exact pass
    if (bad) atomicExch(flag, ticket);
#else
    (void)bad; (void)flag; (void)ticket;
#endif
}

// Grid: per batch, first heads * nbig big blocks (256 queries: [i * 256, +256)), then heads * nsmall small blocks (128 queries:
// [nbig * 256 + i * 128, +128)).  Inside each class blocks L, L + 8 share an XCD, which is given whole heads (attention.hip).
__global__ __launch_bounds__(256, 1) void attn_q128_kernel(const AttnArgs a, int nbig, int nsmall, int* flag, int ticket) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int per_b = a.heads * (nbig + nsmall);
    int L = blockIdx.x;
    const int b = L / per_b; L -= b * per_b;
    const bool big = L < a.heads * nbig;
    if (!big) L -= a.heads * nbig;
    const int n = big ? nbig : nsmall;
    int head, qb;
    if (a.xcd_heads) { const int xcd = L & 7, j = L >> 3; head = xcd + 8 * (j / n); qb = j % n; }
    else { head = L / n; qb = L - head * n; }
    if (big) q128_block<2>(a, smem, b, head, qb * 256, flag, ticket);
    else q128_block<1>(a, smem, b, head, nbig * 256 + qb * 128, flag, ticket);
}

std::mutex g_q128_mu;
std::map<int, int*> g_q128_flag;                 // per device: 64 ints, zero-initialised; word (ticket & 63) holds the ticket of an overflowing launch
std::atomic<int> g_q128_ticket{1};

}  // namespace

// shapes the generated loop serves: prescaled bf16, head_dim 128, at least two key tiles, 32-bit buffer offsets
bool ltx_attention_q128_fits(const AttnArgs& a) {
    if (ltx_opt().attn_off & LTX_ATTN_Q128) return false;      // attn_off=q128: attn_bf16_kernel<128> (A/B aid)
    if (a.hd != 128 || !a.q_prescaled || a.bias || a.Sk < 128 || a.Sq < 1) return false;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 8 || ((uintptr_t)a.q & 15) || ((uintptr_t)a.k & 15) || ((uintptr_t)a.v & 15) || ((uintptr_t)a.o & 15)) return false;
    const double lim = 2147483648.0 - 512.0;
    return ((double)a.Sk + 4 * 64) * a.ldk * 2.0 < lim && ((double)a.Sk + 4 * 64) * a.ldv * 2.0 < lim && (double)a.Sq * a.ldq * 2.0 < lim && (double)a.Sq * a.ldo * 2.0 < lim;
}

// The overflow flags of a device: 64 words, allocated and zeroed ONCE.  ltx_dit_create (head_dim 128) and ltx_warmup call
// ltx_attention_q128_prepare, so that the launch path neither allocates nor runs a synchronous memset (both would break a
// stream capture and stall other streams); a launch on a device nobody prepared still works, paying that cost once.  A failed
// memset frees the buffer and leaves nothing cached.
static int q128_flags_locked(int dev, int** out) {
    auto it = g_q128_flag.find(dev);
    if (it != g_q128_flag.end()) { *out = it->second; return LTX_OK; }
    int* f = nullptr;
    HIP_TRY(hipMalloc(&f, 64 * sizeof(int)));
    if (hipMemset(f, 0, 64 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(f); LTX_FAIL(LTX_ERR_HIP, "attn_q128: cannot zero the overflow flags"); }
    g_q128_flag[dev] = f;
    *out = f;
    return LTX_OK;
}
int ltx_attention_q128_prepare() {
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_q128_mu);
    int* f = nullptr;
    return q128_flags_locked(dev, &f);
}

// Launches the kernel; *flag_out / *ticket_out identify the overflow flag the caller gates its exact pass on.
// Flag words rotate by ticket (64 of them): a launch finds its word untouched unless 64 later launches have been ISSUED before
// its gated exact pass has RUN, and one of those overflowed into the same word - i.e. two overflows (never seen on the DiT's
// normed q / k) 64 launches apart on different streams; the exact pass of the earlier launch would then be skipped.
int ltx_launch_attention_q128(const AttnArgs& a, hipStream_t s, int** flag_out, int* ticket_out) {
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    int* flag = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_q128_mu);
        LTX_TRY(q128_flags_locked(dev, &flag));
    }
    const int ticket = g_q128_ticket.fetch_add(1) | 0x40000000;       // never 0 (the flags' initial value)
    flag += ticket & 63;                                               // 64 flag words in rotation: launches in flight on other streams keep their own
    constexpr int smem = NSLOT * TILE_BYTES;
    static std::atomic<unsigned long long> attr_devs{0};
    {
        const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
        if (!(attr_devs.load() & bit)) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_devs.fetch_or(bit);
        }
    }
    // 256-query blocks (64 queries per wave: half the K/V bytes per FLOP), the queries left over as 128-query blocks.  How many
    // of each: the split whose greedy schedule (one block per CU, big blocks first, a small block costs 0.55 of a big one)
    // finishes first - at S = 17556 x 32 heads all 68 big blocks per head (9.0 rounds for 8.57 of work whatever the split), at
    // shapes with few rounds a shorter tail (S = 4992: 15 big + 9 small per head, 2.55 rounds instead of 3.0).
    int nbig = a.Sq / 256;
    {
        static std::mutex mu; static std::map<std::tuple<int, int, int>, int> memo;
        static int cu_of_dev[64] = {0};                      // per device, queried once
        std::lock_guard<std::mutex> lock(mu);
        int n_cu = 256;
        if (dev >= 0 && dev < 64) {
            if (!cu_of_dev[dev]) { int v = 0; cu_of_dev[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
            n_cu = cu_of_dev[dev];
        }
        const auto key = std::make_tuple(a.Sq, a.B * a.heads, n_cu);
        auto it = memo.find(key);
        if (it == memo.end()) {
            const int bmax = a.Sq / 256, ht = a.B * a.heads;
            double best = 1e30; int best_b = bmax;
            for (int b = bmax; b >= 0 && b >= bmax - 2 * (n_cu / (ht > 0 ? ht : 1)) - 2; --b) {
                std::priority_queue<double, std::vector<double>, std::greater<double>> cu;
                for (int i = 0; i < n_cu; ++i) cu.push(0.0);
                const int64_t bigs = (int64_t)ht * b, smalls = (int64_t)ht * ((a.Sq - 256 * b + 127) / 128);
                double end = 0.0;
                for (int64_t i = 0; i < bigs + smalls; ++i) { const double t = cu.top() + (i < bigs ? 1.0 : 0.552); cu.pop(); cu.push(t); if (t > end) end = t; }
                if (end < best - 1e-9) { best = end; best_b = b; }
            }
            it = memo.emplace(key, best_b).first;
        }
        nbig = it->second;
    }
    { const int v = ltx_exp("attn_q128_big", -1); if (v >= 0 && v <= a.Sq / 256) nbig = v; }   // tuning aid (experiment builds; 0: 128-query blocks only)
    const int nsmall = (a.Sq - nbig * 256 + 127) / 128;
    const int blocks = a.B * a.heads * (nbig + nsmall);
    LTX_LAUNCH_TIMED(attn_q128_kernel, dim3((unsigned)blocks), dim3(256), smem, s, a, nbig, nsmall, flag, ticket);
    LTX_CHECK_LAUNCH();
    *flag_out = flag; *ticket_out = ticket;
    return LTX_OK;
}
