// DiT self-attention, head_dim 64, q prescaled by scale*log2(e), no key bias (candle-flash-attn at
// ltx_transformer.rs:699-712): one wave per SIMD, 64 queries per wave.
//
// Why this shape (docs/lab_notes.md, attention): at d = 64 the SIMD's vector issue port, not the matrix pipe, bounds the
// kernel - per 64-key tile a wave must issue one v_exp per score, one v_cvt_pk per two, the K/V fragment reads and its
// share of the LDS-DMA pieces.  Per FLOP, a wave that owns 64 queries (two 32-query MFMA column blocks) reads each K
// and V^T fragment once for both blocks and issues half the DMA pieces of a 32-query wave.  That needs the whole
// 512-register file (two S^T accumulator sets of 64, O^T 64, Q^T 32, K 32, V^T 32 fragments) -> one wave per SIMD, so
// the overlap of matrix and vector work cannot come from a SIMD partner: it is written into the instruction stream.
//
//   * workgroup = 4 waves = 256 queries of one head ("big" block) or 128 queries (QB = 1, 32 queries per wave: the
//     blocks that fill the last, partial round of the grid - see ltx_launch_attention_q64);
//   * K/V tiles of 64 keys arrive by buffer LDS-DMA into a ring of four 16-KiB slots, tile t+4 issued in iteration t,
//     counted vmcnt, one barrier per tile; the LDS images and fragment maps are those of attention.hip (K rows
//     chunk-XOR-swizzled, V row-major read through ds_read_b64_tr_b16, P^T kept in registers as the B operand);
//   * software pipeline over tiles: iteration t issues S^T(t+1) = K(t+1) . Q^T (accumulators start at -m) and
//     O^T += V^T(t) . P^T(t); the exp/convert work of tile t is spread two v_exp + one v_cvt_pk per MFMA over ALL
//     MFMAs of the iteration (groups of eight: QK block 0 | QK block 1 | PV block 0 | PV block 1), pinned with
//     sched_barrier between MFMAs; V^T(t) fragments are read under the QK groups, K(t+2) fragments and the DMA pieces
//     of tile t+4 under the PV groups;
//   * row sums come from the matrix pipe: one v_mfma_f32_16x16x32_bf16 per P operand against a constant 0/1 matrix
//     sums the bf16-rounded P of a query over its 8 keys x 2 lane halves straight into a running f32 accumulator
//     (64 v_add per tile -> 8 short MFMAs; the normaliser is the sum of exactly the P values that enter P.V);
//   * FIXED max: m of a query = its maximum over the FIRST key tile (prologue); p = exp2(s - m) is then evaluated with
//     no per-tile max, no rescale and no branch.  Softmax is invariant to m; f32 sums and bf16's 8-bit exponent make
//     the rounding independent of the magnitude of p, so the only failure is overflow (a later score exceeding the
//     first tile's maximum by ~127 in log2 units).  Overflow leaves inf/NaN in l or O^T, which is checked once per
//     block; the block then computes the exact row maxima over all keys (plain loads, no pipeline) and runs again
//     with those: a slow, always-correct path that ordinary activations never take.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "options.h"

#ifndef Q64_ROWSUM_MFMA
#define Q64_ROWSUM_MFMA 1      // 0: row sums by VALU adds
#endif
#ifndef Q64_PIN
#define Q64_PIN 1              // 0: leave the MFMA / VALU interleave to the compiler's scheduler
#endif
#ifndef Q64_ABL
#define Q64_ABL 0              // timing ablations (wrong results): 1 no DMA in the loop, 2 no exp, 3 no barrier
#endif

#ifndef Q64_ASM_LOOP
#define Q64_ASM_LOOP 1         // 0: the compiler-scheduled C++ loop for every tile (reference for the asm loop)
#endif

#ifndef Q64_STAMP
#define Q64_STAMP 0            // 1 (diagnostic builds, loop generated with stamp=1): cycle / realtime stamps around the asm loop
#endif
#if Q64_STAMP
__device__ unsigned long long q64_dbg[8 * 4096];   // per workgroup (wave 0): loop cycles, realtime ticks, iterations, QB, prologue cycles, epilogue cycles
__device__ unsigned long long q64_t_asm_end;
extern "C" int ltx_dbg_q64_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(q64_dbg), sizeof(unsigned long long) * (size_t)n);
}
#endif

// Diagnostic: workgroups that took the exact-max second pass since the last reset (fixed first-tile max overflowed).  Touched only
// on that path - the fast path never reads or writes it.
__device__ unsigned long long q64_fallback_blocks;
int ltx_q64_fallback_read(unsigned long long* out, int reset) {
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(q64_fallback_blocks), sizeof(*out)));
    if (reset) { const unsigned long long z = 0; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(q64_fallback_blocks), &z, sizeof(z))); }
    return LTX_OK;
}

namespace {

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#ifndef Q64_LOOP_INC
#define Q64_LOOP_INC "attn_q64_loop.inc"
#endif
#include Q64_LOOP_INC

constexpr int BKV = 64, KROW = 128, VROW = 128, TILE_BYTES = BKV * (KROW + VROW), NSLOT = 4;
constexpr int PW = 2, PIECES = 2 * PW;          // 1-KiB LDS-DMA pieces per wave and tile (K: 2, V: 2)
constexpr int FLAG_OFF = NSLOT * TILE_BYTES;    // one word behind the ring: "a wave saw inf/NaN"

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ int kswz8(int row, int c) { return c ^ ((row >> 1) & 7); }
__device__ __forceinline__ int vswz8(int row, int c) { return c ^ (((row >> 1) & 1) << 2); }

// LDS reads outside the compiler's memory model (it cannot tell them from the pending LDS-DMA writes of the other ring
// slots and would drain vmcnt to 0 before each); waited for with explicit lgkmcnt statements that name the registers.
template <int OFF> __device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ u32x2 lds_tr(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void wait_k(bf16x8 (&k)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7]));
}
__device__ __forceinline__ void wait_v(u32x2 (&v)[8][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[3][0]), "+v"(v[3][1]),
                 "+v"(v[4][0]), "+v"(v[4][1]), "+v"(v[5][0]), "+v"(v[5][1]), "+v"(v[6][0]), "+v"(v[6][1]), "+v"(v[7][0]), "+v"(v[7][1]));
}
__device__ __forceinline__ void pin() {
#if Q64_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// MFMAs of the compiler-scheduled parts (prologue, ragged tail iterations, exact-max pass): builtins, so that hipcc
// pads their hazards.  The bulk of the work runs in the generated asm loop (attn_q64_loop.inc).
__device__ __forceinline__ void mfma_s_first(f32x16& s, const bf16x8& k, const bf16x8& q, const f32x16& init) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q, init, 0, 0, 0);
}
__device__ __forceinline__ void mfma_s(f32x16& s, const bf16x8& k, const bf16x8& q) {
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q, s, 0, 0, 0);
}
__device__ __forceinline__ void mfma_o(f32x16& o, const bf16x8& v, const bf16x8& p) {
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, p, o, 0, 0, 0);
}
__device__ __forceinline__ void mfma_l(f32x4& l, const bf16x8& ones, const bf16x8& p) {
    l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p, l, 0, 0, 0);
}
__device__ __forceinline__ void settle(f32x16&) {}
__device__ __forceinline__ void mfma_drain() {}

// 16-byte row stores of O (attention.hip store_o_wide): lanes l and l^32 exchange one packed column group per pair
__device__ __forceinline__ void store_o64(const f32x16 (&acc_o)[2], float inv, bf16_t* O, int h, bool wide) {
    if (wide) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                union { bf16x4 v; unsigned u[2]; } a, b;
                a.v = (bf16x4){(bf16_t)(acc_o[d][8 * k + 0] * inv), (bf16_t)(acc_o[d][8 * k + 1] * inv), (bf16_t)(acc_o[d][8 * k + 2] * inv), (bf16_t)(acc_o[d][8 * k + 3] * inv)};
                b.v = (bf16x4){(bf16_t)(acc_o[d][8 * k + 4] * inv), (bf16_t)(acc_o[d][8 * k + 5] * inv), (bf16_t)(acc_o[d][8 * k + 6] * inv), (bf16_t)(acc_o[d][8 * k + 7] * inv)};
                const auto s0 = __builtin_amdgcn_permlane32_swap(a.u[0], b.u[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a.u[1], b.u[1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                *reinterpret_cast<u32x4*>(O + d * 32 + 16 * k + 8 * h) = o;
            }
    } else {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;
                bf16x4 o4 = {(bf16_t)(acc_o[d][4 * g4 + 0] * inv), (bf16_t)(acc_o[d][4 * g4 + 1] * inv),
                             (bf16_t)(acc_o[d][4 * g4 + 2] * inv), (bf16_t)(acc_o[d][4 * g4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(O + dd) = o4;
            }
    }
}

// Slab of one key-range part of a split 256-query block (persistent kernel below): per wave 16 fragment groups of O^T
// (a[4g:4g+3] of the generated loop = acc_o[g >> 3][(g >> 2) & 1][4 * (g & 3) ..]) and one group (own l of q0, of q1, -m of
// q0, of q1), each group 64 lanes x 16 B.  Written with write-through (sc1) 16-byte stores and read back by the merging
// workgroup with sc1 loads behind the arrival ticket (MI355X guide, hand-off table row 1: no fences).
constexpr int SLAB_GROUPS = 17, SLAB_WAVE_F = SLAB_GROUPS * 256, SLAB_F = 4 * SLAB_WAVE_F;     // floats
__device__ __forceinline__ void st16_sc1(float* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld16_sc1(const float* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); return v; }

// One workgroup: queries [q_first, q_first + 128*QB) of (batch b, head) against the keys [k_row0, k_row0 + Sk).
// QB = 32-query column blocks per wave.  slab == nullptr: the keys are all keys, the result is normalised and stored;
// slab != nullptr (QB = 2): a key-range part - un-normalised O^T, l and -m go to the slab (merge: attn_q64_merge).
template <int QB>
__device__ __forceinline__ void attn_q64_block(const AttnArgs& a, unsigned char* smem, int b, int head, int q_first, int k_row0, int Sk, float* slab) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = q_first + wave * 32 * QB;
#if Q64_STAMP
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + head * 64;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + ((int64_t)b * a.Sk + k_row0) * a.ldk + head * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + ((int64_t)b * a.Sk + k_row0) * a.ldv + head * 64;
    const int nt = (Sk + BKV - 1) / BKV;
    const bool ragged = (Sk % BKV) != 0;

    // Q^T operand fragments (B operand: k = d, col = query); rows past Sq repeat the last one (never stored)
    bf16x8 qf[QB][4];
    auto load_q = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            int qr = q0 + 32 * qb + r; if (qr > a.Sq - 1) qr = a.Sq - 1;
            const bf16_t* qp = Q + (int64_t)qr * a.ldq + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
        }
    };

    // ---- LDS-DMA geometry (attention.hip): pieces of 8 rows x 128 B, wave w issues pieces 2w, 2w+1 of K and of V;
    // the tile's bank swizzles are applied to the SOURCE chunk; rows past Sk are out of the buffer's range -> zeros
    __amdgpu_buffer_rsrc_t rk_rsrc, rv_rsrc;
    uint32_t k_voff[PW], v_voff[PW];
    {
        const uint32_t k_bytes = (uint32_t)(Sk - 1) * (uint32_t)a.ldk * 2u + 128u;
        const uint32_t v_bytes = (uint32_t)(Sk - 1) * (uint32_t)a.ldv * 2u + 128u;
        rk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, (int)k_bytes, 0x00020000);
        rv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V), 0, (int)v_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int row = (wave * PW + j) * 8 + (lane >> 3), pc = lane & 7;
            k_voff[j] = (uint32_t)row * (uint32_t)a.ldk * 2u + (uint32_t)kswz8(row, pc) * 16u;
            v_voff[j] = (uint32_t)row * (uint32_t)a.ldv * 2u + (uint32_t)vswz8(row, pc) * 16u;
        }
    }
    // the same two descriptors as plain words in SGPRs, for the generated loop
    u32x4 rk_words, rv_words;
    {
        const uint64_t kp = (uint64_t)(uintptr_t)K, vp = (uint64_t)(uintptr_t)V;
        const uint32_t k_bytes = (uint32_t)(Sk - 1) * (uint32_t)a.ldk * 2u + 128u, v_bytes = (uint32_t)(Sk - 1) * (uint32_t)a.ldv * 2u + 128u;
        rk_words = (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kp >> 32)) & 0xffffu,
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)k_bytes), 0x00020000u};
        rv_words = (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(vp >> 32)) & 0xffffu,
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)v_bytes), 0x00020000u};
    }
    auto dma_piece = [&](int t, int slot, int which) {       // which: 0,1 = K pieces, 2,3 = V pieces
        unsigned char* Ks = smem + slot * TILE_BYTES;
        const int j = which & 1;
        if (which < 2) {
            const uint32_t ks = (uint32_t)t * BKV * (uint32_t)a.ldk * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk_rsrc, (__attribute__((address_space(3))) void*)(Ks + (wave * PW + j) * 1024), 16, (int)k_voff[j], (int)ks, 0, 0);
        } else {
            const uint32_t vs = (uint32_t)t * BKV * (uint32_t)a.ldv * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv_rsrc, (__attribute__((address_space(3))) void*)(Ks + BKV * KROW + (wave * PW + j) * 1024), 16, (int)v_voff[j], (int)vs, 0, 0);
        }
    };
    auto dma_tile = [&](int t, int slot) { for (int w = 0; w < 4; ++w) dma_piece(t, slot, w); };

    // lane parts of the LDS read addresses (row offsets inside a tile and static ring slots are immediates)
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t k_base[4], tr_base[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_base[ks] = smem_base + r * KROW + kswz8(r, 2 * ks + h) * 16;   // rows r, r+32 swizzle alike
    {
        const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int row = 4 * h + trq, cv = d * 4 + trdh * 2 + (trp >> 1);
            tr_base[d] = smem_base + row * VROW + vswz8(row, cv) * 16 + (trp & 1) * 8;
        }
    }
    // K fragment i = kb*4 + ks of the tile in ring slot SLOT (>= 0: immediate) or slot_dyn
    auto read_k = [&](auto slot_tag, int slot_dyn, auto i_tag, bf16x8 (&kf)[8]) {
        constexpr int SLOT = decltype(slot_tag)::value, i = decltype(i_tag)::value, kb = i >> 2, ks = i & 3;
        constexpr int imm = (SLOT >= 0 ? SLOT : 0) * TILE_BYTES + kb * 32 * KROW;
        kf[i] = lds_b128<imm>(k_base[ks] + (SLOT >= 0 ? 0u : (uint32_t)slot_dyn * TILE_BYTES));
    };
    // V^T operand n = d*4 + j (keys (j>>1)*32 + (j&1)*16 .. +15 of d block d): half hf of its two transpose reads
    auto read_v = [&](auto slot_tag, int slot_dyn, auto n_tag, auto hf_tag, u32x2 (&vf)[8][2]) {
        constexpr int SLOT = decltype(slot_tag)::value, n = decltype(n_tag)::value, hf = decltype(hf_tag)::value, d = n >> 2, j = n & 3;
        constexpr int rowc = (j >> 1) * 32 + (j & 1) * 16 + 8 * hf;
        constexpr int imm = (SLOT >= 0 ? SLOT : 0) * TILE_BYTES + BKV * KROW + rowc * VROW;
        vf[n][hf] = lds_tr<imm>(tr_base[d] + (SLOT >= 0 ? 0u : (uint32_t)slot_dyn * TILE_BYTES));
    };

    // constant A operand of the row-sum MFMA (16x16x32): A[i][k] = 1 where the parities of row i and of k's group of 8
    // agree -> D[even rows][c] = sum over lanes c, c+32 (query c), D[odd rows][c] = lanes 16+c, 48+c (query 16+c);
    // a lane finds its own query's sum in register (lane >> 4) & 1 of the result
    bf16x8 ones;
    {
        const bf16_t v1 = (bf16_t)(((lane & 1) == ((lane >> 4) & 1)) ? 1.0f : 0.0f);
#pragma unroll
        for (int i = 0; i < 8; ++i) ones[i] = v1;
    }

    f32x16 S[2][QB][2];            // S^T accumulator sets: [set][query block][key block]
    constexpr int NPS = QB == 1 ? 2 : 1;   // QB = 1 exponentiates tile t+1's first slice while P.V reads tile t's -> two sets
    bf16x8 P[NPS][QB][2][2];       // P^T operands: [set][query block][key block][k-step]
    f32x16 acc_o[QB][2];
    f32x16 minit[QB];              // -m in all 16 registers (a lane owns one query column)
    f32x4 lacc[QB];
    float lsum[QB];                // Q64_ROWSUM_MFMA == 0
    bf16x8 kf[8];
    u32x2 vf[8][2];
    float m_fix[QB];

    // exp2 + convert of one key block of one query block: 16 v_exp + 8 v_cvt_pk, cut into 8 slices (one per MFMA gap)
    auto exp_slice = [&](f32x16& s, bf16x8 (&p)[2], float& ls, auto i_tag) {
        constexpr int i = decltype(i_tag)::value;          // slice i: registers 2i, 2i+1
#if Q64_ABL == 2
        const float p0 = s[2 * i], p1 = s[2 * i + 1];
#else
        const float p0 = __builtin_amdgcn_exp2f(s[2 * i]), p1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
#endif
#if !Q64_ROWSUM_MFMA
        ls += p0 + p1;
#endif
        p[i >> 2][(2 * i) & 7] = (bf16_t)p0;
        p[i >> 2][((2 * i) & 7) + 1] = (bf16_t)p1;
    };
    auto mask_tail = [&](f32x16 (&s)[QB][2], int t) {      // keys past Sk of tile t -> -inf
        const int kv0 = t * BKV;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= Sk) s[qb][kb][i] = -INFINITY;
                }
    };
    auto qk_mfma = [&](f32x16& s, const bf16x8& k, const bf16x8& q, const f32x16& init, auto first_tag) {
        if constexpr (decltype(first_tag)::value) mfma_s_first(s, k, q, init); else mfma_s(s, k, q);
    };

    // ---- iteration t.  cur = set holding S^T(t) - m (its block-0 / key-block-0 slice already exponentiated),
    // nxt = set receiving S^T(t+1).  FLAGS: bit 0 = tile t+1 exists, bit 1 = tile t+2 exists, bit 2 = tile t+1 is the
    // ragged last tile.  Static bodies (SLOT >= 0) have all of bits 0,1 set.
    auto body = [&](int t, auto slot_tag, auto cur_tag, int flags) {
        constexpr int SLOT = decltype(slot_tag)::value, CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        constexpr int PC = QB == 1 ? CUR : 0, PN = QB == 1 ? NXT : 0;
        constexpr bool STATIC = SLOT >= 0;
        const bool next1 = STATIC || (flags & 1), next2 = STATIC || (flags & 2);
        const int slot = STATIC ? SLOT : (t & 3);
        using vslot = std::integral_constant<int, SLOT>;
        using kslot = std::integral_constant<int, STATIC ? ((SLOT + 2) & 3) : -1>;
        const int kslot_dyn = (t + 2) & 3;
        // --- QK groups: S^T(t+1), exps of tile t, V^T(t) reads
        sfor<0, QB>([&](auto qb_tag) {
            constexpr int qb = decltype(qb_tag)::value;
            sfor<0, 8>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, kb = i >> 2, ks = i & 3;
                if (next1) qk_mfma(S[NXT][qb][kb], kf[i], qf[qb][ks], minit[qb], std::bool_constant<ks == 0>{});
                // exps: QB = 2: group 0 -> (q0, k1), group 1 -> (q1, k0);  QB = 1: group 0 -> (q0, k1)
                if constexpr (QB == 2 && qb == 1) exp_slice(S[CUR][1][0], P[PC][1][0], lsum[1], i_tag);
                else exp_slice(S[CUR][0][1], P[PC][0][1], lsum[0], i_tag);
                // V^T(t): 16 transpose reads over the QK gaps
                if constexpr (QB == 2) read_v(vslot{}, slot, std::integral_constant<int, ((qb * 8 + i) / 2)>{}, std::integral_constant<int, ((qb * 8 + i) & 1)>{}, vf);
                else { read_v(vslot{}, slot, i_tag, std::integral_constant<int, 0>{}, vf); read_v(vslot{}, slot, i_tag, std::integral_constant<int, 1>{}, vf); }
                pin();
            });
        });
        if constexpr (!STATIC) { if ((flags & 4)) mask_tail(S[NXT], t + 1); }
        // --- tile t+2 landed for every wave (tile t+3's pieces may stay in flight); V^T(t) fragments are in
#if Q64_ABL != 1
        if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        wait_v(vf);
#if Q64_ABL != 3
        __builtin_amdgcn_s_barrier();
#endif
        pin();
        // --- PV groups: O^T += V^T(t) . P^T(t), row sums, remaining exps, K(t+2) reads, DMA of tile t+4
        const bool do_dma = Q64_ABL != 1 && (t + 4 < nt);
        sfor<0, QB>([&](auto qb_tag) {
            constexpr int qb = decltype(qb_tag)::value;
            sfor<0, 8>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, d = i >> 2, j = i & 3;
                union { u32x2 u[2]; bf16x8 v; } cvt;
                cvt.u[0] = vf[i][0]; cvt.u[1] = vf[i][1];
                mfma_o(acc_o[qb][d], cvt.v, P[PC][qb][j >> 1][j & 1]);
                // exps: QB = 2: group 0 -> (q1, k1) of tile t, group 1 -> (q0, k0) of tile t+1;  QB = 1: (q0, k0) of tile t+1
                if constexpr (QB == 2 && qb == 0) exp_slice(S[CUR][1][1], P[PC][1][1], lsum[1], i_tag);
                else { if (next1) exp_slice(S[NXT][0][0], P[PN][0][0], lsum[0], i_tag); }
                // K(t+2) fragments: one per gap of the first PV group
                if constexpr (qb == 0) { if (next2) read_k(kslot{}, kslot_dyn, i_tag, kf); }
                // DMA pieces of tile t+4 (slot of tile t, free since the barrier): last PV group, every other gap
                if constexpr (qb == QB - 1 && (i & 1) == 0) { if (do_dma) dma_piece(t + 4, slot, i >> 1); }
#if Q64_ROWSUM_MFMA
                if constexpr (i >= 4) mfma_l(lacc[qb], ones, P[PC][qb][(i - 4) >> 1][(i - 4) & 1]);
#endif
                pin();
            });
        });
        if (next2) wait_k(kf);
        pin();
    };

    // ---- one pass over all keys with the row maxima m_fix[] ------------------------------------------------------
    auto run_pass = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { minit[qb][i] = -m_fix[qb]; acc_o[qb][0][i] = 0.f; acc_o[qb][1][i] = 0.f; }
            lacc[qb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lsum[qb] = 0.f;
        }
        // S^T(0) - m with K(0), then K(1) fragments, then the first exp slice of tile 0
        sfor<0, 8>([&](auto i_tag) { read_k(std::integral_constant<int, 0>{}, 0, i_tag, kf); });
        wait_k(kf);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) settle(minit[qb]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
            sfor<0, 8>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                qk_mfma(S[0][qb][i >> 2], kf[i], qf[qb][i & 3], minit[qb], std::bool_constant<(i & 3) == 0>{});
            });
        mfma_drain();
        if (nt == 1 && ragged) mask_tail(S[0], 0);
        if (nt > 1) { sfor<0, 8>([&](auto i_tag) { read_k(std::integral_constant<int, 1>{}, 0, i_tag, kf); }); wait_k(kf); }
        sfor<0, 8>([&](auto i_tag) { exp_slice(S[0][0][0], P[0][0][0], lsum[0], i_tag); });
        pin();
        int t = 0;
#if Q64_ASM_LOOP
        // the generated loop runs every iteration whose next tile needs no masking: all of them, or all but the last two
        if (const int N = ragged ? nt - 2 : nt; N > 0) {
            union B8 { bf16x8 v; uint32_t u[4]; };
            u32x32 pk, kk; u32x4 ones_u, kbase_u, dma_u; u32x2 tr_u;
            { B8 c; c.v = ones; ones_u = (u32x4){c.u[0], c.u[1], c.u[2], c.u[3]}; }
            kbase_u = (u32x4){k_base[0], k_base[1], k_base[2], k_base[3]};
            tr_u = (u32x2){tr_base[0], tr_base[1]};
            dma_u = (u32x4){k_voff[0], k_voff[1], v_voff[0], v_voff[1]};
#pragma unroll
            for (int i = 0; i < 8; ++i) { B8 c; c.v = kf[i]; kk[4 * i] = c.u[0]; kk[4 * i + 1] = c.u[1]; kk[4 * i + 2] = c.u[2]; kk[4 * i + 3] = c.u[3]; }
            int cnt = N;
            const uint32_t kstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldk * 2u));
            const uint32_t vstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldv * 2u));
            uint32_t koff = 4u * kstep, voff = 4u * vstep;
            const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * (PW * 1024)));
#if Q64_STAMP
            unsigned long long st0 = 0, sr0 = 0, st1 = 0, sr1 = 0;
#define Q64_ST , st0, sr0, st1, sr1
#else
#define Q64_ST
#endif
            if constexpr (QB == 2) {
                f32x32 sx0, sx1, sy0, sy1, mi, o0, o1; f32x8 la; u32x32 qq;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sx0[i] = S[0][0][0][i]; sx0[16 + i] = S[0][0][1][i]; sx1[i] = S[0][1][0][i]; sx1[16 + i] = S[0][1][1][i];
                    sy0[i] = 0.f; sy0[16 + i] = 0.f; sy1[i] = 0.f; sy1[16 + i] = 0.f;
                    mi[i] = minit[0][i]; mi[16 + i] = minit[1][i];
                    o0[i] = acc_o[0][0][i]; o0[16 + i] = acc_o[0][1][i]; o1[i] = acc_o[1][0][i]; o1[16 + i] = acc_o[1][1][i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { la[i] = lacc[0][i]; la[4 + i] = lacc[1][i]; }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { B8 c; c.v = qf[qb][ks]; for (int w = 0; w < 4; ++w) qq[(qb * 4 + ks) * 4 + w] = c.u[w]; }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) { B8 c; c.v = P[0][qb][kb][s2]; for (int w = 0; w < 4; ++w) pk[((qb * 2 + kb) * 2 + s2) * 4 + w] = c.u[w]; }
                q64_loop_qb2(sx0, sx1, sy0, sy1, pk, mi, la, o0, o1, qq, kk, ones_u, kbase_u, tr_u, dma_u, rk_words, rv_words, cnt, koff, kstep, voff, vstep, ldsw Q64_ST);
                const bool odd = (N & 1) != 0;                 // tile N (if any) sits in set N & 1: the tail below wants it in set 0
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[0][0][0][i] = odd ? sy0[i] : sx0[i]; S[0][0][1][i] = odd ? sy0[16 + i] : sx0[16 + i];
                    S[0][1][0][i] = odd ? sy1[i] : sx1[i]; S[0][1][1][i] = odd ? sy1[16 + i] : sx1[16 + i];
                    acc_o[0][0][i] = o0[i]; acc_o[0][1][i] = o0[16 + i]; acc_o[1][0][i] = o1[i]; acc_o[1][1][i] = o1[16 + i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { lacc[0][i] = la[i]; lacc[1][i] = la[4 + i]; }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) { B8 c; for (int w = 0; w < 4; ++w) c.u[w] = pk[((qb * 2 + kb) * 2 + s2) * 4 + w]; P[0][qb][kb][s2] = c.v; }
            } else {
                f32x32 sx0, sy0, o0; f32x16 mi; f32x4 la; u32x16 qq;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sx0[i] = S[0][0][0][i]; sx0[16 + i] = S[0][0][1][i]; sy0[i] = 0.f; sy0[16 + i] = 0.f;
                    mi[i] = minit[0][i]; o0[i] = acc_o[0][0][i]; o0[16 + i] = acc_o[0][1][i];
                }
                la = lacc[0];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { B8 c; c.v = qf[0][ks]; for (int w = 0; w < 4; ++w) qq[ks * 4 + w] = c.u[w]; }
#pragma unroll
                for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) { B8 c; c.v = P[ps][0][kb][s2]; for (int w = 0; w < 4; ++w) pk[ps * 16 + (kb * 2 + s2) * 4 + w] = c.u[w]; }
                q64_loop_qb1(sx0, sy0, pk, mi, la, o0, qq, kk, ones_u, kbase_u, tr_u, dma_u, rk_words, rv_words, cnt, koff, kstep, voff, vstep, ldsw Q64_ST);
                const bool odd = (N & 1) != 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[0][0][0][i] = odd ? sy0[i] : sx0[i]; S[0][0][1][i] = odd ? sy0[16 + i] : sx0[16 + i];
                    acc_o[0][0][i] = o0[i]; acc_o[0][1][i] = o0[16 + i];
                }
                lacc[0] = la;
                // P set of tile N -> set 0
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) { B8 c; for (int w = 0; w < 4; ++w) c.u[w] = odd ? pk[16 + (kb * 2 + s2) * 4 + w] : pk[(kb * 2 + s2) * 4 + w]; P[0][0][kb][s2] = c.v; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { B8 c; c.u[0] = kk[4 * i]; c.u[1] = kk[4 * i + 1]; c.u[2] = kk[4 * i + 2]; c.u[3] = kk[4 * i + 3]; kf[i] = c.v; }
#if Q64_STAMP
            if (tid == 0 && blockIdx.x < 4096) {
                q64_dbg[8 * blockIdx.x] = st1 - st0; q64_dbg[8 * blockIdx.x + 1] = sr1 - sr0;
                q64_dbg[8 * blockIdx.x + 2] = (unsigned long long)N; q64_dbg[8 * blockIdx.x + 3] = QB;
                q64_dbg[8 * blockIdx.x + 4] = st0 - t_entry; q64_dbg[8 * blockIdx.x + 5] = st1;
            }
#endif
            t = N;
        }
        // remaining (ragged) iterations: tile t sits in set 0, so the parity of the compiler-scheduled bodies restarts
        using dyn0 = std::integral_constant<int, -1>;
        auto flags0 = [&](int tt) { return (tt + 1 < nt ? 1 : 0) | (tt + 2 < nt ? 2 : 0) | ((ragged && tt + 1 == nt - 1) ? 4 : 0); };
        for (int u = 0; t < nt; t += 2, ++u) {
            body(t, dyn0{}, std::integral_constant<int, 0>{}, flags0(t));
            if (t + 1 < nt) body(t + 1, dyn0{}, std::integral_constant<int, 1>{}, flags0(t + 1));
        }
        mfma_drain();
        return;
#endif
        for (; t <= nt - 6; t += 4) {
            body(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 3);
            body(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, 3);
            body(t + 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, 3);
            body(t + 3, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, 3);
        }
        using dyn = std::integral_constant<int, -1>;
        auto flags_of = [&](int tt) { return (tt + 1 < nt ? 1 : 0) | (tt + 2 < nt ? 2 : 0) | ((ragged && tt + 1 == nt - 1) ? 4 : 0); };
        for (; t < nt; t += 2) {
            body(t, dyn{}, std::integral_constant<int, 0>{}, flags_of(t));
            if (t + 1 < nt) body(t + 1, dyn{}, std::integral_constant<int, 1>{}, flags_of(t + 1));
        }
        mfma_drain();
    };

    // ---- prologue: first four tiles in flight, row maxima over tile 0 -------------------------------------------
    auto issue_prologue = [&]() {
        dma_tile(0, 0);
        if (nt > 1) dma_tile(1, 1);
        if (nt > 2) dma_tile(2, 2);
        if (nt > 3) dma_tile(3, 3);
        // tiles 0 and 1 landed; tiles 2, 3 may stay in flight
        if (nt > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        else if (nt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto lane_pair_max = [&](float mt) {
        unsigned mu = __float_as_uint(mt);
        auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    auto own_l = [&](int qb) {
#if Q64_ROWSUM_MFMA
        return (lane & 16) ? lacc[qb][1] : lacc[qb][0];
#else
        return lsum[qb] + __shfl_xor(lsum[qb], 32);
#endif
    };
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem + FLAG_OFF);
    auto block_any = [&](bool bad) {                       // block-uniform OR through one LDS word
        if (tid == 0) *flag = 0u;
        __syncthreads();
        if (__any(bad) && lane == 0) *flag = 1u;
        __syncthreads();
        return *flag != 0u;
    };
#ifndef Q64_NO_FALLBACK
#define Q64_NO_FALLBACK 0   // 1: timing ablations whose garbage results must not start the second pass
#endif
#ifndef Q64_ASM_FULL
#define Q64_ASM_FULL 1      // 0: compiler-scheduled prologue / epilogue around the generated loop for every shape
#endif

    bool need_exact = false;
#if Q64_ASM_LOOP && Q64_ASM_FULL && Q64_ROWSUM_MFMA
    // ---- fast path (key count a multiple of 64): prologue, loop and epilogue are one generated asm statement; only
    // the addresses come from here.  It stores its result; the overflow check below decides whether that stands.
    if (!ragged) {
        union B8 { bf16x8 v; uint32_t u[4]; };
        u32x4 ones_u; { B8 c; c.v = ones; ones_u = (u32x4){c.u[0], c.u[1], c.u[2], c.u[3]}; }
        const u32x4 kbase_u = {k_base[0], k_base[1], k_base[2], k_base[3]};
        const u32x2 tr_u = {tr_base[0], tr_base[1]};
        const u32x4 dma_u = {k_voff[0], k_voff[1], v_voff[0], v_voff[1]};
        u32x2 qoff, ooff;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qr = q0 + 32 * (qb < QB ? qb : 0) + r;
            const int qc = qr > a.Sq - 1 ? a.Sq - 1 : qr;
            qoff[qb] = (uint32_t)qc * (uint32_t)a.ldq * 2u + 16u * h;
            ooff[qb] = qr < a.Sq ? (uint32_t)qr * (uint32_t)a.ldo * 2u + 16u * h : 0x80000000u;     // rows past Sq: out of range, dropped
        }
        const uint64_t qp = (uint64_t)(uintptr_t)Q;
        uint64_t op = (uint64_t)(uintptr_t)(reinterpret_cast<bf16_t*>(a.o) + (int64_t)b * a.Sq * a.ldo + head * 64);
        uint32_t o_bytes = (uint32_t)(a.Sq - 1) * (uint32_t)a.ldo * 2u + 128u;
        if (slab) {                                         // part: this wave's 17 KiB of the slab, lane-linear 16-byte groups
            op = (uint64_t)(uintptr_t)(slab + wave * SLAB_WAVE_F); o_bytes = SLAB_WAVE_F * 4u;
            ooff[0] = (uint32_t)lane * 16u; ooff[1] = 0u;
        }
        const u32x4 rq_words = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)qp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(qp >> 32)) & 0xffffu,
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a.Sq - 1) * (uint32_t)a.ldq * 2u + 128u)), 0x00020000u};
        const u32x4 ro_words = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)op), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(op >> 32)) & 0xffffu,
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)o_bytes), 0x00020000u};
        const uint32_t kstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldk * 2u));
        const uint32_t vstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldv * 2u));
        const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * (PW * 1024)));
        const int nt_u = __builtin_amdgcn_readfirstlane(nt);
        const uint32_t sel = (lane & 16) ? 1u : 0u;
#if Q64_STAMP
        unsigned long long sp0 = 0, sq0 = 0, st0 = 0, sr0 = 0, st1 = 0, sr1 = 0, sp1 = 0, sq1 = 0, sa1 = 0, sa2 = 0, sa3 = 0, sa4 = 0;
#define Q64_STF , sp0, sq0, st0, sr0, st1, sr1, sp1, sq1, sa1, sa2, sa3, sa4
#else
#define Q64_STF
#endif
        bool bad = false;
        if constexpr (QB == 2) {
            f32x8 la;
#if Q64_STAMP
            q64_full_qb2(la, ones_u, kbase_u, tr_u, dma_u, qoff, ooff, sel, rk_words, rv_words, rq_words, ro_words, nt_u, kstep, vstep, ldsw Q64_STF);
#else
            if (slab) q64_part_qb2(la, ones_u, kbase_u, tr_u, dma_u, qoff, ooff, sel, rk_words, rv_words, rq_words, ro_words, nt_u, kstep, vstep, ldsw);
            else q64_full_qb2(la, ones_u, kbase_u, tr_u, dma_u, qoff, ooff, sel, rk_words, rv_words, rq_words, ro_words, nt_u, kstep, vstep, ldsw);
#endif
            const float l0 = (lane & 16) ? la[1] : la[0], l1 = (lane & 16) ? la[5] : la[4];
            bad = !(l0 < 0x1p100f) || !(l1 < 0x1p100f);
        } else {
            f32x4 la;
            q64_full_qb1(la, ones_u, kbase_u, tr_u, dma_u, qoff, ooff, sel, rk_words, rv_words, rq_words, ro_words, nt_u, kstep, vstep, ldsw Q64_STF);
            const float l0 = (lane & 16) ? la[1] : la[0];
            bad = !(l0 < 0x1p100f);
        }
#if Q64_STAMP
        if (tid == 0 && blockIdx.x < 4096) {
            q64_dbg[8 * blockIdx.x] = st1 - st0; q64_dbg[8 * blockIdx.x + 1] = sr1 - sr0;
            q64_dbg[8 * blockIdx.x + 2] = (unsigned long long)nt; q64_dbg[8 * blockIdx.x + 3] = QB;
            q64_dbg[8 * blockIdx.x + 4] = st0 - t_entry; q64_dbg[8 * blockIdx.x + 5] = sp1 - st1;
            q64_dbg[8 * blockIdx.x + 6] = ((sp0 - t_entry) << 48) | ((sa1 - sp0) << 32) | ((sa2 - sa1) << 16) | (sa3 - sa2);
            q64_dbg[8 * blockIdx.x + 7] = ((sa4 - sa3) << 32) | (st0 - sa4);
        }
#endif
        // l beyond 2^100 (or NaN): some p overflowed, or came within 2^27 of it - O^T is then not to be trusted either
        if (Q64_NO_FALLBACK || !block_any(bad)) return;
        need_exact = true;
        if (tid == 0) atomicAdd(&q64_fallback_blocks, 1ull);
    }
#endif

    // ---- generic path: ragged key counts, and the exact-max pass after an overflow.  Pass 0 takes the row maxima
    // from the first tile; pass 1 (only after an overflow) computes them over all keys first.
    load_q();
    for (int pass = need_exact ? 1 : 0; pass < 2; ++pass) {
        f32x16 zero16;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
        if (pass == 1) {
            // exact row maxima over all keys: K fragments by plain global loads (32 rows x 32 B each; slow path)
            float mx[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) mx[qb] = -INFINITY;
            for (int kb = 0; kb < 2 * nt; ++kb) {
                int key = kb * 32 + r; if (key > Sk - 1) key = Sk - 1;     // rows past Sk repeat the last key: harmless for a maximum
                bf16x8 kk[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kk[ks] = *reinterpret_cast<const bf16x8*>(K + (int64_t)key * a.ldk + 16 * ks + 8 * h);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x16 sc = zero16;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kk[ks], qf[qb][ks], sc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 16; ++i) mx[qb] = fmaxf(mx[qb], sc[i]);
                }
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) m_fix[qb] = lane_pair_max(mx[qb]);
            __syncthreads();                               // every wave is done with the ring before it is refilled
        }
        issue_prologue();
        if (pass == 0) {
            sfor<0, 8>([&](auto i_tag) { read_k(std::integral_constant<int, 0>{}, 0, i_tag, kf); });
            wait_k(kf);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                sfor<0, 8>([&](auto i_tag) {
                    constexpr int i = decltype(i_tag)::value;
                    qk_mfma(S[0][qb][i >> 2], kf[i], qf[qb][i & 3], zero16, std::bool_constant<(i & 3) == 0>{});
                });
            }
            if (nt == 1 && ragged) mask_tail(S[0], 0);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mt = fmaxf(S[0][qb][0][0], S[0][qb][1][0]);
#pragma unroll
                for (int i = 1; i < 16; ++i) mt = fmaxf(fmaxf(mt, S[0][qb][0][i]), S[0][qb][1][i]);
                m_fix[qb] = lane_pair_max(mt);
            }
        }
        run_pass();
        if (pass == 0) {
            float chk = 0.f;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                chk += own_l(qb) * 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) chk += acc_o[qb][0][i] * 0.f + acc_o[qb][1][i] * 0.f;
            }
            if (Q64_NO_FALLBACK || !block_any(!(chk == 0.f))) break;      // inf * 0 and NaN * 0 are NaN
            if (tid == 0) atomicAdd(&q64_fallback_blocks, 1ull);
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    if (slab) {                                             // key-range part: the generated part epilogue's slab image
        float* sw = slab + wave * SLAB_WAVE_F + lane * 4;
#pragma unroll
        for (int g = 0; g < 8 * QB; ++g) {
            const f32x16& o = acc_o[g >> 3][(g >> 2) & 1];
            st16_sc1(sw + g * 256, (f32x4){o[4 * (g & 3)], o[4 * (g & 3) + 1], o[4 * (g & 3) + 2], o[4 * (g & 3) + 3]});
        }
        st16_sc1(sw + 16 * 256, (f32x4){own_l(0), QB == 2 ? own_l(QB - 1) : 0.f, -m_fix[0], -m_fix[QB - 1]});
        return;
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float inv = 1.0f / own_l(qb);
        const int qr = q0 + 32 * qb + r;
        if (qr < a.Sq) {
            bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + ((int64_t)b * a.Sq + qr) * a.ldo + head * 64;
            store_o64(acc_o[qb], inv, O, h, a.wide_o != 0);
        }
    }
#if Q64_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x < 4096) q64_dbg[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime() - q64_dbg[8 * blockIdx.x + 5];
#endif
}

// Grid: per batch, first heads*nbig big blocks (256 queries: [i*256, +256)), then heads*nsmall small blocks
// (128 queries: [nbig*256 + i*128, +128)).  Inside each class blocks L, L+8 share an XCD, which is given whole heads.
__global__ __launch_bounds__(256, 1) void attn_q64_kernel(const AttnArgs a, int nbig, int nsmall) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * TILE_BYTES + 16];
    const int per_b = a.heads * (nbig + nsmall);
    int L = blockIdx.x;
    const int b = L / per_b; L -= b * per_b;
    const bool big = L < a.heads * nbig;
    if (!big) L -= a.heads * nbig;
    const int n = big ? nbig : nsmall;
    int head, qb;
    if (a.xcd_heads) { const int xcd = L & 7, j = L >> 3; head = xcd + 8 * (j / n); qb = j % n; }
    else { head = L / n; qb = L - head * n; }
    if (big) attn_q64_block<2>(a, smem, b, head, qb * 256, 0, a.Sk, nullptr);
    else attn_q64_block<1>(a, smem, b, head, nbig * 256 + qb * 128, 0, a.Sk, nullptr);
}

// ---- persistent form: one workgroup per CU walks a static list of items ---------------------------------------------
// The one-block-per-(head, 256 queries) grid above runs 2.44 rounds of work in 2 + 0.70 rounds at S = 4992 (the last
// round as 128-query blocks at 70 % of a big block's time each).  Here the host cuts the work evenly: every CU gets
// whole blocks plus ONE key range ("part") of a block that is shared with one or two other CUs; the parts of a block
// leave un-normalised (O^T, l, m) in slabs and the workgroup that draws the last arrival ticket merges them
// (O = sum_p 2^(m_p - M) O_p in part order: the result does not depend on who merges).  No workgroup ever waits for another.
struct Q64Item { int b, head, q_first, kind, k_row0, Sk, slab0, part, nparts, cnt; };   // kind: QB (2 = 256 queries, 1 = 128)

// the merging workgroup: all parts of the block are in their slabs (ticket), this wave's 64 queries
__device__ __forceinline__ void attn_q64_merge(const AttnArgs& a, const Q64Item& it, const float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const float* base = slabs + (int64_t)it.slab0 * SLAB_F + wave * SLAB_WAVE_F + lane * 4;
    constexpr int MAXP = 8;
    float negm[2][MAXP], lp[2][MAXP];
    float negM[2] = {INFINITY, INFINITY};
    for (int p = 0; p < it.nparts; ++p) {
        f32x4 t = ld16_sc1(base + (int64_t)p * SLAB_F + 16 * 256);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t));
        lp[0][p] = t[0]; lp[1][p] = t[1]; negm[0][p] = t[2]; negm[1][p] = t[3];
        negM[0] = fminf(negM[0], t[2]); negM[1] = fminf(negM[1], t[3]);
    }
    f32x16 acc[2][2];
    float L[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[qb][d][i] = 0.f;
    for (int p = 0; p < it.nparts; ++p) {
        const float f0 = __builtin_amdgcn_exp2f(negM[0] - negm[0][p]), f1 = __builtin_amdgcn_exp2f(negM[1] - negm[1][p]);   // 2^(m_p - M) <= 1
        L[0] += f0 * lp[0][p]; L[1] += f1 * lp[1][p];
        const float* sp = base + (int64_t)p * SLAB_F;
        f32x4 g[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = ld16_sc1(sp + i * 256);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),
                     "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]), "+v"(g[13]), "+v"(g[14]), "+v"(g[15]));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float f = i < 8 ? f0 : f1;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i >> 3][(i >> 2) & 1][4 * (i & 3) + j] += f * g[i][j];
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = it.q_first + wave * 64 + 32 * qb + r;
        if (qr < a.Sq) {
            bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + ((int64_t)it.b * a.Sq + qr) * a.ldo + it.head * 64;
            store_o64(acc[qb], 1.0f / L[qb], O, h, a.wide_o != 0);
        }
    }
}

#ifndef Q64_TRACE
#define Q64_TRACE 0            // 1 (diagnostic builds): s_memrealtime stamps per item of the persistent kernel (ltx_dbg_q64_trace)
#endif
#if Q64_TRACE
__device__ unsigned long long q64_trace[512 * 8 * 4];   // [workgroup][item][start, block done, published, merged] in 10 ns ticks
#ifndef Q64_TR_MASK
#define Q64_TR_MASK 15
#endif
#define Q64_TR(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && i - i0 < 8) q64_trace[(blockIdx.x * 8 + (i - i0)) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define Q64_TR(slot) do { } while (0)
#endif

__global__ __launch_bounds__(256, 1) void attn_q64_persist_kernel(const AttnArgs a, const Q64Item* __restrict__ items, const int* __restrict__ first,
                                                                   float* slabs, unsigned* cnt) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * TILE_BYTES + 16];
    volatile unsigned* last_flag = reinterpret_cast<volatile unsigned*>(smem + FLAG_OFF + 4);
    const int i0 = first[blockIdx.x], i1 = first[blockIdx.x + 1];
    for (int i = i0; i < i1; ++i) {
        const Q64Item it = items[i];
        if (i > i0) __syncthreads();                        // every wave is done with the ring (and the flags) of the previous item
        Q64_TR(0);
        if (it.kind == 2) {
            float* slab = it.nparts > 1 ? slabs + (int64_t)(it.slab0 + it.part) * SLAB_F : nullptr;
            attn_q64_block<2>(a, smem, it.b, it.head, it.q_first, it.k_row0, it.Sk, slab);
            Q64_TR(1);
            if (slab) {
                // publish: every storing wave drains its sc1 stores, the workgroup meets, ONE lane draws the ticket
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    const unsigned ticket = __hip_atomic_fetch_add(cnt + it.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned last = ticket == (unsigned)it.nparts - 1u;
                    if (last) __hip_atomic_store(cnt + it.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // handed back at zero
                    *last_flag = last;
                }
                __syncthreads();
                Q64_TR(2);
                if (*last_flag) attn_q64_merge(a, it, slabs);
#if Q64_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                Q64_TR(3);
            }
        } else {
            attn_q64_block<1>(a, smem, it.b, it.head, it.q_first, it.k_row0, it.Sk, nullptr);
            Q64_TR(1);
        }
    }
}


// ---- stream form (round 6): the persistent list above with the seams between items hidden -------------------------------------
// A workgroup's items run as one continuous K/V/Q stream: the last four loop iterations of an item request the NEXT item's first
// four key tiles into the ring slots they free (where the block grid requests out-of-range zeros), its Q^T goes by LDS-DMA into a
// per-wave staging area laid out like a K tile (so its fragments are K fragment reads) when the loop ends, and the epilogue runs under
// their flight: a block start costs the S^T(0) chain and the row maxima instead of a cold 96-KiB burst that all 256 CUs of a
// lockstep grid issue at once (5.5 us per block).  The tail of the grid (S = 4992: 2.44 rounds) is cut into key-range
// parts of 256-query blocks, merged by the last arriver (attn_q64_merge): every CU gets 2.44 blocks' worth of tiles.
// One generated statement per item (q64_stream_qb2 / q64_stream_part_qb2); what crosses from one to the next is LDS contents and
// outstanding vector-memory operations only - nothing the compiler sees.  Every scalar of an item is host-built (Q64SItem::w).
struct Q64SItem { uint32_t w[16]; int kind, b, head, q_first, k_row0, Sk, slab0, part, nparts, cnt, pad[6]; };     // kind: 2 whole block, 3 part, 1 128-query block (QB = 1, cold)
constexpr int QS_OFF = NSLOT * TILE_BYTES + 1024;          // Q^T staging: 8 KiB per wave, behind the ring and its flag words

__global__ __launch_bounds__(256, 1) void attn_q64_stream_kernel(const AttnArgs a, const Q64SItem* __restrict__ items, const int* __restrict__ first,
                                                                  float* slabs, unsigned* cnt) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[QS_OFF + 4 * 8192];
    volatile unsigned* flagw = reinterpret_cast<volatile unsigned*>(smem + FLAG_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, lr = lane >> 3, pc = lane & 7;
    const int i0 = __builtin_amdgcn_readfirstlane(first[blockIdx.x]), i1 = __builtin_amdgcn_readfirstlane(first[blockIdx.x + 1]);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // whole-tensor descriptors: every item offset (batch, head, first row) rides in the scalar offset of the loads / stores
    auto desc = [&](const void* p, uint32_t bytes) {
        const uint64_t u = (uint64_t)(uintptr_t)p;
        return (u32x4){(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) & 0xffffu,
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    };
    const u32x4 rk = desc(a.k, (uint32_t)a.B * (uint32_t)a.Sk * (uint32_t)a.ldk * 2u), rv = desc(a.v, (uint32_t)a.B * (uint32_t)a.Sk * (uint32_t)a.ldv * 2u);
    const u32x4 rq = desc(a.q, (uint32_t)a.B * (uint32_t)a.Sq * (uint32_t)a.ldq * 2u), ro_full = desc(a.o, (uint32_t)a.B * (uint32_t)a.Sq * (uint32_t)a.ldo * 2u);
    const uint32_t kstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldk * 2u));
    const uint32_t vstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)BKV * (uint32_t)a.ldv * 2u));
    const uint32_t qstep8 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(8u * (uint32_t)a.ldq * 2u));
    const uint32_t ldsw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)wave * (PW * 1024)));
    const uint32_t ldsq = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_base + (uint32_t)QS_OFF + (uint32_t)wave * 8192u));
    int n_item = -1;
    auto tr = [&](int slot) {                               // diagnostic builds (-DQ64_TRACE=1): 10-ns stamps per item
#if Q64_TRACE
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (((Q64_TR_MASK >> slot) & 1) && tid == 0 && blockIdx.x < 512 && n_item < 8) q64_trace[(blockIdx.x * 8 + n_item) * 4 + slot] = t;
#else
        (void)slot;
#endif
    };
    bool force_cold = false;                               // block-uniform: the item before this one ran the exact pass (the ring was re-used)
    for (int i = i0; i < i1; ++i) {
        const Q64SItem* itp = items + __builtin_amdgcn_readfirstlane(i);
        const int kind = __builtin_amdgcn_readfirstlane(itp->kind);
        n_item = __builtin_amdgcn_readfirstlane(n_item + 1);
        tr(0);
#if Q64_TRACE
        if (((Q64_TR_MASK >> 4) & 1) && tid == 0 && blockIdx.x < 512 && n_item < 8)        // what the item is: kind, key tiles, part / parts
            q64_trace[(blockIdx.x * 8 + n_item) * 4 + 3] = ((unsigned long long)kind << 48) | ((unsigned long long)(itp->Sk / 64) << 32) | ((unsigned long long)itp->part << 16) | (unsigned long long)itp->nparts;
#endif
        if (kind == 1) {                               // 128-query block: the self-contained QB = 1 form (last in a list: nothing was requested ahead)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            attn_q64_block<1>(a, smem, itp->b, itp->head, itp->q_first, 0, a.Sk, nullptr);
            force_cold = true;
            tr(1);
            continue;
        }
        // lane vectors (re-derived per item: as loop invariants they would be spilled across the statement, and a spill's reload
        // drains vmcnt - the requests in flight for the next item)
        u32x16 lanes; u32x4 qbase, ones_u;
        {
            const int rr_ = __builtin_amdgcn_readfirstlane(0) + r;      // (opaque zero: keeps the arithmetic inside the loop)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lanes[ks] = smem_base + rr_ * KROW + kswz8(rr_, 2 * ks + h) * 16;
                qbase[ks] = smem_base + QS_OFF + wave * 8192 + rr_ * KROW + kswz8(rr_, 2 * ks + h) * 16;
            }
            const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int row = 4 * h + trq, cv = d * 4 + trdh * 2 + (trp >> 1);
                lanes[4 + d] = smem_base + row * VROW + vswz8(row, cv) * 16 + (trp & 1) * 8;
            }
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                const int row = (wave * PW + j) * 8 + lr;
                lanes[6 + j] = (uint32_t)row * (uint32_t)a.ldk * 2u + (uint32_t)kswz8(row, pc) * 16u;
                lanes[8 + j] = (uint32_t)row * (uint32_t)a.ldv * 2u + (uint32_t)vswz8(row, pc) * 16u;
            }
            // Q^T pieces: piece j = rows 8 j .. 8 j + 7 of the wave's 64 queries; the K image's swizzle on the SOURCE chunk (even / odd j differ in bit 2)
            const uint32_t qrow = (uint32_t)(wave * 64 + lr) * (uint32_t)a.ldq * 2u;
            lanes[10] = qrow + (uint32_t)(pc ^ (lr >> 1)) * 16u;
            lanes[11] = qrow + (uint32_t)(pc ^ (lr >> 1) ^ 4) * 16u;
            lanes[14] = (lane & 16) ? 1u : 0u;
            lanes[15] = 0u;
            const bf16_t v1 = (bf16_t)(((lane & 1) == ((lane >> 4) & 1)) ? 1.0f : 0.0f);
            union { bf16x8 v; uint32_t u[4]; } c;
#pragma unroll
            for (int e = 0; e < 8; ++e) c.v[e] = v1;
            ones_u = (u32x4){c.u[0], c.u[1], c.u[2], c.u[3]};
        }
        u32x16 item;
#pragma unroll
        for (int e = 0; e < 16; ++e) item[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)itp->w[e]);
        if (force_cold) { item[11] = 1u; force_cold = false; }
#pragma unroll
        for (int e = 0; e < 16; ++e) item[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)item[e]);
        const bool part = kind == 3;
        const int slab_i = __builtin_amdgcn_readfirstlane(itp->slab0 + itp->part);
        f32x8 la;
        if (part) {
            float* slab = slabs + (int64_t)slab_i * SLAB_F;
            const uint64_t sp = (uint64_t)(uintptr_t)(slab + wave * SLAB_WAVE_F);
            const u32x4 ro_slab = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sp), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sp >> 32)) & 0xffffu,
                                   (uint32_t)(SLAB_WAVE_F * 4), 0x00020000u};
            lanes[12] = (uint32_t)lane * 16u; lanes[13] = 0u;
            q64_stream_part_qb2(la, ones_u, lanes, qbase, item, rk, rv, rq, ro_slab, kstep, vstep, ldsw, ldsq, qstep8);
        } else {
            lanes[12] = (uint32_t)(wave * 64 + r) * (uint32_t)a.ldo * 2u + 16u * h;
            lanes[13] = (uint32_t)(wave * 64 + 32 + r) * (uint32_t)a.ldo * 2u + 16u * h;
            q64_stream_qb2(la, ones_u, lanes, qbase, item, rk, rv, rq, ro_full, kstep, vstep, ldsw, ldsq, qstep8);
        }
        tr(1);
        const float l0 = (lane & 16) ? la[1] : la[0], l1 = (lane & 16) ? la[5] : la[4];
        const bool bad = !(l0 < 0x1p100f) || !(l1 < 0x1p100f);
        // block-uniform OR through one LDS word, with bare barriers: __syncthreads() carries a fence that would drain vmcnt
        if (tid == 0) *flagw = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (__any(bad) && lane == 0) *flagw = 1u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool any_bad = __builtin_amdgcn_readfirstlane((int)*flagw) != 0;      // (block-uniform, and known to be: the item record is SGPR data)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // (the word is rewritten by the next item)
        if (!Q64_NO_FALLBACK && any_bad) {                 // overflow of the fixed max: the self-contained form with its exact second pass
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            float* slab = part ? slabs + (int64_t)slab_i * SLAB_F : nullptr;
            attn_q64_block<2>(a, smem, itp->b, itp->head, itp->q_first, itp->k_row0, itp->Sk, slab);
            force_cold = true;
        }
        if (part) {
            // publish: every storing wave drains its sc1 stores, the workgroup meets, ONE lane draws the ticket (attn_q64_persist_kernel)
            volatile unsigned* last_flag = reinterpret_cast<volatile unsigned*>(smem + FLAG_OFF + 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(cnt + itp->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned last = ticket == (unsigned)itp->nparts - 1u;
                if (last) __hip_atomic_store(cnt + itp->cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *last_flag = last;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool is_last = __builtin_amdgcn_readfirstlane((int)*last_flag) != 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tr(2);
            if (is_last) {
                Q64Item mi; mi.b = itp->b; mi.head = itp->head; mi.q_first = itp->q_first; mi.kind = 2; mi.k_row0 = 0; mi.Sk = a.Sk; mi.slab0 = itp->slab0; mi.part = itp->part; mi.nparts = itp->nparts; mi.cnt = itp->cnt;
                attn_q64_merge(a, mi, slabs);
#if Q64_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                tr(3);
            }
        }
    }
}

}  // namespace
#if Q64_TRACE
extern "C" int ltx_dbg_q64_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(q64_trace), sizeof(unsigned long long) * (size_t)n);
}
#endif

// ---- host side of the persistent form: the static item lists ----------------------------------------------------------
namespace {
struct Q64Plan { Q64Item* items = nullptr; int* first = nullptr; int grid = 0, nslab = 0, ncnt = 0; };
struct Q64Ws { float* slabs = nullptr; size_t slab_f = 0; unsigned* cnt = nullptr; size_t ncnt = 0; };
std::map<std::tuple<int, int, int, int, int, int>, Q64Plan> g_q64_plans;     // (device, B, heads, Sq, Sk, xcd order)
std::map<std::pair<int, hipStream_t>, Q64Ws> g_q64_ws;
std::mutex g_q64_mu;

constexpr int Q64_SMALL_PCT = 70;        // a 128-query block's key tile costs 70 % of a 256-query block's (measured)
constexpr int Q64_MIN_PART = 8;          // key tiles: shorter parts are not worth a publish + merge (a block has at most 4 parts)
constexpr int Q64_PART_OVH = 8;          // what a part costs beyond its tiles, in key tiles of a 256-query block (prologue, slab publish, merge duty)

// Cuts the blocks of the (batch, head) pairs `bh` over `ncu` workgroups.  Every workgroup's list: its parts first (the
// merges then happen early, off the tail of the launch), then its whole blocks.
void q64_schedule_group(const std::vector<std::pair<int, int>>& bh, int Sq, int Sk, int ncu, std::vector<std::vector<Q64Item>>& lists,
                        int& nslab, int& ncnt, int unit = 1, int small_pct = Q64_SMALL_PCT, int part_ovh = Q64_PART_OVH, int part_step_pct = 100) {
    // unit: key tiles per scheduling step (the stream form cuts at even tiles: 2); costs are in steps (a whole block's step = 100).
    // A part of s steps costs s * part_step_pct + ovh * 100 (the stream form: measured, tools/attn_stream_trace.py)
    const int nt = Sk / (64 * unit);
    const int min_part = Q64_MIN_PART / unit > 1 ? Q64_MIN_PART / unit : 1, ovh = part_ovh / unit;
    struct Blk { int b, head, q_first; };
    std::vector<Blk> bigs, smalls;
    const int nfull = Sq / 256, rest = Sq - nfull * 256;
    // head-major: the blocks in flight at any time belong to few heads (their K/V stay in the XCD's L2); the tape (the
    // tail of this list) is cut from the last head(s)
    for (const auto& p : bh) {
        for (int i = 0; i < nfull + (rest > 128 ? 1 : 0); ++i) bigs.push_back({p.first, p.second, i * 256});
        if (rest > 0 && rest <= 128) smalls.push_back({p.first, p.second, nfull * 256});
    }
    const int64_t big_cost = (int64_t)nt * 100, small_cost = (int64_t)nt * small_pct;
    const int64_t total = big_cost * (int64_t)bigs.size() + small_cost * (int64_t)smalls.size();
    const int64_t target = (total + ncu - 1) / ncu;
    std::vector<int64_t> load(ncu, 0);
    std::vector<std::vector<Q64Item>> whole(ncu);
    for (size_t i = 0; i < smalls.size(); ++i) {
        const int c = (int)(i % ncu);
        whole[c].push_back({smalls[i].b, smalls[i].head, smalls[i].q_first, 1, 0, Sk, 0, 0, 1, 0});
        load[c] += small_cost;
    }
    size_t nb = 0;                                          // whole bigs: as many as fit under the target, dealt round by round
    for (bool any = true; any;) {
        any = false;
        for (int c = 0; c < ncu && nb < bigs.size(); ++c)
            if (load[c] + big_cost <= target) {
                whole[c].push_back({bigs[nb].b, bigs[nb].head, bigs[nb].q_first, 2, 0, Sk, 0, 0, 1, 0});
                load[c] += big_cost; ++nb; any = true;
            }
    }
    // the tape: the remaining bigs.  Every workgroup takes at most ONE part (a part costs a prologue and a slab publish on top
    // of its tiles: Q64_PART_OVH), every block is covered by 1..4 workgroups.  The smallest makespan T for which that works
    // is found by bisection; for a given T a workgroup can take cap = (T - load) / 100 - overhead tiles, and the blocks are
    // covered greedily: the roomiest workgroup left, completed by the tightest one that still closes the block.
    const int ntape = (int)(bigs.size() - nb);
    struct Cut { int cu, t0, t1; };
    std::vector<std::vector<Cut>> cuts(ntape);
    auto cover = [&](int64_t T, bool commit) {
        std::vector<std::pair<int, int>> caps;             // (cap tiles, cu), ascending
        for (int c = 0; c < ncu; ++c) {
            int64_t cap = (T - load[c] - (int64_t)ovh * 100) / part_step_pct;
            if (cap >= nt) cap = nt; else if (cap < min_part) continue;
            caps.push_back({(int)cap, c});
        }
        std::sort(caps.begin(), caps.end());
        std::vector<char> used(caps.size(), 0);
        int hi = (int)caps.size() - 1;
        for (int k = 0; k < ntape; ++k) {
            std::vector<std::pair<int, int>> team;           // (cap, cu)
            int sum = 0;
            while (sum < nt && (int)team.size() < 4) {
                while (hi >= 0 && used[hi]) --hi;
                if (hi < 0) return false;
                // the tightest unused workgroup that closes the block, else the roomiest
                const int need = nt - sum;
                int pick = -1;
                if (!team.empty()) {
                    auto it = std::lower_bound(caps.begin(), caps.end(), std::make_pair(need, -1));
                    for (int j = (int)(it - caps.begin()); j < (int)caps.size(); ++j) if (!used[j]) { pick = j; break; }
                }
                if (pick < 0) pick = hi;
                used[pick] = 1; team.push_back(caps[pick]); sum += caps[pick].first;
            }
            if (sum < nt) return false;
            if (!commit) continue;
            // cut the block in proportion to the caps (every part <= its cap, >= Q64_MIN_PART by construction of caps unless scaled below)
            std::sort(team.begin(), team.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.second < y.second; });
            int pos = 0; double acc = 0.0;
            for (size_t i = 0; i < team.size(); ++i) {
                acc += (double)team[i].first * nt / sum;
                int end = i + 1 == team.size() ? nt : (int)(acc + 0.5);
                if (end - pos < 1) end = pos + 1;
                if (end > nt) end = nt;
                cuts[k].push_back({team[i].second, pos, end});
                pos = end;
            }
        }
        return true;
    };
    if (ntape > 0) {
        int64_t lo = target, hi = target + (int64_t)(2 * nt + ovh) * 100;
        while (!cover(hi, false)) hi += (int64_t)nt * 100;                    // always ends: with cap = nt every block is one workgroup's
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (cover(mid, false)) hi = mid; else lo = mid + 1; }
        (void)cover(hi, true);
    }
    lists.assign(ncu, {});
    for (int k = 0; k < ntape; ++k) {
        const Blk& bk = bigs[nb + k];
        const int np = (int)cuts[k].size();
        const int slab0 = np > 1 ? nslab : 0, cn = np > 1 ? ncnt : 0;
        if (np > 1) { nslab += np; ++ncnt; }
        for (int p = 0; p < np; ++p)
            lists[cuts[k][p].cu].push_back({bk.b, bk.head, bk.q_first, 2, cuts[k][p].t0 * 64 * unit, (cuts[k][p].t1 - cuts[k][p].t0) * 64 * unit, slab0, p, np, cn});
    }
    for (int c = 0; c < ncu; ++c) for (const auto& w : whole[c]) lists[c].push_back(w);
}
}  // namespace

#ifdef LTX_EXPERIMENTS
// tuning / test aid: the schedule as text ("cu: kind head q_first k_row0 Sk part/nparts | ...")
extern "C" int ltx_dbg_q64_schedule(int B, int heads, int Sq, int Sk, int n_cu, int xcd, char* out, int cap) {
    std::vector<std::vector<Q64Item>> all(n_cu);
    int nslab = 0, ncnt = 0;
    const int G = xcd ? 8 : 1;
    for (int x = 0; x < G; ++x) {
        std::vector<std::pair<int, int>> bh;
        for (int b = 0; b < B; ++b) for (int h = 0; h < heads; ++h) if (!xcd || (h & 7) == x) bh.push_back({b, h});
        std::vector<std::vector<Q64Item>> lists;
        q64_schedule_group(bh, Sq, Sk, n_cu / G, lists, nslab, ncnt);
        for (int c = 0; c < n_cu / G; ++c) all[c * G + x] = lists[c];
    }
    std::string t;
    for (int c = 0; c < n_cu; ++c) {
        t += std::to_string(c) + ":";
        for (const auto& it : all[c]) t += " " + std::to_string(it.kind) + "," + std::to_string(it.b) + "," + std::to_string(it.head) + "," + std::to_string(it.q_first) + "," + std::to_string(it.k_row0) + "," +
                                          std::to_string(it.Sk) + "," + std::to_string(it.part) + "/" + std::to_string(it.nparts) + "," + std::to_string(it.slab0) + "," + std::to_string(it.cnt);
        t += "\n";
    }
    if ((int)t.size() + 1 > cap) return (int)t.size() + 1;
    memcpy(out, t.c_str(), t.size() + 1);
    return 0;
}
#endif  // LTX_EXPERIMENTS

static int q64_n_cu() {
    static std::mutex mu; static std::map<int, int> per_dev;
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_dev.find(dev);
    if (it != per_dev.end()) return it->second;
    hipDeviceProp_t p; int n = 256;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    per_dev[dev] = n;
    return n;
}

// 32-bit buffer arithmetic of the kernels above: every row offset (and the "dropped row" offset 0x80000000) must stay below 2^31
bool ltx_attention_q64_fits(const AttnArgs& a) {
    const double lim = 2147483648.0 - 256.0;
    return (double)a.Sk * a.ldk * 2.0 < lim && (double)a.Sk * a.ldv * 2.0 < lim && (double)a.Sq * a.ldq * 2.0 < lim && (double)a.Sq * a.ldo * 2.0 < lim;
}

static int launch_q64_persist(const AttnArgs& a, int n_cu, hipStream_t s) {
    int dev = 0; (void)hipGetDevice(&dev);
    const int xcd = (a.xcd_heads && n_cu % 8 == 0) ? 1 : 0;
    Q64Plan plan; Q64Ws ws;
    {
        std::lock_guard<std::mutex> lock(g_q64_mu);
        const auto key = std::make_tuple(dev, a.B, a.heads, a.Sq, a.Sk, xcd);
        auto it = g_q64_plans.find(key);
        if (it == g_q64_plans.end()) {
            std::vector<Q64Item> flat; std::vector<int> first(n_cu + 1, 0);
            std::vector<std::vector<Q64Item>> all(n_cu);
            int nslab = 0, ncnt = 0;
            const int G = xcd ? 8 : 1;
            for (int x = 0; x < G; ++x) {
                std::vector<std::pair<int, int>> bh;
                for (int b = 0; b < a.B; ++b) for (int h = 0; h < a.heads; ++h) if (!xcd || (h & 7) == x) bh.push_back({b, h});
                std::vector<std::vector<Q64Item>> lists;
                q64_schedule_group(bh, a.Sq, a.Sk, n_cu / G, lists, nslab, ncnt);
                for (int c = 0; c < n_cu / G; ++c) all[c * G + x] = lists[c];      // blocks b and b + 8 share an XCD
            }
            for (int c = 0; c < n_cu; ++c) { first[c] = (int)flat.size(); flat.insert(flat.end(), all[c].begin(), all[c].end()); }
            first[n_cu] = (int)flat.size();
            Q64Plan np; np.grid = n_cu; np.nslab = nslab; np.ncnt = ncnt;
            HIP_TRY(hipMalloc(&np.items, flat.size() * sizeof(Q64Item) + 16));
            HIP_TRY(hipMalloc(&np.first, first.size() * sizeof(int)));
            HIP_TRY(hipMemcpy(np.items, flat.data(), flat.size() * sizeof(Q64Item), hipMemcpyHostToDevice));      // once per shape (ltx_warmup)
            HIP_TRY(hipMemcpy(np.first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
            it = g_q64_plans.emplace(key, np).first;
        }
        plan = it->second;
        Q64Ws& w = g_q64_ws[std::make_pair(dev, s)];
        const size_t need_f = (size_t)(plan.nslab > 0 ? plan.nslab : 1) * SLAB_F, need_c = (size_t)(plan.ncnt > 0 ? plan.ncnt : 1);
        if (w.slab_f < need_f) { if (w.slabs) (void)hipFree(w.slabs); w.slabs = nullptr; HIP_TRY(hipMalloc(&w.slabs, need_f * sizeof(float))); w.slab_f = need_f; }
        if (w.ncnt < need_c) {
            const size_t n = need_c < 1024 ? 1024 : need_c;
            if (w.cnt) (void)hipFree(w.cnt);
            w.cnt = nullptr; HIP_TRY(hipMalloc(&w.cnt, n * sizeof(unsigned))); w.ncnt = n;
            HIP_TRY(hipMemsetAsync(w.cnt, 0, n * sizeof(unsigned), s));      // once: every merger hands its counter back at zero
        }
        ws = w;
    }
    hipLaunchKernelGGL(attn_q64_persist_kernel, dim3((unsigned)plan.grid), dim3(256), 0, s, a, plan.items, plan.first, ws.slabs, ws.cnt);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

// ---- host side of the stream form: the persistent lists + everything the generated statement reads from an item record -------
namespace {
struct Q64SPlan { Q64SItem* items = nullptr; int* first = nullptr; int grid = 0, nslab = 0, ncnt = 0; };
std::map<std::tuple<int, int, int, int, int, int, int, int, int, int>, Q64SPlan> g_q64s_plans;     // (device, B, heads, Sq, Sk, xcd order, ldq, ldk, ldv, ldo)
}  // namespace

namespace {
// Stream schedule of the (batch, head) pairs `bh` over `ncu` workgroups (one XCD's share).  Whole 256-query blocks are dealt while they
// fit under the finish time; the rest lie on a TAPE of key tiles (block after block) that is cut once per workgroup - a workgroup's
// stretch may cross ONE block boundary, it then runs two parts - so every workgroup ends at the same time up to a two-tile step (the
// one-part-per-workgroup rule of the persistent form left a 26-us spread: profiles/r6_attn_stream_trace.json).  Times in microseconds
// from in-kernel stamps (tools/attn_stream_trace.py).  Lists: parts (tape order), whole blocks, the 128-query block.
void q64_stream_schedule(const std::vector<std::pair<int, int>>& bh, int Sq, int Sk, int ncu, std::vector<std::vector<Q64Item>>& lists, int& nslab, int& ncnt) {
    const int nt = Sk / 64;                                  // even (ltx_attention_q64_stream_ok)
    const double whole_us = 0.806 * nt + 2.6, small_us = 0.617 * nt, part_tile_us = 0.68, part1_us = 17.0, part2_us = 10.0;
    const int min_part = 8;
    struct Blk { int b, head, q_first; };
    std::vector<Blk> bigs, smalls;
    const int nfull = Sq / 256, rest = Sq - nfull * 256;
    for (const auto& p : bh) {
        for (int i = 0; i < nfull; ++i) bigs.push_back({p.first, p.second, i * 256});
        if (rest > 0) smalls.push_back({p.first, p.second, nfull * 256});          // rest == 128 (stream_ok)
    }
    lists.assign(ncu, {});
    std::vector<int> nsmall(ncu, 0);
    for (size_t i = 0; i < smalls.size(); ++i) ++nsmall[ncu - 1 - (int)(i % ncu)];
    const double total_us = whole_us * bigs.size() + small_us * smalls.size();
    // whole blocks per workgroup: as many as leave at least a minimal part's room under the even share
    const double share = total_us / ncu;
    std::vector<int> nwhole(ncu, 0);
    size_t used = 0;
    for (bool any = true; any;) {
        any = false;
        for (int c = 0; c < ncu && used < bigs.size(); ++c)
            if (nsmall[c] * small_us + (nwhole[c] + 1) * whole_us <= share) { ++nwhole[c]; ++used; any = true; }
    }
    const int ntape = (int)(bigs.size() - used);
    const int64_t tape_tiles = (int64_t)ntape * nt;
    struct Seg { int cu, blk, t0, t1; };
    std::vector<Seg> segs;
    auto cut = [&](double T, bool commit) {
        int64_t pos = 0;                                       // tape position in tiles
        if (commit) segs.clear();
        for (int c = 0; c < ncu && pos < tape_tiles; ++c) {
            double room = T - nsmall[c] * small_us - nwhole[c] * whole_us - part1_us;
            int cap = (int)(room / part_tile_us); cap &= ~1;
            if (cap < min_part) continue;
            const int in_blk = (int)(pos % nt), left = nt - in_blk;         // tiles left of the block the tape stands in
            int take1 = cap < left ? cap : left;
            if (left - take1 > 0 && left - take1 < min_part) take1 = left;  // never leave a sliver of a block: this workgroup runs a little over
            if (commit) segs.push_back({c, (int)(pos / nt), in_blk, in_blk + take1});
            pos += take1;
            int cap2 = ((int)((room - take1 * part_tile_us - part2_us) / part_tile_us)) & ~1;
            if (take1 == left && cap2 >= min_part && pos < tape_tiles) {    // room for a stretch of the next block
                int take2 = cap2 < nt ? cap2 : nt;
                if (nt - take2 > 0 && nt - take2 < min_part) take2 = nt - min_part;
                if (take2 >= min_part) { if (commit) segs.push_back({c, (int)(pos / nt), 0, take2}); pos += take2; }
            }
        }
        return pos >= tape_tiles;
    };
    if (ntape > 0) {
        double lo = share, hi = share + 2.0 * whole_us;
        while (!cut(hi, false)) hi += whole_us;
        for (int it = 0; it < 40; ++it) { const double mid = 0.5 * (lo + hi); if (cut(mid, false)) hi = mid; else lo = mid; }
        (void)cut(hi, true);
    }
    // parts of a tape block: slab / counter indices
    std::vector<int> nparts(ntape, 0), slab0(ntape, 0), cnt0(ntape, 0);
    for (const auto& sg : segs) ++nparts[sg.blk];
    for (int k = 0; k < ntape; ++k) if (nparts[k] > 1) { slab0[k] = nslab; nslab += nparts[k]; cnt0[k] = ncnt++; }
    std::vector<int> seen(ntape, 0);
    for (const auto& sg : segs) {
        const Blk& bk = bigs[used + sg.blk];
        lists[sg.cu].push_back({bk.b, bk.head, bk.q_first, 2, sg.t0 * 64, (sg.t1 - sg.t0) * 64, slab0[sg.blk], seen[sg.blk]++, nparts[sg.blk], cnt0[sg.blk]});
    }
    size_t nb = 0;
    for (int round = 0; nb < used; ++round)
        for (int c = 0; c < ncu && nb < used; ++c)
            if (round < nwhole[c]) { lists[c].push_back({bigs[nb].b, bigs[nb].head, bigs[nb].q_first, 2, 0, Sk, 0, 0, 1, 0}); ++nb; }
    size_t ns = 0;
    for (int c = ncu - 1; c >= 0 && ns < smalls.size(); --c)
        for (int j = 0; j < nsmall[c]; ++j, ++ns) lists[c].push_back({smalls[ns].b, smalls[ns].head, smalls[ns].q_first, 1, 0, Sk, 0, 0, 1, 0});
}
}  // namespace

bool ltx_attention_q64_stream_ok(const AttnArgs& a, int n_cu) {
    if (!ltx_opt().attn_q64_stream || ltx_opt().attn_q64_big >= 0) return false;
    if (a.Sk % 128 != 0 || a.Sq % 128 != 0 || a.Sk < 64 * 2 * Q64_MIN_PART || a.B * a.heads * ((a.Sq + 255) / 256) <= n_cu) return false;
    const double lim = 2147483648.0 - 65536.0;               // whole-tensor descriptors, 32-bit scalar offsets
    return (double)a.B * a.Sk * a.ldk * 2.0 < lim && (double)a.B * a.Sk * a.ldv * 2.0 < lim && (double)a.B * a.Sq * a.ldq * 2.0 < lim && (double)a.B * a.Sq * a.ldo * 2.0 < lim;
}

static int launch_q64_stream(const AttnArgs& a, int n_cu, hipStream_t s) {
    int dev = 0; (void)hipGetDevice(&dev);
    const int xcd = (a.xcd_heads && n_cu % 8 == 0) ? 1 : 0;
    Q64SPlan plan; Q64Ws ws;
    {
        std::lock_guard<std::mutex> lock(g_q64_mu);
        const auto key = std::make_tuple(dev, a.B, a.heads, a.Sq, a.Sk, xcd, a.ldq, a.ldk, a.ldv, a.ldo);
        auto it = g_q64s_plans.find(key);
        if (it == g_q64s_plans.end()) {
            std::vector<std::vector<Q64Item>> all(n_cu);
            int nslab = 0, ncnt = 0;
            const int G = xcd ? 8 : 1;
            for (int x = 0; x < G; ++x) {
                std::vector<std::pair<int, int>> bh;
                for (int b = 0; b < a.B; ++b) for (int h = 0; h < a.heads; ++h) if (!xcd || (h & 7) == x) bh.push_back({b, h});
                std::vector<std::vector<Q64Item>> lists;
                q64_stream_schedule(bh, a.Sq, a.Sk, n_cu / G, lists, nslab, ncnt);
                for (int c = 0; c < n_cu / G; ++c) all[c * G + x] = lists[c];      // blocks b and b + 8 share an XCD
            }
            std::vector<Q64SItem> flat; std::vector<int> first(n_cu + 1, 0);
            for (int c = 0; c < n_cu; ++c) {
                first[c] = (int)flat.size();
                // order inside a list: parts (their merges then happen early), whole 256-query blocks, the 128-query block last (it runs cold)
                std::vector<Q64Item> ord;
                for (const auto& q : all[c]) if (q.kind == 2 && q.nparts > 1) ord.push_back(q);
                for (const auto& q : all[c]) if (q.kind == 2 && q.nparts == 1) ord.push_back(q);
                for (const auto& q : all[c]) if (q.kind == 1) ord.push_back(q);
                uint32_t rot = 0;
                for (size_t i = 0; i < ord.size(); ++i) {
                    const Q64Item& q = ord[i];
                    Q64SItem o; memset(&o, 0, sizeof(o));
                    o.kind = q.kind == 1 ? 1 : (q.nparts > 1 ? 3 : 2);
                    o.b = q.b; o.head = q.head; o.q_first = q.q_first; o.k_row0 = q.k_row0; o.Sk = q.Sk; o.slab0 = q.slab0; o.part = q.part; o.nparts = q.nparts; o.cnt = q.cnt;
                    if (o.kind != 1) {
                        const uint32_t nt = (uint32_t)q.Sk / 64u;
                        if (nt < 6 || (nt & 1) || (q.k_row0 % 128) != 0 || q.nparts > 8) return -2;      // (not a schedule the stream statement runs: the block grid serves)
                        auto koff = [&](const Q64Item& t, int ld) { return (uint32_t)((((int64_t)t.b * a.Sk + t.k_row0) * ld + t.head * 64) * 2); };
                        auto qoff = [&](const Q64Item& t, int ld) { return (uint32_t)((((int64_t)t.b * a.Sq + t.q_first) * ld + t.head * 64) * 2); };
                        o.w[0] = koff(q, a.ldk); o.w[1] = koff(q, a.ldv); o.w[2] = qoff(q, a.ldq); o.w[3] = qoff(q, a.ldo);
                        const bool has_next = i + 1 < ord.size() && ord[i + 1].kind != 1;
                        o.w[4] = has_next ? koff(ord[i + 1], a.ldk) : 0x80000000u;
                        o.w[5] = has_next ? koff(ord[i + 1], a.ldv) : 0x80000000u;
                        o.w[6] = has_next ? qoff(ord[i + 1], a.ldq) : 0x80000000u;
                        o.w[7] = nt - 4u; o.w[8] = rot; o.w[9] = (rot + nt) & 3u; o.w[10] = rot * 16384u;
                        o.w[11] = i == 0 ? 1u : 0u;
                        o.w[12] = (i > 0 && ord[i - 1].nparts > 1) ? 17u : 8u;
                        rot = (rot + nt) & 3u;
                    }
                    flat.push_back(o);
                }
            }
            first[n_cu] = (int)flat.size();
            Q64SPlan np; np.grid = n_cu; np.nslab = nslab; np.ncnt = ncnt;
            HIP_TRY(hipMalloc(&np.items, flat.size() * sizeof(Q64SItem) + 16));
            HIP_TRY(hipMalloc(&np.first, first.size() * sizeof(int)));
            HIP_TRY(hipMemcpy(np.items, flat.data(), flat.size() * sizeof(Q64SItem), hipMemcpyHostToDevice));      // once per shape (ltx_warmup)
            HIP_TRY(hipMemcpy(np.first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
            it = g_q64s_plans.emplace(key, np).first;
        }
        plan = it->second;
        Q64Ws& w = g_q64_ws[std::make_pair(dev, s)];
        const size_t need_f = (size_t)(plan.nslab > 0 ? plan.nslab : 1) * SLAB_F, need_c = (size_t)(plan.ncnt > 0 ? plan.ncnt : 1);
        if (w.slab_f < need_f) { if (w.slabs) (void)hipFree(w.slabs); w.slabs = nullptr; HIP_TRY(hipMalloc(&w.slabs, need_f * sizeof(float))); w.slab_f = need_f; }
        if (w.ncnt < need_c) {
            const size_t n = need_c < 1024 ? 1024 : need_c;
            if (w.cnt) (void)hipFree(w.cnt);
            w.cnt = nullptr; HIP_TRY(hipMalloc(&w.cnt, n * sizeof(unsigned))); w.ncnt = n;
            HIP_TRY(hipMemsetAsync(w.cnt, 0, n * sizeof(unsigned), s));      // once: every merger hands its counter back at zero
        }
        ws = w;
    }
    LTX_LAUNCH_TIMED(attn_q64_stream_kernel, dim3((unsigned)plan.grid), dim3(256), 0, s, a, plan.items, plan.first, ws.slabs, ws.cnt);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

// Split of a head's queries into big (256) and small (128) blocks: as many big blocks as fill whole rounds of the
// chip's CUs (one block per CU), the rest as small blocks that run in about half a big block's time, so the last
// round of the grid is short instead of running a few long blocks on a mostly idle chip.
int ltx_launch_attention_q64(const AttnArgs& a, hipStream_t s) {
    const int n_cu = q64_n_cu();
    const int heads_total = a.heads * a.B;
    if (ltx_attention_q64_stream_ok(a, n_cu)) {                  // option attn_q64_stream=0: the block grid below
        const int rc = launch_q64_stream(a, n_cu, s);
        if (rc != -2) return rc;
    }
    {
        // persistent form (LTX_ATTN_Q64_PERSIST=1; key counts in whole tiles, more work than one round of blocks).  Measured
        // on MI355X at S = 4992, 32 heads: 183 us against 178 us for the block grid below - a 128-query block costs 0.59 of a
        // 256-query block (not the 0.70 the split was sized for), and publish + merge cost a workgroup 1.4 + 4.5 (up to 9.5) us,
        // which is what halving the last round saves.  Kept as a tested option; the block grid stays the default.
#ifdef LTX_EXPERIMENTS     // x_attn_q64_persist=1 (experiment builds only)
        const bool on = ltx_exp("attn_q64_persist", 0) == 1;
        const int64_t blocks = (int64_t)heads_total * ((a.Sq + 255) / 256);
        if (on && a.Sk % 64 == 0 && a.Sk >= 64 * 2 * Q64_MIN_PART && blocks > n_cu && ltx_opt().attn_q64_big < 0) return launch_q64_persist(a, n_cu, s);
#endif
    }
    const int nbig_max = a.Sq / 256;                                   // whole big blocks per head
    int nbig = nbig_max;
    if (ltx_opt().attn_q64_big >= 0) { nbig = ltx_opt().attn_q64_big; if (nbig > nbig_max) nbig = nbig_max; }      // option attn_q64_big: tests of both block sizes
    else {
        // The split whose greedy schedule (one block per CU, big blocks first; a 128-query block costs 0.59 of a 256-query one,
        // tools/attn_q64_tune.py) finishes first.  Round 2's rule - big blocks in whole rounds - is that split at one batch
        // element (S = 4992, 32 heads: 16 big per head, 2.59 rounds for 2.44 of work) but not for batched guidance forwards
        // (64 heads: 19 big per head = 5.00 rounds against 5.18; 96 heads: 18 = 7.59 against 7.77).  Speed only: a query's
        // arithmetic does not depend on the size of its block.
        static std::mutex mu; static std::map<std::tuple<int, int, int>, int> memo;
        const auto key = std::make_tuple(a.Sq, heads_total, n_cu);
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it == memo.end()) {
            double best = 1e30; int best_b = nbig_max;
            for (int b = nbig_max; b >= 0; --b) {
                std::priority_queue<double, std::vector<double>, std::greater<double>> cu;
                for (int i = 0; i < n_cu; ++i) cu.push(0.0);
                const int64_t bigs = (int64_t)heads_total * b, smalls = (int64_t)heads_total * ((a.Sq - 256 * b + 127) / 128);
                if (bigs + smalls > 200000) break;             // (absurd shapes: keep the all-big split)
                double end = 0.0;
                for (int64_t i = 0; i < bigs + smalls; ++i) { const double t = cu.top() + (i < bigs ? 1.0 : 0.59); cu.pop(); cu.push(t); if (t > end) end = t; }
                if (end < best - 1e-9) { best = end; best_b = b; }
            }
            it = memo.emplace(key, best_b).first;
        }
        nbig = it->second;
    }
    const int rest = a.Sq - nbig * 256;
    const int nsmall = (rest + 127) / 128;
    AttnArgs ax = a;
    dim3 grid((unsigned)(heads_total * (nbig + nsmall))), block(256);
    LTX_LAUNCH_TIMED(attn_q64_kernel, grid, block, 0, s, ax, nbig, nsmall);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
