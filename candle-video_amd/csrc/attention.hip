// Fused (flash-style) attention for LtxAttention::forward (ltx_transformer.rs:648-750):
//   O = softmax(scale * Q K^T + key_bias) V   per (batch, head), non-causal.
// The reference either calls candle-flash-attn (bf16, self-attention on GPU, :699-712) or
// materialises the f32 [B,H,Sq,Sk] score matrix (:719-740); this kernel never materialises it.
//
// bf16 kernel (MFMA, gfx950):
//   * workgroup = 4 waves = 128 queries of one head; wave = 32 queries; KV tile = 64 keys,
//     K and V tiles double-buffered in LDS (register-staged prefetch, one barrier per tile);
//   * "swapped" QK^T: S^T[key][query] = K . Q^T with v_mfma_f32_32x32x16_bf16, so a lane owns ONE
//     query column and 16 keys per 32-key block -> the softmax row reduction is in-register plus
//     one cross-half exchange (lanes l and l^32 hold the other 16 keys of the same query);
//   * P never leaves registers: the S^T accumulator, rounded pairwise to bf16, is directly the
//     B operand of O^T[d][query] += V^T[d][key] . P^T[key][query]  (the accumulator's row index
//     is the contraction index; k-order inside a step is 16s + 8(j>>2) + 4h + (j&3));
//   * V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware
//     transpose read), two reads per MFMA operand, with a chunk XOR that keeps the 4x16 blocks
//     of one read on distinct banks;
//   * K tile rows are chunk-XOR-swizzled so the 32-row ds_read_b128 operand reads are conflict-free.
// f32 kernel: exact-f32 VALU flash kernel (one query per lane), used only by the f32 parity mode.
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "options.h"

#ifndef ATTN_DMA
#define ATTN_DMA 1
#endif
#ifndef ATTN_STATIC_SLOTS
#define ATTN_STATIC_SLOTS 1
#endif
#ifndef ATTN_TR_ASM
#define ATTN_TR_ASM 1
#endif
// Diagnostic: launches of the gated exact pass that actually ran (attn_q128_kernel's fixed first-tile max overflowed) since the
// last reset.  Touched only by such a launch.
__device__ unsigned long long q128_exact_launches;
int ltx_q128_fallback_read(unsigned long long* out, int reset) {
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(q128_exact_launches), sizeof(*out)));
    if (reset) { const unsigned long long z = 0; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(q128_exact_launches), &z, sizeof(z))); }
    return LTX_OK;
}

namespace {

constexpr int BQ = 128, BKV = 64;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Transpose read issued OUTSIDE the compiler's memory model: for the builtin form the wait-count pass cannot tell the
// read from the LDS-DMA writes of the other ring slots and drains vmcnt to 0 before the first one, which serialises
// the K/V prefetch with the softmax.  The asm form is waited for explicitly (lds_wait ties the registers).
template <int OFF> __device__ __forceinline__ u32x2 ds_tr_read(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void lds_wait(u32x2& a, u32x2& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int CPR> __device__ __forceinline__ int kswz(int row, int c) {
    if constexpr (CPR >= 4) { constexpr int RPB = 16 / CPR; return c ^ ((row / RPB) % CPR); }
    else return c;
}
template <int CPR> __device__ __forceinline__ int vswz(int row, int c) {
    if constexpr (CPR == 8) return c ^ (((row >> 1) & 1) << 2);
    else if constexpr (CPR == 16) return c ^ ((row & 3) << 2);
    else return c;
}

// O^T accumulators -> bf16 rows with 16-byte stores.  A lane holds, per 32-wide d block, four groups of 4 consecutive
// columns at 8*g4 + 4*h; lanes l and l^32 (h = 0 / 1) exchange one packed group per pair with v_permlane32_swap so
// that each ends up with 8 consecutive columns: four 16-byte stores per lane instead of eight 8-byte ones.
template <int NDB>
__device__ __forceinline__ void store_o_wide(const f32x16 (&acc_o)[NDB], float inv, bf16_t* O, int h) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
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
}

// PRE: q already carries scale*log2(e) (folded into the q RMSNorm+RoPE kernel, one bf16 rounding of the product), no key bias.
// The running max (log2 units) then enters the S^T MFMA chain as its INITIAL ACCUMULATOR (a persistent 16-register
// tuple holding -m, rewritten only when the max moves), so the accumulator comes out as S - m and p = exp2(acc):
// no per-score multiply/subtract and no per-tile accumulator zeroing.
template <int HD, bool PRE>
__global__ __launch_bounds__(256, (HD >= 128 ? (PRE ? 2 : 1) : (HD == 64 ? 3 : 2))) void attn_bf16_kernel(const AttnArgs a) {
    constexpr int HDP = HD < 32 ? 32 : HD;          // padded head dim for the PV d-blocks
    constexpr int KROW = HD * 2, VROW = HDP * 2;    // bytes per LDS row
    constexpr int KCPR = KROW / 16, VCPR = VROW / 16, VCV = (HD * 2) / 16;  // chunks per row; valid V chunks
    constexpr int NKS = HD / 16;                     // k-steps over head dim for S^T
    constexpr int NDB = HDP / 32;                    // 32-row d blocks of O^T
    constexpr int KCH = (BKV * KCPR + 255) / 256, VCH = (BKV * VCV + 255) / 256;
    constexpr int TILE_BYTES = BKV * (KROW + VROW);
    // DMA: K/V tiles go global -> LDS with buffer_load ... lds into a ring of three tiles (the load of tile t+2 is in
    // flight while t is multiplied; no staging registers, no ds_write pass); otherwise register-staged double buffer.
    constexpr bool DMA = PRE && (HD == 64 || HD == 128) && ATTN_DMA;
    constexpr int NBUF = (DMA && HD == 64) ? 3 : 2;            // head_dim 128: 32 KiB tiles, two of them (two blocks per CU)
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * TILE_BYTES];

    if (a.gate_flag && *a.gate_flag != a.gate_ticket) return;     // exact pass behind attn_q128_kernel: only after an overflow in THAT launch
    if (a.gate_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&q128_exact_launches, 1ull);      // diagnostic (ltx_attention_fallback_counts)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid.  Blocks L and L+8 share an XCD (one L2): with heads % 8 == 0 every XCD is given WHOLE heads
    // (head = xcd + 8*(j / nqb)), so the K/V of a head (2 x Sk x hd bf16, 1.3 MB at 4992 x 64) is re-read by its
    // query blocks from that XCD's L2 instead of from the Infinity Cache by all eight.
    int head, b, qb;
    {
        const int nqb = (a.Sq + BQ - 1) / BQ, per_b = nqb * a.heads;
        int L = blockIdx.x;
        b = L / per_b; L -= b * per_b;
        if (a.xcd_heads) { const int xcd = L & 7, j = L >> 3; head = xcd + 8 * (j / nqb); qb = j % nqb; }
        else { head = L / nqb; qb = L - head * nqb; }
    }
    const int q0 = qb * BQ + wave * 32;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + head * HD;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Sk * a.ldk + head * HD;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Sk * a.ldv + head * HD;
    const float* bias = a.bias ? a.bias + (int64_t)b * a.Sk : nullptr;
    const float LOG2E = 1.4426950408889634f;
    const float sc2 = a.scale * LOG2E;

    // Q^T operand fragments (B operand: k = d, col = query)
    bf16x8 qf[NKS];
    {
        int qr = q0 + r; if (qr > a.Sq - 1) qr = a.Sq - 1;
        const bf16_t* qp = Q + (int64_t)qr * a.ldq + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    u32x4 rk[KCH], rv[VCH];
    auto gload = [&](int t) {
        const int kv0 = t * BKV;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            int idx = tid + 256 * i;
            if (idx < BKV * KCPR) {
                int row = idx / KCPR, c = idx % KCPR;
                int key = kv0 + row; if (key > a.Sk - 1) key = a.Sk - 1;
                rk[i] = *reinterpret_cast<const u32x4*>(K + (int64_t)key * a.ldk + c * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            int idx = tid + 256 * i;
            if (idx < BKV * VCV) {
                int row = idx / VCV, c = idx % VCV;
                int key = kv0 + row; if (key > a.Sk - 1) key = a.Sk - 1;
                rv[i] = *reinterpret_cast<const u32x4*>(V + (int64_t)key * a.ldv + c * 8);
            }
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* Ks = smem + buf * TILE_BYTES;
        unsigned char* Vs = Ks + BKV * KROW;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            int idx = tid + 256 * i;
            if (idx < BKV * KCPR) {
                int row = idx / KCPR, c = idx % KCPR;
                *reinterpret_cast<u32x4*>(Ks + row * KROW + kswz<KCPR>(row, c) * 16) = rk[i];
            }
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            int idx = tid + 256 * i;
            if (idx < BKV * VCV) {
                int row = idx / VCV, c = idx % VCV;
                *reinterpret_cast<u32x4*>(Vs + row * VROW + vswz<VCPR>(row, c) * 16) = rv[i];
            }
        }
    };

    f32x16 acc_o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[d][i] = 0.f;
    float m_run = PRE ? 0.f : -INFINITY, l_run = 0.f;
    f32x16 minit;                                   // initial accumulator of the S^T chains: 0, or -m_run (PRE)
#pragma unroll
    for (int i = 0; i < 16; ++i) minit[i] = 0.f;

    const int nt = (a.Sk + BKV - 1) / BKV;
    // ---- DMA staging geometry (HD = 64: 8 pieces of 8 rows x 128 B per K tile and per V tile; wave w issues pieces 2w, 2w+1)
    // lane -> row (lane>>3) of the piece, physical 16-B chunk lane&7; the LDS image is lane-linear, so the bank swizzles
    // of the tile (kswz / vswz, both involutions) are applied to the SOURCE chunk.  Rows past Sk are out of range of the
    // buffer descriptor and read as zeros (masked again by the tail path).
    constexpr int RPP = 1024 / KROW;                           // rows per 1-KiB DMA piece (8 at head_dim 64, 4 at 128)
    constexpr int PW = DMA ? (BKV / RPP) / 4 : 1;              // pieces per wave per K tile (and per V tile)
    constexpr int INFLIGHT = NBUF == 3 ? 2 * PW : 0;           // DMA instructions of the tile that may stay in flight
    __amdgpu_buffer_rsrc_t rk_rsrc, rv_rsrc;
    uint32_t k_voff[PW], v_voff[PW];
    if constexpr (DMA) {
        const uint32_t k_bytes = (uint32_t)(a.Sk - 1) * (uint32_t)a.ldk * 2u + HD * 2u;
        const uint32_t v_bytes = (uint32_t)(a.Sk - 1) * (uint32_t)a.ldv * 2u + HD * 2u;
        rk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, (int)k_bytes, 0x00020000);
        rv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V), 0, (int)v_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int row = (wave * PW + j) * RPP + lane / KCPR, pc = lane % KCPR;
            k_voff[j] = (uint32_t)row * (uint32_t)a.ldk * 2u + (uint32_t)kswz<KCPR>(row, pc) * 16u;
            v_voff[j] = (uint32_t)row * (uint32_t)a.ldv * 2u + (uint32_t)vswz<VCPR>(row, pc) * 16u;
        }
    }
    auto dma_tile = [&](int t, int buf) {
        if constexpr (DMA) {
            unsigned char* Ks = smem + buf * TILE_BYTES;
            unsigned char* Vs = Ks + BKV * KROW;
            const uint32_t ks = (uint32_t)t * BKV * (uint32_t)a.ldk * 2u, vs = (uint32_t)t * BKV * (uint32_t)a.ldv * 2u;
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk_rsrc, (__attribute__((address_space(3))) void*)(Ks + (wave * PW + j) * 1024), 16, (int)k_voff[j], (int)ks, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv_rsrc, (__attribute__((address_space(3))) void*)(Vs + (wave * PW + j) * 1024), 16, (int)v_voff[j], (int)vs, 0, 0);
            }
        }
    };
    if constexpr (DMA) {
        dma_tile(0, 0);
        if (NBUF == 3 && nt > 1) { dma_tile(1, 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        gload(0); swrite(0);
        __syncthreads();
    }

    // tr-read lane geometry (see header): 16-lane group g = lane>>4 -> (h = g>>1, d-half = g&1)
    const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
    // lane part of the V^T read address per d-block (the swizzles depend only on row bits below 8, so the tile row
    // offsets kb*32 + 16s (+8) and the ring slot go into the instruction's immediate)
    uint32_t tr_base[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
        const int row = 4 * h + trq, cv = d * 4 + trdh * 2 + (trp >> 1);
        tr_base[d] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem
                     + row * VROW + vswz<VCPR>(row, cv) * 16 + (trp & 1) * 8;
    }

    // Softmax bookkeeping.  Fast path (no key bias): the running max m_run is kept in RAW score units and
    // the scale is folded into the exponent, p = exp2(fma(s, c, -m*c)) with c = scale*log2(e): one FMA + one
    // v_exp per element, no separate scaling pass.  With a key bias the scores are first moved to the log2
    // domain (x = s*c + bias*log2e) and c becomes 1.  The O/l rescale is LAZY: it is skipped while the tile
    // max exceeds the running max by less than 2^RESCALE_THR for every query of the wave (P then stays <= 32,
    // exact in bf16's floating format; sums are f32).  Tail-key masking runs only on the last, partial tile.
    const bool has_bias = bias != nullptr;
    const float c = has_bias ? 1.0f : sc2;
    constexpr float RESCALE_THR = 5.0f;
    // slot_tag >= 0: the ring slot of tile t is a compile-time constant (the steady-state loop is unrolled by NBUF), so
    // every LDS read is <per-lane base register> + immediate and the DMA target is a scalar constant; -1: computed.
    auto tile_body = [&](int t, auto masked_tag, auto slot_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
        const int buf = SLOT >= 0 ? SLOT : (NBUF == 3 ? t % 3 : (t & 1));
        if constexpr (DMA) { if (t + NBUF - 1 < nt) dma_tile(t + NBUF - 1, SLOT >= 0 ? (SLOT + NBUF - 1) % NBUF : (t + NBUF - 1) % NBUF); }
        else if (t + 1 < nt) gload(t + 1);
        const unsigned char* Ks = smem + buf * TILE_BYTES;
        const unsigned char* Vs = Ks + BKV * KROW;
        const int kv0 = t * BKV;

        // ---- S^T = K . Q^T  (two 32-key blocks); PRE: accumulators start at -m_run
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + r;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * KROW + kswz<KCPR>(row, 2 * ks + h) * 16);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? minit : sacc[kb], 0, 0, 0);
            }
        }
        if ((!PRE && has_bias) || MASKED) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    float x = sacc[kb][i];
                    if (!PRE && has_bias) x = fmaf(x, sc2, bias[key < a.Sk ? key : a.Sk - 1] * LOG2E);
                    if (MASKED && key >= a.Sk) x = -INFINITY;
                    sacc[kb][i] = x;
                }
        }
        // ---- tile max per query column (in-register over 32 scores, then across the two lane halves)
        float mt = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(fmaxf(mt, sacc[0][i]), sacc[1][i]);
        {
            unsigned mu = __float_as_uint(mt);
            auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float ls = 0.f;
        bf16x8 pf[2][2];
        if constexpr (PRE) {
            // mt is the tile max RELATIVE to m_run.  First tile: adopt it unconditionally (m_run was a placeholder 0).
            if (t == 0 || !__all(mt <= RESCALE_THR)) {
                const float delta = t == 0 ? mt : fmaxf(mt, 0.f);
                const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc_o[d][i] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { minit[i] = -m_run; sacc[0][i] -= delta; sacc[1][i] -= delta; }
            }
            f32x2 ls2 = {0.f, 0.f};                                  // two partial sums: v_pk_add_f32
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f32x2 p = {__builtin_amdgcn_exp2f(sacc[kb][i]), __builtin_amdgcn_exp2f(sacc[kb][i + 1])};
                    ls2 += p;
                    pf[kb][i >> 3][i & 7] = (bf16_t)p[0];
                    pf[kb][i >> 3][(i & 7) + 1] = (bf16_t)p[1];
                }
            ls = ls2[0] + ls2[1];
        } else {
            if (!__all((mt - m_run) * c <= RESCALE_THR)) {
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc_o[d][i] *= alpha;
            }
            const float mc = -m_run * c;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], c, mc));
                    ls += p;
                    pf[kb][i >> 3][i & 7] = (bf16_t)p;
                }
        }
        l_run += ls;
        // ---- O^T += V^T . P^T
        if constexpr (DMA && ATTN_TR_ASM) {
            // one MFMA operand = two transpose reads; a rolling window of four operands (16 registers) is in flight:
            // the reads of step n+4 are issued right after MFMA n has taken its operand
            constexpr int NSTEP = NDB * 4;                          // step n = d*4 + kb*2 + s
            u32x2 vr[4][2];
            auto issue = [&](auto n_tag) {
                constexpr int n = decltype(n_tag)::value, d = n >> 2, j = n & 3;
                constexpr int rowc = (j >> 1) * 32 + (j & 1) * 16;
                constexpr int imm = (SLOT >= 0 ? SLOT : 0) * TILE_BYTES + BKV * KROW + rowc * VROW;
                const uint32_t vb = tr_base[d] + (SLOT >= 0 ? 0u : (uint32_t)buf * TILE_BYTES);
                vr[j][0] = ds_tr_read<imm>(vb);
                vr[j][1] = ds_tr_read<imm + 8 * VROW>(vb);
            };
            static_for<0, 4>([&](auto n_tag) { issue(n_tag); });
            static_for<0, NSTEP>([&](auto n_tag) {
                constexpr int n = decltype(n_tag)::value, d = n >> 2, j = n & 3;
                constexpr int after = (NSTEP - 1 - n < 3 ? NSTEP - 1 - n : 3) * 2;
                lds_wait<after>(vr[j][0], vr[j][1]);
                union { u32x2 u[2]; bf16x8 v; } cvt;
                cvt.u[0] = vr[j][0]; cvt.u[1] = vr[j][1];
                acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt.v, pf[j >> 1][j & 1], acc_o[d], 0, 0, 0);
                if constexpr (n + 4 < NSTEP) issue(std::integral_constant<int, n + 4>{});
            });
        } else {
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const int cv = d * 4 + trdh * 2 + (trp >> 1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int row0 = kb * 32 + 16 * s + 4 * h + trq;
                    const int row1 = row0 + 8;
                    bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(Vs + row0 * VROW + vswz<VCPR>(row0, cv) * 16 + (trp & 1) * 8));
                    bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(Vs + row1 * VROW + vswz<VCPR>(row1, cv) * 16 + (trp & 1) * 8));
                    bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], acc_o[d], 0, 0, 0);
                }
        }
        }
        if constexpr (DMA) {
            // tile t+1 must have landed for every wave; with the 3-tile ring tile t+2's pieces may stay in flight
            if (NBUF == 3 && t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            if (t + 1 < nt) swrite(buf ^ 1);
            __syncthreads();
        }
    };
    const bool ragged = (a.Sk % BKV) != 0;
    using dyn_slot = std::integral_constant<int, -1>;
    int t = 0;
    if constexpr (DMA && ATTN_STATIC_SLOTS) {
        for (; t + NBUF <= nt - 1; t += NBUF) {
            tile_body(t, std::false_type{}, std::integral_constant<int, 0>{});
            tile_body(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
            if constexpr (NBUF == 3) tile_body(t + 2, std::false_type{}, std::integral_constant<int, 2>{});
        }
    }
    for (; t < nt - 1; ++t) tile_body(t, std::false_type{}, dyn_slot{});
    if (ragged) tile_body(nt - 1, std::true_type{}, dyn_slot{}); else tile_body(nt - 1, std::false_type{}, dyn_slot{});

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + r;
    if (qr < a.Sq) {
        bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + ((int64_t)b * a.Sq + qr) * a.ldo + head * HD;
        if (HD % 32 == 0 && a.wide_o) { store_o_wide<NDB>(acc_o, inv, O, h); return; }
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;
                if (dd < HD) {
                    bf16x4 o4 = {(bf16_t)(acc_o[d][4 * g4 + 0] * inv), (bf16_t)(acc_o[d][4 * g4 + 1] * inv),
                                 (bf16_t)(acc_o[d][4 * g4 + 2] * inv), (bf16_t)(acc_o[d][4 * g4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(O + dd) = o4;
                }
            }
    }
}

// ---- head_dim 64, q prescaled, no bias: two-tile software pipeline inside one wave --------------------------------
// The kernel above runs a tile's phases back to back in each wave (S^T MFMAs -> max -> exp -> PV MFMAs) and leaves the
// overlap of matrix and vector work to the three co-resident waves of a SIMD, which overlap poorly (one block per CU
// already reaches 2/3 of the rate of three).  Here S^T of tile t+1 is issued BEFORE the softmax of tile t, so a single
// wave's instruction stream carries independent matrix work beside its exp/max/convert work; the two S^T accumulator
// sets alternate (loop unrolled by four = ring slots, all LDS addresses immediates).  K/V ring: four 16-KiB tiles,
// tile t+3 in flight while tile t is multiplied; 64 KiB LDS and <= 256 registers -> two blocks per CU.
// S^T(t+1) is formed against the running max as it stood BEFORE tile t's update; when tile t moves the max, the
// pending accumulators are shifted with it.
#ifndef ATTN_ABL
#define ATTN_ABL 0      // timing ablations (wrong results): 1 no K/V loads in the loop, 2 no exp, 3 no barrier
#endif
#ifndef ATTN_ORDER
#define ATTN_ORDER 0    // where tile t+3's LDS-DMA is issued: 0 top of the body, 1 after the S^T MFMAs, 2 after the exps
#endif
#ifndef ATTN_VEARLY
#define ATTN_VEARLY 0   // 1: first V^T operand reads issued ahead of the softmax
#endif
#ifndef ATTN_PIPE_OCC
#define ATTN_PIPE_OCC 2
#endif
__global__ __launch_bounds__(256, ATTN_PIPE_OCC) void attn_pipe64_kernel(const AttnArgs a) {
    constexpr int HD = 64, KROW = 128, VROW = 128, KCPR = 8, VCPR = 8, NKS = 4, NDB = 2;
    constexpr int TILE_BYTES = BKV * (KROW + VROW), NSLOT = 4;
    constexpr int RPP = 8, PW = 2, PIECES = 2 * PW;              // DMA pieces (1 KiB) per wave and tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int head, b, qb;
    {
        const int nqb = (a.Sq + BQ - 1) / BQ, per_b = nqb * a.heads;
        int L = blockIdx.x;
        b = L / per_b; L -= b * per_b;
        if (a.xcd_heads) { const int xcd = L & 7, j = L >> 3; head = xcd + 8 * (j / nqb); qb = j % nqb; }
        else { head = L / nqb; qb = L - head * nqb; }
    }
    const int q0 = qb * BQ + wave * 32;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + head * HD;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Sk * a.ldk + head * HD;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Sk * a.ldv + head * HD;

    bf16x8 qf[NKS];
    {
        int qr = q0 + r; if (qr > a.Sq - 1) qr = a.Sq - 1;
        const bf16_t* qp = Q + (int64_t)qr * a.ldq + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }
    const int nt = (a.Sk + BKV - 1) / BKV;
    __amdgpu_buffer_rsrc_t rk_rsrc, rv_rsrc;
    uint32_t k_voff[PW], v_voff[PW];
    {
        const uint32_t k_bytes = (uint32_t)(a.Sk - 1) * (uint32_t)a.ldk * 2u + HD * 2u;
        const uint32_t v_bytes = (uint32_t)(a.Sk - 1) * (uint32_t)a.ldv * 2u + HD * 2u;
        rk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, (int)k_bytes, 0x00020000);
        rv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V), 0, (int)v_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int row = (wave * PW + j) * RPP + lane / KCPR, pc = lane % KCPR;
            k_voff[j] = (uint32_t)row * (uint32_t)a.ldk * 2u + (uint32_t)kswz<KCPR>(row, pc) * 16u;
            v_voff[j] = (uint32_t)row * (uint32_t)a.ldv * 2u + (uint32_t)vswz<VCPR>(row, pc) * 16u;
        }
    }
    auto dma_tile = [&](int t, int slot) {
        unsigned char* Ks = smem + slot * TILE_BYTES;
        unsigned char* Vs = Ks + BKV * KROW;
        const uint32_t ks = (uint32_t)t * BKV * (uint32_t)a.ldk * 2u, vs = (uint32_t)t * BKV * (uint32_t)a.ldv * 2u;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk_rsrc, (__attribute__((address_space(3))) void*)(Ks + (wave * PW + j) * 1024), 16, (int)k_voff[j], (int)ks, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv_rsrc, (__attribute__((address_space(3))) void*)(Vs + (wave * PW + j) * 1024), 16, (int)v_voff[j], (int)vs, 0, 0);
        }
    };

    f32x16 acc_o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[d][i] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    f32x16 minit;
#pragma unroll
    for (int i = 0; i < 16; ++i) minit[i] = 0.f;

    // lane parts of the LDS addresses (tile row offsets and the ring slot are instruction immediates)
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t k_base[NKS], tr_base[NDB];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k_base[ks] = smem_base + r * KROW + kswz<KCPR>(r, 2 * ks + h) * 16;   // rows r and r+32 swizzle alike
    {
        const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const int row = 4 * h + trq, cv = d * 4 + trdh * 2 + (trp >> 1);
            tr_base[d] = smem_base + row * VROW + vswz<VCPR>(row, cv) * 16 + (trp & 1) * 8;
        }
    }

    // S^T of one tile: two chains of four MFMAs, accumulators start at -m_run
    auto qk_tile = [&](auto slot_tag, int slot_dyn, f32x16 (&sacc)[2]) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const uint32_t off = SLOT >= 0 ? 0u : (uint32_t)slot_dyn * TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(__attribute__((address_space(3))) const bf16x8*)(uintptr_t)(
                    k_base[ks] + off + (SLOT >= 0 ? SLOT : 0) * TILE_BYTES + kb * 32 * KROW);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? minit : sacc[kb], 0, 0, 0);
            }
    };

    constexpr float RESCALE_THR = 5.0f;
    // body of tile t: cur holds S^T(t) - m; nxt receives S^T(t+1)
    auto body = [&](int t, auto slot_tag, auto last_tag, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;               // last tile: nothing to prefetch, tail keys masked
        const int slot = SLOT >= 0 ? SLOT : (t & 3);
        auto prefetch = [&] {
#if ATTN_ABL != 1
            if (t + 3 < nt) dma_tile(t + 3, SLOT >= 0 ? (SLOT + 3) & 3 : (t + 3) & 3);
#endif
        };
        // V^T operands: a rolling window of four MFMA operands (16 registers), first four issued ahead of the softmax
        constexpr int NSTEP = NDB * 4;
        u32x2 vr[4][2];
        auto issue = [&](auto n_tag) {
            constexpr int n = decltype(n_tag)::value, d = n >> 2, j = n & 3;
            constexpr int rowc = (j >> 1) * 32 + (j & 1) * 16;
            constexpr int imm = (SLOT >= 0 ? SLOT : 0) * TILE_BYTES + BKV * KROW + rowc * VROW;
            const uint32_t vb = tr_base[d] + (SLOT >= 0 ? 0u : (uint32_t)slot * TILE_BYTES);
            vr[j][0] = ds_tr_read<imm>(vb);
            vr[j][1] = ds_tr_read<imm + 8 * VROW>(vb);
        };
        if (ATTN_ORDER == 0) prefetch();
        if constexpr (!LAST) {
            if constexpr (SLOT >= 0) qk_tile(std::integral_constant<int, (SLOT + 1) & 3>{}, 0, nxt);
            else qk_tile(std::integral_constant<int, -1>{}, (t + 1) & 3, nxt);
        }
        if (ATTN_ORDER == 1) prefetch();
        if (ATTN_VEARLY) static_for<0, 4>([&](auto n_tag) { issue(n_tag); });
        if constexpr (LAST) {
            const int kv0 = t * BKV;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= a.Sk) cur[kb][i] = -INFINITY;
                }
        }
        float mt = fmaxf(cur[0][0], cur[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(fmaxf(mt, cur[0][i]), cur[1][i]);
        {
            unsigned mu = __float_as_uint(mt);
            auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (t == 0 || !__all(mt <= RESCALE_THR)) {
            const float delta = t == 0 ? mt : fmaxf(mt, 0.f);
            const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_o[d][i] *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                minit[i] = -m_run; cur[0][i] -= delta; cur[1][i] -= delta;
                if constexpr (!LAST) { nxt[0][i] -= delta; nxt[1][i] -= delta; }
            }
        }
        bf16x8 pf[2][2];
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
#if ATTN_ABL == 2
                f32x2 p = {cur[kb][i], cur[kb][i + 1]};
#else
                f32x2 p = {__builtin_amdgcn_exp2f(cur[kb][i]), __builtin_amdgcn_exp2f(cur[kb][i + 1])};
#endif
                ls2 += p;
                pf[kb][i >> 3][i & 7] = (bf16_t)p[0];
                pf[kb][i >> 3][(i & 7) + 1] = (bf16_t)p[1];
            }
        l_run += ls2[0] + ls2[1];
        if (ATTN_ORDER == 2) prefetch();
        if (!ATTN_VEARLY) static_for<0, 4>([&](auto n_tag) { issue(n_tag); });
        static_for<0, NSTEP>([&](auto n_tag) {
            constexpr int n = decltype(n_tag)::value, d = n >> 2, j = n & 3;
            constexpr int after = (NSTEP - 1 - n < 3 ? NSTEP - 1 - n : 3) * 2;
            lds_wait<after>(vr[j][0], vr[j][1]);
            union { u32x2 u[2]; bf16x8 v; } cvt;
            cvt.u[0] = vr[j][0]; cvt.u[1] = vr[j][1];
            acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt.v, pf[j >> 1][j & 1], acc_o[d], 0, 0, 0);
            if constexpr (n + 4 < NSTEP) issue(std::integral_constant<int, n + 4>{});
        });
        // tile t+2 (its K is read by the next body) must have landed for every wave; tile t+3's pieces stay in flight
#if ATTN_ABL != 1
        if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if ATTN_ABL != 3
        __builtin_amdgcn_s_barrier();
#endif
    };

    dma_tile(0, 0);
    if (nt > 1) dma_tile(1, 1);
    if (nt > 2) { dma_tile(2, 2); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x16 sA[2], sB[2];
    qk_tile(std::integral_constant<int, 0>{}, 0, sA);
    using dyn = std::integral_constant<int, -1>;
    int t = 0;
    for (; t + 4 <= nt - 1; t += 4) {
        body(t, std::integral_constant<int, 0>{}, std::false_type{}, sA, sB);
        body(t + 1, std::integral_constant<int, 1>{}, std::false_type{}, sB, sA);
        body(t + 2, std::integral_constant<int, 2>{}, std::false_type{}, sA, sB);
        body(t + 3, std::integral_constant<int, 3>{}, std::false_type{}, sB, sA);
    }
    for (; t + 2 <= nt - 1; t += 2) {
        body(t, dyn{}, std::false_type{}, sA, sB);
        body(t + 1, dyn{}, std::false_type{}, sB, sA);
    }
    if (t < nt - 1) { body(t, dyn{}, std::false_type{}, sA, sB); body(t + 1, dyn{}, std::true_type{}, sB, sA); }
    else body(t, dyn{}, std::true_type{}, sA, sB);

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + r;
    if (qr < a.Sq) {
        bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + ((int64_t)b * a.Sq + qr) * a.ldo + head * HD;
        if (a.wide_o) { store_o_wide<NDB>(acc_o, inv, O, h); return; }
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;
                bf16x4 o4 = {(bf16_t)(acc_o[d][4 * g4 + 0] * inv), (bf16_t)(acc_o[d][4 * g4 + 1] * inv),
                             (bf16_t)(acc_o[d][4 * g4 + 2] * inv), (bf16_t)(acc_o[d][4 * g4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(O + dd) = o4;
            }
    }
}

// ---- head_dim 64, at most 128 keys (the DiT's cross-attention over the text tokens, ltx_transformer.rs:719-740 with the
// additive key mask): K and V of a head sit in LDS once per block, every wave then runs whole 32-query units on its own -
// one-shot softmax over all keys (no running max, no rescale), no barrier after the initial load, next unit's Q
// fragments in flight while the current unit is multiplied.  Keys past Sk read as zeros (buffer range) and get -inf.
constexpr int XKV = 128;                                 // key capacity of the cross kernel
#ifdef XTRACE
__device__ unsigned g_xtrace[4096 * 8];
extern "C" int ltx_dbg_xtrace(unsigned* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xtrace), (size_t)n * 4) == hipSuccess ? 0 : -1; }
#define XSTAMP(i) if (lane == 0 && blockIdx.x < 1024) g_xtrace[(blockIdx.x * 4 + wave) * 8 + (i)] = (unsigned)wall_clock64()
#else
#define XSTAMP(i)
#endif
#ifndef XATTN_WPS
#define XATTN_WPS 3          // waves per SIMD the kernel is compiled for (blocks per CU): 2 = round 3's build (254 registers)
#endif
// The units of one block (after K/V and the key bias sit in LDS), for a key set of NKB blocks of 32 keys: everything that scales
// with the number of keys (S MFMAs, exps, P V MFMAs, bias registers) is sized by NKB, chosen per BLOCK from the number of keys
// its batch row really has (AttnArgs::k_count, or Sk): BASELINE's prompts keep 32 of 128 text tokens.
template <bool B2D, int NKB>
__device__ __forceinline__ void cross64_units(const AttnArgs& a, int groups, const unsigned char* smem, const float* sbias, int b, int head, int grp) {
    constexpr int HD = 64, KROW = 128, VROW = 128, KCPR = 8, VCPR = 8, NKS = 4, NDB = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + head * HD;
    const float LOG2E = 1.4426950408889634f;
    const float c = a.q_prescaled ? 1.0f : a.scale * LOG2E;
    // the key bias stays in LDS and is read in accumulator order (key = kb*32 + (i&3) + 8*(i>>2) + 4h) where the scores are
    // scaled: 64 registers less than keeping the tuples (round 4: three blocks per CU instead of two)
    auto bias4 = [&](int kb, int g4) { return *reinterpret_cast<const f32x4*>(&sbias[kb * 32 + 8 * g4 + 4 * h]); };
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem;
    uint32_t k_base[NKS], tr_base[NDB];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k_base[ks] = smem_base + r * KROW + kswz<KCPR>(r, 2 * ks + h) * 16;
    {
        const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const int row = 4 * h + trq, cv = d * 4 + trdh * 2 + (trp >> 1);
            tr_base[d] = smem_base + XKV * KROW + row * VROW + vswz<VCPR>(row, cv) * 16 + (trp & 1) * 8;
        }
    }
    const int nunits = (a.Sq + 31) / 32, ustride = groups * 4;
    // AttnArgs::q_rowsq: queries arrive un-normalised; the RMS-norm scalar of a query row multiplies its score row, i.e. it
    // joins the factor c below (a lane owns one query column), and costs nothing per score.  The partial sums of squares come
    // from the projection's epilogue (GemmArgs::rowsq), q_rowsq_n per row (a multiple of 4), summed in ascending order.
    const float* RS = a.q_rowsq ? a.q_rowsq + (int64_t)b * a.Sq * a.q_rowsq_n : nullptr;
    const float rs_invD = a.q_rowsq ? 1.0f / (float)a.q_rowsq_D : 0.f;
    // (the partials are only LOADED here - up to 16 per row, launcher-checked - so that the prefetch of the next unit stays
    // asynchronous; row_factor() folds them where the unit's scores are scaled)
    auto load_q = [&](int u, bf16x8 (&qf)[NKS], f32x4 (&rs)[4]) {
        int qr = u * 32 + r; if (qr > a.Sq - 1) qr = a.Sq - 1;
        const bf16_t* qp = Q + (int64_t)qr * a.ldq + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
        if (RS) {
            const float* rp = RS + (int64_t)qr * a.q_rowsq_n;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                rs[g4] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (4 * g4 < a.q_rowsq_n) rs[g4] = *reinterpret_cast<const f32x4*>(rp + 4 * g4);
            }
        }
    };
    auto row_factor = [&](const f32x4 (&rs)[4]) {
        if (!RS) return c;
        float ss = 0.f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { ss += rs[g4][0]; ss += rs[g4][1]; ss += rs[g4][2]; ss += rs[g4][3]; }
        return c * __builtin_amdgcn_rsqf(ss * rs_invD + a.q_rowsq_eps);     // one v_rsq (1 ulp): this is the bf16 path
    };
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    // UB units per iteration of a wave.  With one key block a unit is 4 + 4 MFMAs and 16 exps behind a global load and in front of
    // a store: the wave's time is memory latency, so it carries two units at once (their loads, and the next two units' loads, in
    // flight together); with more key blocks the registers go to the scores instead.
    constexpr int UB = (NKB == 1 && !B2D) ? 2 : 1;
    int u = grp * 4 + wave;
    bf16x8 qf[UB][NKS], qn[UB][NKS];
    f32x4 rsq[UB][4], rsn[UB][4];
    XSTAMP(1);
#pragma unroll
    for (int j = 0; j < UB; ++j) if (u + j * ustride < nunits) load_q(u + j * ustride, qf[j], rsq[j]);
    int xi = 2;
    for (; u < nunits; u += UB * ustride) {
        if (xi < 8) { XSTAMP(xi); ++xi; }
#pragma unroll
        for (int j = 0; j < UB; ++j) if (u + (UB + j) * ustride < nunits) load_q(u + (UB + j) * ustride, qn[j], rsn[j]);
#pragma unroll
        for (int j = 0; j < UB; ++j) {
        const int uj = u + j * ustride;
        if (uj >= nunits) break;                             // wave-uniform
        // (B2D) this lane's query row of the [heads, Sq, Sk] bias, in accumulator order, in flight under the S MFMAs
        f32x4 b2[B2D ? NKB : 1][B2D ? 4 : 1];
        if constexpr (B2D) {
            int qr = uj * 32 + r; if (qr > a.Sq - 1) qr = a.Sq - 1;
            const float* brow = a.bias2d + ((int64_t)head * a.Sq + qr) * a.Sk;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int kk = kb * 32 + 8 * g4 + 4 * h;
                    b2[kb][g4] = kk < a.Sk ? *reinterpret_cast<const f32x4*>(brow + kk) : (f32x4){0.f, 0.f, 0.f, 0.f};      // Sk % 4 == 0: a group lies inside or outside
                }
        }
        // S^T = K . Q^T for all keys
        f32x16 sacc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(__attribute__((address_space(3))) const bf16x8*)(uintptr_t)(k_base[ks] + kb * 32 * KROW);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[j][ks], ks == 0 ? zero16 : sacc[kb], 0, 0, 0);
            }
        float mt = -INFINITY;
        const float cq = row_factor(rsq[j]);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 bq = bias4(kb, g4);
                if constexpr (B2D) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bq[i] = fmaf(b2[kb][g4][i], LOG2E, bq[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { sacc[kb][4 * g4 + i] = fmaf(sacc[kb][4 * g4 + i], cq, bq[i]); mt = fmaxf(mt, sacc[kb][4 * g4 + i]); }
            }
        {
            unsigned mu = __float_as_uint(mt);
            auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        bf16x8 pf[NKB][2];
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 p = {__builtin_amdgcn_exp2f(sacc[kb][i] - mt), __builtin_amdgcn_exp2f(sacc[kb][i + 1] - mt)};
                ls2 += p;
                pf[kb][i >> 3][i & 7] = (bf16_t)p[0];
                pf[kb][i >> 3][(i & 7) + 1] = (bf16_t)p[1];
            }
        float l = ls2[0] + ls2[1];
        l += __shfl_xor(l, 32);
        // O^T = V^T . P^T
        f32x16 acc_o[NDB];
        constexpr int NSTEP = NDB * NKB * 2;                 // step n = d*(2*NKB) + kb*2 + s
        constexpr int RING = NSTEP < 4 ? NSTEP : 4;
        u32x2 vr[4][2];
        auto issue = [&](auto n_tag) {
            constexpr int n = decltype(n_tag)::value, d = n / (2 * NKB), jj = n % (2 * NKB);
            constexpr int imm = ((jj >> 1) * 32 + (jj & 1) * 16) * VROW;
            vr[n & 3][0] = ds_tr_read<imm>(tr_base[d]);
            vr[n & 3][1] = ds_tr_read<imm + 8 * VROW>(tr_base[d]);
        };
        static_for<0, RING>([&](auto n_tag) { issue(n_tag); });
        static_for<0, NSTEP>([&](auto n_tag) {
            constexpr int n = decltype(n_tag)::value, d = n / (2 * NKB), jj = n % (2 * NKB);
            constexpr int after = (NSTEP - 1 - n < 3 ? NSTEP - 1 - n : 3) * 2;
            lds_wait<after>(vr[n & 3][0], vr[n & 3][1]);
            union { u32x2 u2[2]; bf16x8 v; } cvt;
            cvt.u2[0] = vr[n & 3][0]; cvt.u2[1] = vr[n & 3][1];
            acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt.v, pf[jj >> 1][jj & 1], jj == 0 ? zero16 : acc_o[d], 0, 0, 0);
            if constexpr (n + 4 < NSTEP) issue(std::integral_constant<int, n + 4>{});
        });
        const float inv = 1.0f / l;
        const int qr = uj * 32 + r;
        if (qr < a.Sq) {
            bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + ((int64_t)b * a.Sq + qr) * a.ldo + head * HD;
            if (a.wide_o) store_o_wide<NDB>(acc_o, inv, O, h);
            else
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = d * 32 + 8 * g4 + 4 * h;
                    bf16x4 o4 = {(bf16_t)(acc_o[d][4 * g4 + 0] * inv), (bf16_t)(acc_o[d][4 * g4 + 1] * inv),
                                 (bf16_t)(acc_o[d][4 * g4 + 2] * inv), (bf16_t)(acc_o[d][4 * g4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(O + dd) = o4;
                }
        }
        }
#pragma unroll
        for (int j = 0; j < UB; ++j) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[j][ks] = qn[j][ks];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) rsq[j][g4] = rsn[j][g4];
        }
    }
#ifdef XTRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (xi < 8) XSTAMP(xi);
#endif
}

template <bool B2D>          // B2D: a [heads, Sq, Sk] bias on top of the key bias (T5 self-attention over <= 128 tokens)
__global__ __launch_bounds__(256, B2D ? 2 : XATTN_WPS) void attn_cross64_kernel(const AttnArgs a, int groups) {
    constexpr int HD = 64, KROW = 128, VROW = 128, KCPR = 8, VCPR = 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[XKV * (KROW + VROW)];
    __shared__ __attribute__((aligned(16))) float sbias[XKV];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x % groups, bh = blockIdx.x / groups;
    const int head = bh % a.heads, b = bh / a.heads;
    XSTAMP(0);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Sk * a.ldk + head * HD;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Sk * a.ldv + head * HD;
    const float LOG2E = 1.4426950408889634f;
    // keys this batch row really has: the first k_count[b] rows of its K / V / bias (AttnArgs::k_count), or all Sk
    int nkeys = a.Sk;
    if (a.k_count) { nkeys = __builtin_amdgcn_readfirstlane(a.k_count[b]); if (nkeys > a.Sk) nkeys = a.Sk; if (nkeys < 1) nkeys = 1; }
    const int nkb = (nkeys + 31) >> 5;

    // K/V -> LDS by buffer LDS-DMA: 16 pieces of 8 rows per matrix, 4 + 4 per wave; only the key blocks that are multiplied
    // (rows of a multiplied block past nkeys must hold finite values: the caller's buffers, or zeros past Sk by buffer range)
    {
        const uint32_t k_bytes = (uint32_t)(a.Sk - 1) * (uint32_t)a.ldk * 2u + HD * 2u;
        const uint32_t v_bytes = (uint32_t)(a.Sk - 1) * (uint32_t)a.ldv * 2u + HD * 2u;
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, (int)k_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V), 0, (int)v_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = wave * 4 + j, row = piece * 8 + lane / KCPR, pc = lane % KCPR;
            if (piece * 8 >= nkb * 32) continue;            // wave-uniform
            const uint32_t ko = (uint32_t)row * (uint32_t)a.ldk * 2u + (uint32_t)kswz<KCPR>(row, pc) * 16u;
            const uint32_t vo = (uint32_t)row * (uint32_t)a.ldv * 2u + (uint32_t)vswz<VCPR>(row, pc) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, (int)ko, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(smem + XKV * KROW + piece * 1024), 16, (int)vo, 0, 0, 0);
        }
        if (tid < XKV) {
            float bv = tid < nkeys ? 0.f : -INFINITY;
            if (a.bias && tid < nkeys) bv = a.bias[(int64_t)b * a.Sk + tid] * LOG2E;
            sbias[tid] = bv;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    switch (nkb) {                                          // block-uniform
        case 1: cross64_units<B2D, 1>(a, groups, smem, sbias, b, head, grp); break;
        case 2: cross64_units<B2D, 2>(a, groups, smem, sbias, b, head, grp); break;
        case 3: cross64_units<B2D, 3>(a, groups, smem, sbias, b, head, grp); break;
        default: cross64_units<B2D, 4>(a, groups, smem, sbias, b, head, grp); break;
    }
}

// ---- exact-f32 flash attention (parity mode): one query per lane, 64 queries per block ----
template <int HD>
__global__ __launch_bounds__(64) void attn_f32_kernel(const AttnArgs a) {
    constexpr int TK = 32;
    __shared__ __attribute__((aligned(16))) float Ks[TK][HD];
    __shared__ __attribute__((aligned(16))) float Vs[TK][HD];
    const int lane = threadIdx.x;
    const int head = blockIdx.y, b = blockIdx.z;
    int qr = blockIdx.x * 64 + lane;
    const bool active = qr < a.Sq;
    if (!active) qr = a.Sq - 1;
    const float* Q = reinterpret_cast<const float*>(a.q) + ((int64_t)b * a.Sq + qr) * a.ldq + head * HD;
    const float* K = reinterpret_cast<const float*>(a.k) + (int64_t)b * a.Sk * a.ldk + head * HD;
    const float* V = reinterpret_cast<const float*>(a.v) + (int64_t)b * a.Sk * a.ldv + head * HD;
    const float* bias = a.bias ? a.bias + (int64_t)b * a.Sk : nullptr;
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Q + d);
        q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
        o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < a.Sk; k0 += TK) {
        __syncthreads();
        for (int i = lane; i < TK * HD / 4; i += 64) {
            int row = i / (HD / 4), c = i % (HD / 4);
            int key = k0 + row; if (key > a.Sk - 1) key = a.Sk - 1;
            *reinterpret_cast<f32x4*>(&Ks[row][c * 4]) = *reinterpret_cast<const f32x4*>(K + (int64_t)key * a.ldk + c * 4);
            *reinterpret_cast<f32x4*>(&Vs[row][c * 4]) = *reinterpret_cast<const f32x4*>(V + (int64_t)key * a.ldv + c * 4);
        }
        __syncthreads();
        const int jmax = a.Sk - k0 < TK ? a.Sk - k0 : TK;
        for (int j = 0; j < jmax; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                f32x4 kv = *reinterpret_cast<const f32x4*>(&Ks[j][d]);
                acc += q[d] * kv[0] + q[d + 1] * kv[1] + q[d + 2] * kv[2] + q[d + 3] * kv[3];
            }
            float x = acc * a.scale;
            if (bias) x += bias[k0 + j];
            const float mn = fmaxf(m, x);
            const float alpha = expf(m - mn);
            const float p = expf(x - mn);
            m = mn; l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[j][d]);
                o[d] = o[d] * alpha + p * vv[0]; o[d + 1] = o[d + 1] * alpha + p * vv[1];
                o[d + 2] = o[d + 2] * alpha + p * vv[2]; o[d + 3] = o[d + 3] * alpha + p * vv[3];
            }
        }
    }
    if (active) {
        float* O = reinterpret_cast<float*>(a.o) + ((int64_t)b * a.Sq + qr) * a.ldo + head * HD;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            f32x4 v = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
            *reinterpret_cast<f32x4*>(O + d) = v;
        }
    }
}

}  // namespace

// option attn_off (A/B aid): "cross" = the generic tiled kernel for short key sets too; "q64" = 32-query waves, two blocks per CU
// (attn_pipe64_kernel); "pipe" = one tile at a time per wave
static bool attn_cross_enabled() { return !(ltx_opt().attn_off & LTX_ATTN_CROSS); }
static bool attn_q64_enabled() { return !(ltx_opt().attn_off & LTX_ATTN_Q64); }
static bool attn_pipe_enabled() { return !(ltx_opt().attn_off & LTX_ATTN_PIPE); }

// Shape-only: whether cross attention may take its queries un-normalised (AttnArgs::q_rowsq; bf16, head_dim 64, a key set
// that fits attn_cross64_kernel, at most 16 partials per row).  LTX_Q2_FOLD=0: the stand-alone q-norm pass (A/B aid).
bool ltx_attention_rowsq_ok(int hd, int Sk, int D) {
    return hd == 64 && Sk <= XKV && attn_cross_enabled() && D % 512 == 0 && D / 128 <= 16 && ltx_opt().q2_fold != 0;
}

// Shape-only: whether the short-key-set kernel (attn_cross64_kernel: bf16, head_dim 64, at most 128 keys) serves a launch; what
// AttnArgs::k_count / bias2d / q_rowsq need
bool ltx_attention_cross64_ok(int hd, int Sk) { return hd == 64 && Sk <= XKV && attn_cross_enabled(); }

bool ltx_attention_prescale_ok(int hd) {
    return (hd == 64 || hd == 128) && ltx_exp("attn_prescale", 1);      // (experiment builds: 0 = keep the per-score scale multiply)
}

int ltx_launch_attention(const AttnArgs& a, int dtype, hipStream_t s) {
    if (a.Sq <= 0 || a.Sk <= 0 || a.heads <= 0) LTX_FAIL(LTX_ERR_ARG, "attention: empty problem");
    void* tok = nullptr;
    ltx_prof_begin(a.bias || a.Sq != a.Sk ? LTX_PROF_ATTN_CROSS : LTX_PROF_ATTN_SELF, 4.0 * a.B * a.heads * (double)a.Sq * a.Sk * a.hd, s, &tok);
    struct End { void* t; hipStream_t s; ~End() { ltx_prof_end(t, s); } } end_{tok, s};
    if (dtype == LTX_DT_BF16) {
        if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 4) LTX_FAIL(LTX_ERR_ARG, "attention: strides must be 16-byte aligned");
        dim3 grid((unsigned)(cdiv(a.Sq, BQ) * a.heads * a.B)), block(256);
        AttnArgs ax = a;
        ax.xcd_heads = (a.heads % 8 == 0 && ltx_exp("attn_xcd", 1)) ? 1 : 0;                 // (experiment builds: 0 = plain head-major block order)
        ax.wide_o = (a.ldo % 8 == 0 && ((uintptr_t)a.o & 15) == 0 && a.hd % 8 == 0 && ltx_exp("attn_wide_o", 1)) ? 1 : 0;   // (0 = 8-byte output stores)
        if (a.q_prescaled && (a.bias || !ltx_attention_prescale_ok(a.hd))) LTX_FAIL(LTX_ERR_ARG, "attention: q_prescaled needs head_dim 64 or 128 and no key bias");
        if (a.q_rowsq && !(a.hd == 64 && a.Sk <= XKV && attn_cross_enabled() && a.q_rowsq_n >= 4 && a.q_rowsq_n <= 16 && a.q_rowsq_n % 4 == 0 && a.q_rowsq_D > 0))
            LTX_FAIL(LTX_ERR_ARG, "attention: q_rowsq is served by the short-key-set head_dim-64 kernel only (4..16 partials per row, a multiple of 4)");
        if (a.bias2d && !(a.hd == 64 && a.Sk <= XKV && attn_cross_enabled())) LTX_FAIL(LTX_ERR_ARG, "attention: bias2d is served by the short-key-set head_dim-64 kernel only");
        if (a.k_count && !ltx_attention_cross64_ok(a.hd, a.Sk)) LTX_FAIL(LTX_ERR_ARG, "attention: k_count is served by the short-key-set head_dim-64 kernel only");
        if (a.hd == 64 && a.Sk <= XKV && attn_cross_enabled()) {
            // few keys (text tokens): K/V resident in LDS, one-shot softmax.  Blocks: (batch, head) x groups, sized for ~2 per CU
            const int nunits = cdiv(a.Sq, 32);
            // Blocks: (batch, head) x groups; a block's four waves step through 32-query units.  The kernel is a latency chain per
            // unit (q load -> S -> max -> 64 exps -> PV -> store), so what matters is how many units a wave walks: as many groups as
            // keep every block resident at once (XATTN_WPS blocks per CU since round 4: the key bias moved from 64 registers to
            // LDS), and among those the MOST groups that reach the smallest units-per-wave.  Measured in the pipeline (S = 4992, 32
            // heads, kernel durations): 13 groups 18.9 us, 16: 19.4, 20: 18.6, 24: 17.4; round 3's build (two blocks per CU, 13
            // groups) 18.6 - the kernel is not bound by the length of a wave's unit chain (docs/lab_notes.md R4.6).
            int groups = 1;
            {
                const int slots = 256 * XATTN_WPS;
                int gmax = slots / (a.B * a.heads); if (gmax < 1) gmax = 1;
                int best_upw = 1 << 30;
                for (int gq = 1; gq <= gmax; ++gq) {
                    const int upw = cdiv(cdiv(nunits, gq), 4);
                    if (upw <= best_upw) { best_upw = upw; groups = gq; }
                }
            }
            { const int xg = ltx_exp("attn_cross_groups", 0); if (xg > 0) groups = xg; }   // tuning aid (experiment builds)
            if (groups > cdiv(nunits, 4)) groups = cdiv(nunits, 4); if (groups < 1) groups = 1;
            if (a.bias2d) {
                if (a.Sk % 4 || ((uintptr_t)a.bias2d & 15)) LTX_FAIL(LTX_ERR_ARG, "attention: bias2d needs Sk % 4 == 0 and a 16-byte aligned table");
                LTX_LAUNCH_TIMED(attn_cross64_kernel<true>, dim3((unsigned)(a.B * a.heads * groups)), block, 0, s, ax, groups);
            } else
            LTX_LAUNCH_TIMED(attn_cross64_kernel<false>, dim3((unsigned)(a.B * a.heads * groups)), block, 0, s, ax, groups);
            LTX_CHECK_LAUNCH();
            return LTX_OK;
        }
        switch (a.hd) {
            case 16: hipLaunchKernelGGL((attn_bf16_kernel<16, false>), grid, block, 0, s, ax); break;
            case 32: hipLaunchKernelGGL((attn_bf16_kernel<32, false>), grid, block, 0, s, ax); break;
            case 64: if (a.q_prescaled && attn_q64_enabled() && ltx_attention_q64_fits(ax)) return ltx_launch_attention_q64(ax, s);
                     else if (a.q_prescaled && attn_pipe_enabled()) hipLaunchKernelGGL(attn_pipe64_kernel, grid, block, 0, s, ax);
                     else if (a.q_prescaled) hipLaunchKernelGGL((attn_bf16_kernel<64, true>), grid, block, 0, s, ax);
                     else hipLaunchKernelGGL((attn_bf16_kernel<64, false>), grid, block, 0, s, ax);
                     break;
            case 128: if (a.q_prescaled && ltx_attention_q128_fits(ax)) {
                          // generated one-wave-per-SIMD loop with the fixed first-tile max, then the exact kernel gated on its overflow flag
                          int* flag = nullptr; int ticket = 0;
                          LTX_TRY(ltx_launch_attention_q128(ax, s, &flag, &ticket));
                          ax.gate_flag = flag; ax.gate_ticket = ticket;
                          hipLaunchKernelGGL((attn_bf16_kernel<128, true>), grid, block, 0, s, ax);
                      } else if (a.q_prescaled) hipLaunchKernelGGL((attn_bf16_kernel<128, true>), grid, block, 0, s, ax);
                      else hipLaunchKernelGGL((attn_bf16_kernel<128, false>), grid, block, 0, s, ax);
                      break;
            default: LTX_FAIL(LTX_ERR_UNSUPPORTED, "attention: head_dim must be 16, 32, 64 or 128");
        }
    } else {
        if (a.q_rowsq || a.k_count) LTX_FAIL(LTX_ERR_ARG, "attention: q_rowsq / k_count are bf16 paths");
        if (a.ldq % 4 || a.ldk % 4 || a.ldv % 4 || a.ldo % 4) LTX_FAIL(LTX_ERR_ARG, "attention: strides must be 16-byte aligned");
        dim3 grid((unsigned)cdiv(a.Sq, 64), (unsigned)a.heads, (unsigned)a.B), block(64);
        switch (a.hd) {
            case 16: hipLaunchKernelGGL(attn_f32_kernel<16>, grid, block, 0, s, a); break;
            case 32: hipLaunchKernelGGL(attn_f32_kernel<32>, grid, block, 0, s, a); break;
            case 64: hipLaunchKernelGGL(attn_f32_kernel<64>, grid, block, 0, s, a); break;
            case 128: hipLaunchKernelGGL(attn_f32_kernel<128>, grid, block, 0, s, a); break;
            default: LTX_FAIL(LTX_ERR_UNSUPPORTED, "attention: head_dim must be 16, 32, 64 or 128");
        }
    }
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
