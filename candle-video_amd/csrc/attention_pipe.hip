// Software-pipelined variant of the bf16 flash attention in attention.hip (same data layout, same
// fragment maps, same numerics) for long key sequences (self-attention of the DiT, S = 4992).
//
// PMC on the plain kernel (profiles/, tools/pmc_summary.py): MFMA busy 33 %, ~200 VALU instructions per
// 64-key tile and wave, and tile time = MFMA time + VALU time — the two pipes did NOT overlap, because inside
// one wave the chain S-MFMAs -> softmax VALU -> PV-MFMAs is strictly dependent and co-resident waves fell
// into lockstep.  Here the dependency is broken inside the wave:
//   * S(t+1) = K(t+1).Q^T is issued BEFORE the softmax of tile t (two named accumulator sets, loop unrolled
//     by two so the ping-pong and every LDS buffer offset are compile-time constants) — the softmax VALU work
//     of tile t overlaps the QK^T MFMAs of tile t+1;
//   * K is staged two tiles ahead and V one tile ahead (each double-buffered in LDS), one barrier per tile;
//   * LDS fragment addresses are loop-invariant registers + immediate buffer offsets (no per-tile address VALU).
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PBQ = 128, PBKV = 64;

template <int CPR> __device__ __forceinline__ int pkswz(int row, int c) {
    if constexpr (CPR >= 4) { constexpr int RPB = 16 / CPR; return c ^ ((row / RPB) % CPR); }
    else return c;
}
template <int CPR> __device__ __forceinline__ int pvswz(int row, int c) {
    if constexpr (CPR == 8) return c ^ (((row >> 1) & 1) << 2);
    else if constexpr (CPR == 16) return c ^ ((row & 3) << 2);
    else return c;
}

template <int HD, bool BIAS>
__global__ __launch_bounds__(256, 1) void attn_bf16_pipe_kernel(const AttnArgs a) {
    constexpr int KROW = HD * 2, VROW = HD * 2;
    constexpr int CPR = KROW / 16;
    constexpr int NKS = HD / 16, NDB = HD / 32;
    constexpr int NCH = (PBKV * CPR) / 256;              // 16-B chunks per thread per tile (K and V each)
    constexpr int KT = PBKV * KROW, VT = PBKV * VROW;    // bytes per K / V tile
    static_assert(HD >= 32 && (PBKV * CPR) % 256 == 0, "pipelined attention needs head_dim >= 32");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KT + 2 * VT];
    unsigned char* const Kb = smem;                      // Kb + p*KT
    unsigned char* const Vb = smem + 2 * KT;             // Vb + p*VT

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * PBQ + wave * 32;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + head * HD;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Sk * a.ldk + head * HD;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Sk * a.ldv + head * HD;
    const float* bias = BIAS ? a.bias + (int64_t)b * a.Sk : nullptr;
    const float LOG2E = 1.4426950408889634f;
    const float sc2 = a.scale * LOG2E;
    constexpr bool has_bias = BIAS;
    const float c = has_bias ? 1.0f : sc2;
    constexpr float RESCALE_THR = 5.0f;
    const int nt = (a.Sk + PBKV - 1) / PBKV;
    const bool ragged = (a.Sk % PBKV) != 0;

    bf16x8 qf[NKS];
    {
        int qr = q0 + r; if (qr > a.Sq - 1) qr = a.Sq - 1;
        const bf16_t* qp = Q + (int64_t)qr * a.ldq + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    // staging: thread -> (row, chunk) of the tile, NCH chunks each for K and V
    u32x4 rk[NCH], rv[NCH];
    int st_off_k[NCH], st_off_v[NCH], st_row[NCH], st_c[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = tid + 256 * i;
        st_row[i] = idx / CPR; st_c[i] = idx % CPR;
        st_off_k[i] = st_row[i] * KROW + pkswz<CPR>(st_row[i], st_c[i]) * 16;
        st_off_v[i] = st_row[i] * VROW + pvswz<CPR>(st_row[i], st_c[i]) * 16;
    }
    auto gloadK = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int key = t * PBKV + st_row[i]; if (key > a.Sk - 1) key = a.Sk - 1;
            rk[i] = *reinterpret_cast<const u32x4*>(K + (int64_t)key * a.ldk + st_c[i] * 8);
        }
    };
    auto gloadV = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int key = t * PBKV + st_row[i]; if (key > a.Sk - 1) key = a.Sk - 1;
            rv[i] = *reinterpret_cast<const u32x4*>(V + (int64_t)key * a.ldv + st_c[i] * 8);
        }
    };

    // loop-invariant fragment addresses: the row-dependent swizzle terms only involve lane bits, so ONE base per
    // k-step (K) / per d-block (V) plus compile-time immediates (kb*32, 16*s, 8*u rows) covers every fragment.
    int k_base[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k_base[ks] = r * KROW + pkswz<CPR>(r, 2 * ks + h) * 16;      // + kb*32*KROW
    const int trq = (lane & 15) >> 2, trp = lane & 3, trdh = (lane >> 4) & 1;
    int v_base[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
        const int row = 4 * h + trq;
        v_base[d] = row * VROW + pvswz<CPR>(row, d * 4 + trdh * 2 + (trp >> 1)) * 16 + (trp & 1) * 8;   // + (kb*32+16*s+8*u)*VROW
    }
    static_assert(CPR == 8 || CPR == 4 || CPR == 16, "swizzle/immediate split assumes 32-row periodicity");

    f32x16 acc_o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    auto compute_S = [&](f32x16 (&s)[2], const unsigned char* Ks) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + k_base[ks] + kb * 32 * KROW);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
            }
        }
    };

    // one tile: [prefetch] [S(t+1)] softmax(t) PV(t) [stage] barrier.  P = t & 1 selects the LDS buffers.
    // FAST = steady state (tiles t+1 and t+2 exist, no tail masking): no branches, so the scheduler is free to
    // interleave the S(t+1) MFMAs with the softmax VALU work of tile t.
    auto step = [&](int t, auto p_tag, auto fast_tag, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
        constexpr int P = decltype(p_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        const bool has_next = FAST || t + 1 < nt, has_k2 = FAST || t + 2 < nt;
        const bool masked = !FAST && ragged && t == nt - 1;
        if (has_k2) gloadK(t + 2);
        if (has_next) gloadV(t + 1);
        if (has_next) compute_S(s_nxt, Kb + (P ^ 1) * KT);
        const int kv0 = t * PBKV;
        if (has_bias || masked) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kv0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    float x = s_cur[kb][i];
                    if (has_bias) x = fmaf(x, sc2, bias[key < a.Sk ? key : a.Sk - 1] * LOG2E);
                    if (!FAST && key >= a.Sk) x = -INFINITY;
                    s_cur[kb][i] = x;
                }
        }
        float mt = fmaxf(s_cur[0][0], s_cur[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(fmaxf(mt, s_cur[0][i]), s_cur[1][i]);
        {
            unsigned mu = __float_as_uint(mt);
            auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
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
        float ls = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = __builtin_amdgcn_exp2f(fmaf(s_cur[kb][i], c, mc));
                ls += p;
                pf[kb][i >> 3][i & 7] = (bf16_t)p;
            }
        l_run += ls;
        const unsigned char* Vs = Vb + P * VT;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Vs + v_base[d] + (kb * 32 + 16 * s) * VROW));
                    bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Vs + v_base[d] + (kb * 32 + 16 * s + 8) * VROW));
                    bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], acc_o[d], 0, 0, 0);
                }
        if (has_k2) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(Kb + P * KT + st_off_k[i]) = rk[i];
        }
        if (has_next) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(Vb + (P ^ 1) * VT + st_off_v[i]) = rv[i];
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from hoisting the next tile's reads across (register pressure)
    };

    // prologue: K(0), V(0), K(1) resident; S(0) computed
    gloadK(0); gloadV(0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) { *reinterpret_cast<u32x4*>(Kb + st_off_k[i]) = rk[i]; *reinterpret_cast<u32x4*>(Vb + st_off_v[i]) = rv[i]; }
    if (nt > 1) {
        gloadK(1);
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(Kb + KT + st_off_k[i]) = rk[i];
    }
    __syncthreads();
    f32x16 sA[2], sB[2];
    compute_S(sA, Kb);
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 3 < nt; t += 2) {                       // both tiles of the pair have successors t+1, t+2
        step(t, P0{}, std::true_type{}, sA, sB);
        step(t + 1, P1{}, std::true_type{}, sB, sA);
    }
    for (; t < nt; ++t) {                              // tail (<= 3 tiles): runtime guards + key masking
        if ((t & 1) == 0) step(t, P0{}, std::false_type{}, sA, sB);
        else step(t, P1{}, std::false_type{}, sB, sA);
    }

    float l_tot;
    {
        unsigned lu = __float_as_uint(l_run);
        auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
    const int qr = q0 + r;
    if (qr < a.Sq) {
        bf16_t* O = reinterpret_cast<bf16_t*>(a.o) + ((int64_t)b * a.Sq + qr) * a.ldo + head * HD;
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

}  // namespace

bool ltx_attention_pipe_eligible(const AttnArgs& a, int dtype) {
    // EXPERIMENTAL, off by default: on MI355X it measured 545-593 TFLOP/s against 641 for the plain kernel at
    // 3 waves/SIMD (the second accumulator set costs a wave of occupancy, or spills) — LTX_ATTN_PIPE=1 enables it.
    const char* e = getenv("LTX_ATTN_PIPE");
    if (!e || e[0] != '1') return false;
    return dtype == LTX_DT_BF16 && a.hd == 64 && a.Sk >= 3 * PBKV;
}

int ltx_launch_attention_pipe(const AttnArgs& a, hipStream_t s) {
    dim3 grid((unsigned)cdiv(a.Sq, PBQ), (unsigned)a.heads, (unsigned)a.B), block(256);
    if (a.bias) hipLaunchKernelGGL((attn_bf16_pipe_kernel<64, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_bf16_pipe_kernel<64, false>), grid, block, 0, s, a);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
