// Row-wise normalisation kernels (HBM-bound; algorithmic bytes = read x once + write y once).
//
//  rownorm      : y = act( norm(x) [*w] * (1+scale_b) + shift_b )
//                 RMS  = RmsNorm::forward (ltx_transformer.rs:99-119; f32 statistics) fused with the
//                        AdaLN modulate of LtxVideoTransformerBlock::forward (:874-889, :912-927),
//                        and rmsnorm_channels_first + maybe_apply_scale_shift + SiLU of the VAE
//                        resnet (vae.rs:148-153, 711-739, 755-800) on channels-last rows.
//                 LN   = LayerNormNoParams (ltx_transformer.rs:72-79) + final modulation (:1149-1161).
//  qknorm_rope  : q/k RMSNorm with weight over the FULL inner dim (ltx_transformer.rs:671-672)
//                 fused with apply_rotary_emb (:314-339), in place, for 1 or 2 segments of a fused
//                 QKV row.
//  rope_table   : LtxVideoRotaryPosEmbed::forward (:436-524) -> half-width cos/sin tables.
//
// Mapping: LPR lanes per row (power of two, >= 16-B chunks per row, <= 64), 64/LPR rows per wave,
// 4 waves per block; every lane moves 16 B per access.  The row lives in registers between the
// statistics pass and the write pass (one HBM read, one HBM write), and every lane issues all of
// its loads before the first reduction (memory-level parallelism instead of a dependent loop):
//   WIDE   rows (> LPR chunks): up to NSLOT chunks of ONE row per lane;
//   NARROW rows (<= LPR chunks, e.g. the VAE's 128..1024-channel voxels): NSLOT different rows per
//          lane group, one chunk each.
// Rows that do not fit NSLOT chunks per lane fall back to a re-reading variant (second read hits L2).
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "gemm_common.h"
#include "options.h"

namespace {

constexpr int NSLOT = 8;

__device__ __forceinline__ float group_sum(float v, int lpr) {
    for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__device__ __forceinline__ void finish_chunk(const RowNormArgs& a, const float* f_in, float mean, float rinv, int c, const float* sc,
                                             const float* sh, T* y, bool active) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    const T* w = reinterpret_cast<const T*>(a.weight);
    float f[CH], wv[CH], scv[CH], shv[CH];
    if (w) { Chunk16 wc; wc.u = *reinterpret_cast<const u32x4*>(w + c * CH); chunk_to_f32<T>(wc, wv); }
    if (sc) {
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(sc + c * CH + 4 * q), b4 = *reinterpret_cast<const f32x4*>(sh + c * CH + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) { scv[4 * q + i] = a4[i]; shv[4 * q + i] = b4[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        float n = (f_in[i] - mean) * rinv;
        if (w) n *= wv[i];
        if (sc) n = n * (1.0f + scv[i]) + shv[i];
        if (a.act == 1) n = silu_f(n);
        f[i] = n;
    }
    Chunk16 o; f32_to_chunk<T>(f, o);
    if (active) *reinterpret_cast<u32x4*>(y + c * CH) = o.u;
}

// Same, with the per-channel operands (weight, scale, shift of ONE chunk) already in registers: the NARROW mapping
// keeps a lane on the same chunk for all of its rows, so they are loaded once instead of once per row
// (64 B of modulation per 16 B of data through the L1/TA path otherwise).
template <typename T, bool NT = false>
__device__ __forceinline__ void finish_chunk_regs(const RowNormArgs& a, const float* f_in, float mean, float rinv, int c, bool has_w, bool has_mod,
                                                  const float* wv, const float* scv, const float* shv, T* y, bool active) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    float f[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        float n = (f_in[i] - mean) * rinv;
        if (has_w) n *= wv[i];
        if (has_mod) n = n * (1.0f + scv[i]) + shv[i];
        if (a.act == 1) n = silu_f(n);
        f[i] = n;
    }
    Chunk16 o; f32_to_chunk<T>(f, o);
    if (active) {
        if constexpr (NT) __builtin_nontemporal_store(o.u, reinterpret_cast<u32x4*>(y + c * CH));
        else *reinterpret_cast<u32x4*>(y + c * CH) = o.u;
    }
}

template <typename T>
__device__ __forceinline__ void load_chunk_operands(const RowNormArgs& a, int c, int64_t b, float* wv, float* scv, float* shv) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    const T* w = reinterpret_cast<const T*>(a.weight);
    if (w) { Chunk16 wc; wc.u = *reinterpret_cast<const u32x4*>(w + c * CH); chunk_to_f32<T>(wc, wv); }
    if (a.scale) {
        const float* sc = a.scale + b * a.mod_stride; const float* sh = a.shift + b * a.mod_stride;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(sc + c * CH + 4 * q), b4 = *reinterpret_cast<const f32x4*>(sh + c * CH + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) { scv[4 * q + i] = a4[i]; shv[4 * q + i] = b4[i]; }
        }
    }
}

// RowNormArgs::parts: chunk c of a row that arrives as the K-range sums of the linear layer before it - finished with the layer's
// epilogue expression (gemm_common.h epilogue<>: v = sum + bias, then resid + gate * v or v + resid), rounded to T, written back
// as the new residual stream and returned for the norm.  Ranges added in part order: ((p0 + p1) + p2) + ... (gemm_big.hip).
// NP > 0: the number of ranges known at compile time (four: the shape rule's cut of K >= 8192 at <= 512 rows) - every range's loads are
// issued before the first add, 8 x 16 bytes in flight per chunk instead of one dependent pair at a time (this pass is pure latency at
// 384 rows: 14.6 us -> see docs/lab_notes.md R5.13); NP == 0: a.nparts at run time.  Same order of additions either way.
template <typename T, int NP = 0>
__device__ __forceinline__ Chunk16 deferred_chunk(const RowNormArgs& a, int64_t row, int64_t b, int c, bool active) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    float v[CH];
    const float* pp = a.parts + row * a.D + c * CH;
    if constexpr (NP > 0) {
        f32x4 t[NP][CH / 4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) t[p][q] = *reinterpret_cast<const f32x4*>(pp + (int64_t)p * a.part_stride + 4 * q);
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            f32x4 acc = t[0][q];
#pragma unroll
            for (int p = 1; p < NP; ++p) acc += t[p][q];
            v[4 * q] = acc[0]; v[4 * q + 1] = acc[1]; v[4 * q + 2] = acc[2]; v[4 * q + 3] = acc[3];
        }
    } else {
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) { const f32x4 t = *reinterpret_cast<const f32x4*>(pp + 4 * q); v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3]; }
        for (int p = 1; p < a.nparts; ++p) {
            const float* pq = pp + (int64_t)p * a.part_stride;
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) { const f32x4 t = *reinterpret_cast<const f32x4*>(pq + 4 * q); v[4 * q] += t[0]; v[4 * q + 1] += t[1]; v[4 * q + 2] += t[2]; v[4 * q + 3] += t[3]; }
        }
    }
    if (a.d_bias) {
        Chunk16 bc; bc.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.d_bias) + c * CH);
        float bf[CH]; chunk_to_f32<T>(bc, bf);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] += bf[i];
    }
    Chunk16 rc; rc.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.x) + row * a.ldx + c * CH);
    float r[CH]; chunk_to_f32<T>(rc, r);
    if (a.d_gate) {
        const float* gp = a.d_gate + b * a.d_gate_stride + c * CH;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            const f32x4 gt = *reinterpret_cast<const f32x4*>(gp + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * q + i] = __builtin_fmaf(gt[i], v[4 * q + i], r[4 * q + i]);      // gemm_common.h epilogue<>: one rounding
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] += r[i];
    }
    Chunk16 o; f32_to_chunk<T>(v, o);
    if (active) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.x_out) + row * a.ldx + c * CH) = o.u;
    return o;
}

// MODE 0: WIDE (row cached in registers), 1: NARROW (NSLOT rows per lane group), 2: re-read fallback; DEFER (MODE 0): RowNormArgs::parts
template <typename T, int MODE, int DEFER = 0>          // DEFER: 0 no; 1: RowNormArgs::parts, a.nparts ranges; 4: four ranges (unrolled)
__global__ __launch_bounds__(256) void rownorm_kernel(const RowNormArgs a, int lpr) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr;
    const int sub = lane % lpr;
    const int nch = a.D / CH;
    const float invD = 1.0f / (float)a.D;
    if constexpr (MODE == 1) {
        // a wave owns a contiguous block of NSLOT*rpw rows; slot i of a lane group is row0 + i*rpw.  Addresses advance by a
        // constant stride (no per-row 64-bit multiply), rows past the end are predicated off instead of clamped.
        const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (rpw * NSLOT) + lane / lpr;
        const bool has = sub < nch;
        const T* xp = reinterpret_cast<const T*>(a.x) + row0 * a.ldx + sub * CH;
        T* yp = reinterpret_cast<T*>(a.y) + row0 * a.ldy;
        const int64_t xs = (int64_t)rpw * a.ldx, ys = (int64_t)rpw * a.ldy;
        Chunk16 v[NSLOT];
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            v[i].u = (u32x4){0u, 0u, 0u, 0u};
            if (has && row0 + i * rpw < a.rows) v[i].u = *reinterpret_cast<const u32x4*>(xp + i * xs);
        }
        // per-channel operands of this lane's chunk, for the batch of its first row (reloaded only if a row crosses a batch)
        float wv[CH], scv[CH], shv[CH];
        const int64_t b0 = (row0 < a.rows ? row0 : a.rows - 1) / a.rows_per_batch;      // ONE 64-bit division per lane
        const int64_t b0_end = (b0 + 1) * a.rows_per_batch;                               // first row of the next batch
        if (has) load_chunk_operands<T>(a, sub, b0, wv, scv, shv);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int64_t row = row0 + (int64_t)i * rpw;
            const bool active = row < a.rows;
            float f[CH]; chunk_to_f32<T>(v[i], f);
            float mean = 0.f;
            if (a.kind == 1) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j) s += f[j];
                mean = group_sum(s, lpr) * invD;
            }
            float ss = 0.f;
            if (has) {
#pragma unroll
                for (int j = 0; j < CH; ++j) { float d = f[j] - mean; ss += d * d; }
            }
            ss = group_sum(ss, lpr);
            const float rinv = __builtin_amdgcn_rsqf(ss * invD + a.eps);
            if (has) {
                if (row < b0_end) {
                    finish_chunk_regs<T>(a, f, mean, rinv, sub, a.weight != nullptr, a.scale != nullptr, wv, scv, shv, yp + i * ys, active);
                } else if (active) {                                                     // a row of a later batch (rare)
                    const int64_t b = row / a.rows_per_batch;
                    const float* sc = a.scale ? a.scale + b * a.mod_stride : nullptr;
                    const float* sh = a.shift ? a.shift + b * a.mod_stride : nullptr;
                    finish_chunk<T>(a, f, mean, rinv, sub, sc, sh, yp + i * ys, active);
                }
            }
        }
    } else {
        const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rpw + lane / lpr;
        const bool active = row < a.rows;
        const int64_t rr = active ? row : a.rows - 1;
        const T* x = reinterpret_cast<const T*>(a.x) + rr * a.ldx;
        T* y = reinterpret_cast<T*>(a.y) + rr * a.ldy;
        const int64_t b = rr / a.rows_per_batch;
        const float* sc = a.scale ? a.scale + b * a.mod_stride : nullptr;
        const float* sh = a.shift ? a.shift + b * a.mod_stride : nullptr;
        if constexpr (MODE == 0) {
            Chunk16 v[NSLOT];
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                int c = sub + i * lpr;
                v[i].u = (u32x4){0u, 0u, 0u, 0u};
                if constexpr (DEFER != 0) { if (c < nch) v[i] = deferred_chunk<T, DEFER == 4 ? 4 : 0>(a, rr, b, c, active); }
                else if (c < nch) v[i].u = *reinterpret_cast<const u32x4*>(x + c * CH);
            }
            float mean = 0.f;
            if (a.kind == 1) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) { float f[CH]; chunk_to_f32<T>(v[i], f);
#pragma unroll
                    for (int j = 0; j < CH; ++j) s += f[j]; }
                mean = group_sum(s, lpr) * invD;
            }
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                if (sub + i * lpr < nch) {
                    float f[CH]; chunk_to_f32<T>(v[i], f);
#pragma unroll
                    for (int j = 0; j < CH; ++j) { float d = f[j] - mean; ss += d * d; }
                }
            }
            ss = group_sum(ss, lpr);
            const float rinv = 1.0f / sqrtf(ss * invD + a.eps);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                int c = sub + i * lpr;
                if (c < nch) { float f[CH]; chunk_to_f32<T>(v[i], f); finish_chunk<T>(a, f, mean, rinv, c, sc, sh, y, active); }
            }
        } else {
            float mean = 0.f;
            if (a.kind == 1) {
                float s = 0.f;
                for (int c = sub; c < nch; c += lpr) {
                    Chunk16 v; v.u = *reinterpret_cast<const u32x4*>(x + c * CH);
                    float f[CH]; chunk_to_f32<T>(v, f);
#pragma unroll
                    for (int i = 0; i < CH; ++i) s += f[i];
                }
                mean = group_sum(s, lpr) * invD;
            }
            float ss = 0.f;
            for (int c = sub; c < nch; c += lpr) {
                Chunk16 v; v.u = *reinterpret_cast<const u32x4*>(x + c * CH);
                float f[CH]; chunk_to_f32<T>(v, f);
#pragma unroll
                for (int i = 0; i < CH; ++i) { float d = f[i] - mean; ss += d * d; }
            }
            ss = group_sum(ss, lpr);
            const float rinv = 1.0f / sqrtf(ss * invD + a.eps);
            for (int c = sub; c < nch; c += lpr) {
                Chunk16 v; v.u = *reinterpret_cast<const u32x4*>(x + c * CH);
                float f[CH]; chunk_to_f32<T>(v, f);
                finish_chunk<T>(a, f, mean, rinv, c, sc, sh, y, active);
            }
        }
    }
}

// Wide rows, several per wave, software-pipelined (round 4).  The one-row-per-wave form (MODE 0 above) has every wave of the launch
// resident at once and therefore in the same phase: the chip first reads the whole matrix, then writes the whole result - 41 MB
// of the DiT's [4992, 2048] pass take 12.7 us where a pure elementwise map of the same bytes takes 6.6 us (tools/norm_probe.py).
// Here a wave walks R rows (row = wave + k * waves) with the loads of the next PD rows in flight while it reduces, modulates and
// stores the current one, so reads and writes overlap; each lane keeps the modulation operands of its NS chunks in registers when
// all rows of the wave share a batch element (they do for B = 1; otherwise they are re-loaded per row).  Per row the arithmetic is
// MODE 0's, expression for expression: same bits.
template <typename T, int NS, int R, int PD>
__global__ __launch_bounds__(256) void rownorm_rows_kernel(const RowNormArgs a, int total_waves) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    static_assert(PD >= 1 && PD < R, "prefetch depth");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * 4 + wave;
    const int nch = a.D / CH;
    const float invD = 1.0f / (float)a.D;
    auto row_of = [&](int k) { return w0 + (int64_t)k * total_waves; };
    Chunk16 v[PD + 1][NS];
    auto load_row = [&](int k, Chunk16 (&dst)[NS]) {
        const int64_t row = row_of(k);
        const bool in = row < a.rows;
        const T* x = reinterpret_cast<const T*>(a.x) + (in ? row : a.rows - 1) * a.ldx;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = lane + i * 64;
            dst[i].u = (u32x4){0u, 0u, 0u, 0u};
            if (in && c < nch) dst[i].u = *reinterpret_cast<const u32x4*>(x + c * CH);
        }
    };
    // operands of this lane's chunks for the batch element of the wave's first row
    const int64_t b_first = (row_of(0) < a.rows ? row_of(0) : a.rows - 1) / a.rows_per_batch;
    const int64_t last_row = row_of(R - 1) < a.rows ? row_of(R - 1) : a.rows - 1;
    const bool one_batch = last_row / a.rows_per_batch == b_first;
    float wv[NS][CH], scv[NS][CH], shv[NS][CH];
#pragma unroll
    for (int i = 0; i < NS; ++i) if (lane + i * 64 < nch) load_chunk_operands<T>(a, lane + i * 64, b_first, wv[i], scv[i], shv[i]);
#pragma unroll
    for (int k = 0; k < PD; ++k) load_row(k, v[k]);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (k + PD < R) load_row(k + PD, v[(k + PD) % (PD + 1)]);
        Chunk16 (&cur)[NS] = v[k % (PD + 1)];
        const int64_t row = row_of(k);
        const bool active = row < a.rows;
        float mean = 0.f;
        if (a.kind == 1) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) { float f[CH]; chunk_to_f32<T>(cur[i], f);
#pragma unroll
                for (int j = 0; j < CH; ++j) s += f[j]; }
            mean = group_sum(s, 64) * invD;
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (lane + i * 64 < nch) {
                float f[CH]; chunk_to_f32<T>(cur[i], f);
#pragma unroll
                for (int j = 0; j < CH; ++j) { float d = f[j] - mean; ss += d * d; }
            }
        }
        ss = group_sum(ss, 64);
        const float rinv = 1.0f / sqrtf(ss * invD + a.eps);
        T* y = reinterpret_cast<T*>(a.y) + (active ? row : a.rows - 1) * a.ldy;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = lane + i * 64;
            if (c >= nch) continue;
            asm volatile("" : "+v"(cur[i].u));              // (keeps the f32 expansion of the statistics from living on: registers)
            float f[CH]; chunk_to_f32<T>(cur[i], f);
            if (one_batch) finish_chunk_regs<T>(a, f, mean, rinv, c, a.weight != nullptr, a.scale != nullptr, wv[i], scv[i], shv[i], y, active);
            else {
                const int64_t b = (active ? row : a.rows - 1) / a.rows_per_batch;
                const float* sc = a.scale ? a.scale + b * a.mod_stride : nullptr;
                const float* sh = a.shift ? a.shift + b * a.mod_stride : nullptr;
                finish_chunk<T>(a, f, mean, rinv, c, sc, sh, y, active);
            }
        }
    }
}

// RMS norm (+ weight, modulation, activation) of rows whose sums of squares are already known (RowNormArgs::presum, the
// by-product of the GEMM that wrote x): one thread per 16-byte chunk, no reduction, no wave waiting for a row - where the
// one-row-per-wave kernel takes 12.6 us for the DiT's [4992, 2048] pass, a pure map of the same bytes takes 6.5 us
// (tools/norm_probe.py).  rinv is formed exactly as in rownorm_kernel; the partials are summed in ascending order.
template <typename T, int R, bool WIDE, bool NT = false>       // WIDE: nch >= 64 (a wave lies inside one row group); NT: streaming hints on x and y
__global__ __launch_bounds__(256) void rownorm_presum_kernel(const RowNormArgs a, int nch, int rpb) {
    // block = rpb * R whole rows: thread (sub, c) owns 16-byte chunk c of rows row0 + sub * R .. + R - 1 (nch a power of two
    // <= 256, rpb = 256 / nch).  What bounds the row-reducing kernel is not its reduction but the vector-memory instruction
    // count: 64 B of f32 modulation operands per 16 B of data, re-loaded for every row (9 - 20 loads per 16 B chunk).  Here a
    // thread loads its chunk's operands ONCE for R rows, all R data chunks are in flight together, and 1 / rms comes from the
    // R lanes of each wave that sum the rows' partials (broadcast with readlane): 4 R + 8 memory instructions per R chunks.
    constexpr int CH = ElemTraits<T>::CHUNK;
    const int lane = threadIdx.x & 63;
    const int sub = threadIdx.x / nch, c = threadIdx.x - sub * nch;
    const int64_t row0 = ((int64_t)blockIdx.x * rpb + sub) * R;
    if (row0 >= a.rows) return;
    // Every load of the block is issued before anything waits: the R data chunks first, then the rows' partials (all of them at
    // once - a loop of "load 16 bytes, wait, add" paid one memory latency per four partials, four in a row at D = 2048, before the
    // data loads were even issued: 10.4 us per [4992, 2048] pass against 6.5 for a plain map of the same bytes), then the operands.
    Chunk16 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < a.rows ? row0 + r : a.rows - 1;
        const u32x4* xp = reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.x) + row * a.ldx + c * CH);
        if constexpr (NT) v[r].u = __builtin_nontemporal_load(xp); else v[r].u = *xp;
    }
    // 1 / rms of the R rows of this thread: lanes 0 .. R-1 of the wave each sum one row's partials (a wave lies inside one sub
    // when nch >= 64; for narrower rows every lane does its own rows' sums).  Ascending order; groups past presum_n add 0.0f,
    // which changes no bit of a non-negative sum.
    float rinv[R];
    auto partials_of = [&](int64_t row, f32x4 (&t)[4]) {
        const float* ps = a.presum + (row < a.rows ? row : a.rows - 1) * a.presum_n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                         // no branch: a group past presum_n re-reads the last one and is zeroed
            const bool in = 4 * q < a.presum_n;
            const f32x4 u = *reinterpret_cast<const f32x4*>(ps + (in ? 4 * q : a.presum_n - 4));
            t[q] = in ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto rinv_from = [&](int64_t row, const f32x4 (&t)[4]) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ss += t[q][0]; ss += t[q][1]; ss += t[q][2]; ss += t[q][3]; }
        if (a.presum_n > 16) {                                // wider rows than the DiT's: the rest in a loop
            const float* ps = a.presum + (row < a.rows ? row : a.rows - 1) * a.presum_n;
            for (int g4 = 16; g4 < a.presum_n; g4 += 4) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ps + g4);
                ss += u[0]; ss += u[1]; ss += u[2]; ss += u[3];
            }
        }
        return 1.0f / sqrtf(ss * (1.0f / (float)a.D) + a.eps);
    };
    f32x4 pt[WIDE ? 1 : R][4];
    if constexpr (WIDE) { if (lane < R) partials_of(row0 + lane, pt[0]); }
    else {
#pragma unroll
        for (int r = 0; r < R; ++r) partials_of(row0 + r, pt[r]);
    }
    const uint32_t b0 = (uint32_t)row0 / (uint32_t)a.rows_per_batch;
    const int64_t last = row0 + R - 1 < a.rows ? row0 + R - 1 : a.rows - 1;
    const bool one_batch = (uint32_t)last / (uint32_t)a.rows_per_batch == b0;
    float wv[CH], scv[CH], shv[CH];
    load_chunk_operands<T>(a, c, (int64_t)b0, wv, scv, shv);
    if constexpr (WIDE) {
        float mine = 0.f;
        if (lane < R) mine = rinv_from(row0 + lane, pt[0]);
#pragma unroll
        for (int r = 0; r < R; ++r) rinv[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), r));
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rinv[r] = rinv_from(row0 + r, pt[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
        const bool active = row < a.rows;
        float f[CH]; chunk_to_f32<T>(v[r], f);
        T* y = reinterpret_cast<T*>(a.y) + (active ? row : a.rows - 1) * a.ldy;
        if (one_batch) finish_chunk_regs<T, NT>(a, f, 0.f, rinv[r], c, a.weight != nullptr, a.scale != nullptr, wv, scv, shv, y, active);
        else if (active) {
            const uint32_t b = (uint32_t)row / (uint32_t)a.rows_per_batch;
            finish_chunk<T>(a, f, 0.f, rinv[r], c, a.scale ? a.scale + (int64_t)b * a.mod_stride : nullptr, a.shift ? a.shift + (int64_t)b * a.mod_stride : nullptr, y, true);
        }
    }
}

// The DiT's case of the map above with nothing left to decide at run time: bf16, a row of nch chunks
// (nch = 64 / 128 / 256), modulation present, no weight, no activation, groups of R rows inside one batch element, no ragged tail.
// rownorm_presum_kernel serves every combination and runs ~550 instructions per wave at 20 000 waves per pass - as many issue
// cycles as the pass has memory time (a bare copy of the same bytes in the same geometry: 7.5 us; that kernel: 10.4-11.6,
// tools/norm_diag.py).  Same expressions in the same order: same bits (tests/test_gpu_q2fold.py).
template <int R>
__global__ __launch_bounds__(256) void rownorm_presum_lean_kernel(const RowNormArgs a, int nch, int rpb) {
    const int lane = threadIdx.x & 63;
    const int sub = threadIdx.x / nch, c = threadIdx.x - sub * nch;
    const uint32_t row0 = ((uint32_t)blockIdx.x * (uint32_t)rpb + (uint32_t)sub) * R;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)row0 * a.ldx + c * 8;
    u32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const u32x4*>(x + (int64_t)r * a.ldx);
    f32x4 pt[4];
    if (lane < R) {
        const float* ps = a.presum + (int64_t)(row0 + lane) * a.presum_n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = 4 * q < a.presum_n;
            const f32x4 u = *reinterpret_cast<const f32x4*>(ps + (in ? 4 * q : a.presum_n - 4));
            pt[q] = in ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const uint32_t b = row0 / (uint32_t)a.rows_per_batch;
    const float* sc = a.scale + (int64_t)b * a.mod_stride + c * 8; const float* sh = a.shift + (int64_t)b * a.mod_stride + c * 8;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
    float mine = 0.f;
    if (lane < R) {
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ss += pt[q][0]; ss += pt[q][1]; ss += pt[q][2]; ss += pt[q][3]; }
        mine = 1.0f / sqrtf(ss * (1.0f / (float)a.D) + a.eps);
    }
    const float scv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    const float shv[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    bf16_t* y = reinterpret_cast<bf16_t*>(a.y) + (int64_t)row0 * a.ldy + c * 8;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float rinv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), r));
        Chunk16 ch; ch.u = v[r];
        float f[8]; chunk_to_f32<bf16_t>(ch, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float n = (f[i] - 0.f) * rinv;
            n = n * (1.0f + scv[i]) + shv[i];
            f[i] = n;
        }
        Chunk16 o; f32_to_chunk<bf16_t>(f, o);
        *reinterpret_cast<u32x4*>(y + (int64_t)r * a.ldy) = o.u;
    }
}

#ifdef LTX_NORM_DIAG
// Diagnosis only (tools/norm_diag.py): what a streaming pass over [rows, D] bf16 costs in this library's launch geometry.
// var 0: copy, one 16-byte chunk per thread, consecutive threads consecutive chunks (grid = chunks / 256)
// var 1: the same, four chunks per thread 4 KiB apart (the presum kernel's pattern), var 2: four chunks per thread 256 x 16 B apart (a block's chunks contiguous)
__global__ __launch_bounds__(256) void norm_diag_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t nchunks, int var, int rowchunks) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (var == 0) { if (t < nchunks) y[t] = x[t]; return; }
    u32x4 v[4];
    if (var == 1) {
        const int64_t row0 = (t / rowchunks) * 4, c = t % rowchunks;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[(row0 + r) * rowchunks + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[(row0 + r) * rowchunks + c] = v[r];
    } else {
        const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[base + r * 256];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[base + r * 256] = v[r];
    }
}
extern "C" int ltx_dbg_norm_diag(const void* x, void* y, long long nchunks, int var, int rowchunks, void* stream) {
    const unsigned grid = (unsigned)(var == 0 ? (nchunks + 255) / 256 : nchunks / 1024);
    hipLaunchKernelGGL(norm_diag_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, (u32x4*)y, (int64_t)nchunks, var, rowchunks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

// CACHED: the (<= NSLOT chunks per lane) segment stays in registers between the two passes
template <typename T, bool CACHED>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const QkNormRopeArgs a, int lpr) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rpw + lane / lpr;
    const int sub = lane % lpr;
    const bool active = row < a.rows;
    const int64_t rr = active ? row : a.rows - 1;
    const int nch = a.D / CH;
    const float* cs = a.cos ? a.cos + rr * (a.D / 2) : nullptr;
    const float* sn = a.sin ? a.sin + rr * (a.D / 2) : nullptr;
    auto finish = [&](Chunk16 v, int c, float rinv, const T* w, const T* wb, T* x) {
        float f[CH]; chunk_to_f32<T>(v, f);
        Chunk16 wc; wc.u = *reinterpret_cast<const u32x4*>(w + c * CH);
        float wv[CH]; chunk_to_f32<T>(wc, wv);
#pragma unroll
        for (int i = 0; i < CH; ++i) f[i] = f[i] * rinv * wv[i];
        if (wb) {                                          // a consumer's weight folded in (QkNormRopeArgs::w0b)
            Chunk16 wc2; wc2.u = *reinterpret_cast<const u32x4*>(wb + c * CH);
            float wv2[CH]; chunk_to_f32<T>(wc2, wv2);
#pragma unroll
            for (int i = 0; i < CH; ++i) f[i] *= wv2[i];
        }
        if (cs) {
            float co[CH / 2], si[CH / 2];
            if constexpr (CH == 8) {
                f32x4 c4 = *reinterpret_cast<const f32x4*>(cs + c * 4), s4 = *reinterpret_cast<const f32x4*>(sn + c * 4);
#pragma unroll
                for (int p = 0; p < 4; ++p) { co[p] = c4[p]; si[p] = s4[p]; }
            } else {
#pragma unroll
                for (int p = 0; p < CH / 2; ++p) { co[p] = cs[c * (CH / 2) + p]; si[p] = sn[c * (CH / 2) + p]; }
            }
#pragma unroll
            for (int p = 0; p < CH / 2; ++p) {
                float re = f[2 * p], im = f[2 * p + 1];
                f[2 * p] = re * co[p] - im * si[p];          // x*cos + (-x_imag)*sin
                f[2 * p + 1] = im * co[p] + re * si[p];      // x*cos + ( x_real)*sin
            }
        }
        Chunk16 o; f32_to_chunk<T>(f, o);
        if (active) *reinterpret_cast<u32x4*>(x + c * CH) = o.u;
    };
    for (int seg = 0; seg < a.nseg; ++seg) {
        T* x = reinterpret_cast<T*>(a.x) + rr * a.ld + (int64_t)seg * (a.seg_stride ? a.seg_stride : a.D);
        const T* w = reinterpret_cast<const T*>(seg == 0 ? a.w0 : a.w1);
        const T* wb = seg == 0 ? reinterpret_cast<const T*>(a.w0b) : nullptr;
        const float oscale = seg == 0 ? a.out_scale0 : 1.0f;
        if constexpr (CACHED) {
            Chunk16 v[NSLOT];
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                int c = sub + i * lpr;
                v[i].u = (u32x4){0u, 0u, 0u, 0u};
                if (c < nch) v[i].u = *reinterpret_cast<const u32x4*>(x + c * CH);
            }
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) { float f[CH]; chunk_to_f32<T>(v[i], f);
#pragma unroll
                for (int j = 0; j < CH; ++j) ss += f[j] * f[j]; }
            ss = group_sum(ss, lpr);
            const float rinv = oscale / sqrtf(ss / (float)a.D + a.eps);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) { int c = sub + i * lpr; if (c < nch) finish(v[i], c, rinv, w, wb, x); }
        } else {
            float ss = 0.f;
            for (int c = sub; c < nch; c += lpr) {
                Chunk16 v; v.u = *reinterpret_cast<const u32x4*>(x + c * CH);
                float f[CH]; chunk_to_f32<T>(v, f);
#pragma unroll
                for (int i = 0; i < CH; ++i) ss += f[i] * f[i];
            }
            ss = group_sum(ss, lpr);
            const float rinv = oscale / sqrtf(ss / (float)a.D + a.eps);
            for (int c = sub; c < nch; c += lpr) { Chunk16 v; v.u = *reinterpret_cast<const u32x4*>(x + c * CH); finish(v, c, rinv, w, wb, x); }
        }
    }
}

// Fast path of the kernel above for rows of <= NS*lpr chunks (the DiT's 2048-wide q/k: 4 chunks per lane): the cos/sin
// chunks of a lane are loaded ONCE and reused by both segments (q and k share the table), and the loads of every
// segment are issued before the first reduction (memory-level parallelism; the slow kernel read the tables twice and
// serialised the segments).
template <typename T, int NS, int OCC = 5>      // OCC: blocks per CU the register allocation must allow (NS = 4: 94 registers, five)
__global__ __launch_bounds__(256, OCC) void qknorm_rope_fused_kernel(const QkNormRopeArgs a, int lpr) {
    constexpr int CH = ElemTraits<T>::CHUNK;
    static_assert(CH == 8, "bf16 rows only");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rpw + lane / lpr;
    const int sub = lane % lpr;
    const bool active = row < a.rows;
    const int64_t rr = active ? row : a.rows - 1;
    const int nch = a.D / CH;
    const bool rope = a.cos != nullptr;
    const float* cs = rope ? a.cos + rr * (a.D / 2) : nullptr;
    const float* sn = rope ? a.sin + rr * (a.D / 2) : nullptr;
    T* x0 = reinterpret_cast<T*>(a.x) + rr * a.ld;
    const int64_t sst = a.seg_stride ? a.seg_stride : a.D;
    Chunk16 v[2][NS];
    f32x4 co[NS], si[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int c = sub + i * lpr;
        const bool in = c < nch;
        v[0][i].u = (u32x4){0u, 0u, 0u, 0u}; v[1][i].u = (u32x4){0u, 0u, 0u, 0u};
        if (in) v[0][i].u = *reinterpret_cast<const u32x4*>(x0 + c * CH);
        if (in && a.nseg == 2) v[1][i].u = *reinterpret_cast<const u32x4*>(x0 + sst + c * CH);
        if (in && rope) { co[i] = *reinterpret_cast<const f32x4*>(cs + c * 4); si[i] = *reinterpret_cast<const f32x4*>(sn + c * 4); }
    }
    // both row statistics first, then chunk-major: chunk i of q and of k are finished together with ONE copy of its cos / sin
    // in registers, which then die (seg-major order kept all four table chunks and both rows alive: 118 VGPRs = 4 waves per SIMD
    // = 4096 wave slots for the DiT's 4992 row-waves, i.e. a second, mostly empty round; chunk-major fits 5 per SIMD)
    float rinv[2] = {0.f, 0.f};
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        if (seg >= a.nseg) break;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) { float f[CH]; chunk_to_f32<T>(v[seg][i], f);
#pragma unroll
            for (int j = 0; j < CH; ++j) ss += f[j] * f[j]; }
        ss = group_sum(ss, lpr);
        rinv[seg] = (seg == 0 ? a.out_scale0 : 1.0f) / sqrtf(ss / (float)a.D + a.eps);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int c = sub + i * lpr;
        if (c >= nch) continue;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            if (seg >= a.nseg) break;
            const T* w = reinterpret_cast<const T*>(seg == 0 ? a.w0 : a.w1);
            // (opaque copy: without it the f32 expansion of all eight chunks made for the row statistics is kept alive - 64 registers)
            asm volatile("" : "+v"(v[seg][i].u));
            float f[CH]; chunk_to_f32<T>(v[seg][i], f);
            Chunk16 wc; wc.u = *reinterpret_cast<const u32x4*>(w + c * CH);
            float wv[CH]; chunk_to_f32<T>(wc, wv);
#pragma unroll
            for (int j = 0; j < CH; ++j) f[j] = f[j] * rinv[seg] * wv[j];
            if (seg == 0 && a.w0b) {                       // a consumer's weight folded in (QkNormRopeArgs::w0b)
                Chunk16 wc2; wc2.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.w0b) + c * CH);
                float wv2[CH]; chunk_to_f32<T>(wc2, wv2);
#pragma unroll
                for (int j = 0; j < CH; ++j) f[j] *= wv2[j];
            }
            if (rope) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float re = f[2 * p], im = f[2 * p + 1];
                    f[2 * p] = re * co[i][p] - im * si[i][p];
                    f[2 * p + 1] = im * co[i][p] + re * si[i][p];
                }
            }
            Chunk16 o; f32_to_chunk<T>(f, o);
            if (active) *reinterpret_cast<u32x4*>(x0 + seg * sst + c * CH) = o.u;
            __builtin_amdgcn_sched_barrier(0);             // one (chunk, segment) at a time: the scheduler's interleave of all eight costs 30 registers
        }
    }
}

__global__ void rope_table_kernel(const RopeTableArgs a) {
    const int half = a.D / 2;
    const int64_t S = (int64_t)a.F * a.H * a.W;
    const int64_t total = (int64_t)a.B * S * half;
    const int rem_pairs = (a.D % 6) / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % half);
        const int64_t row = i / half;
        float co = 1.0f, si = 0.0f;
        if (p >= rem_pairs) {
            const int fi = p - rem_pairs;
            const int step = fi / 3, ax = fi - step * 3;
            float g;
            if (a.use_coords) {
                g = a.coords[row * 3 + ax] * a.gscale[ax];
            } else {
                const int64_t s = row % S;
                const int w = (int)(s % a.W); const int64_t t1 = s / a.W;
                const int h = (int)(t1 % a.H); const int f = (int)(t1 / a.H);
                const float idx = ax == 0 ? (float)f : (ax == 1 ? (float)h : (float)w);
                g = idx * a.gscale[ax];
            }
            const float ang = a.freqs[step] * (g * 2.0f - 1.0f);
            co = cosf(ang); si = sinf(ang);
        }
        a.cos[i] = co; a.sin[i] = si;
    }
}

// Canonical per-row partial sums of squares (GemmArgs::rowsq) of a stored matrix: one f32 per (row, 128-column group) =
// sequential sum over the group's 32 four-column leaves (ltx_rowsq_leaf), ascending.  The stand-alone form of what
// gemm_asm16's epilogue writes as a by-product - same order, same bits.  A wave takes 64 (row, group) pairs: in step c lane L
// would read leaf c of pair L (uncoalesced), so the roles are transposed through LDS: each half-wave reads ONE pair's 256
// contiguous bytes (bf16) per step, writes its 32 leaves, and after 32 steps every lane sums one pair's leaves in order.
template <typename T>
__global__ __launch_bounds__(256) void rowsq_kernel(const T* x, int64_t rows, int N, int ld, float* out) {
    __shared__ float leaves[4][64][33];                    // [wave][pair][leaf] (+1: conflict-free column reads)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, cg = lane & 31;
    const int ng = (N + 127) / 128;
    const int64_t npairs = rows * ng;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
#pragma unroll 4
    for (int st = 0; st < 32; ++st) {
        const int64_t pr = p0 + 2 * st + half;
        float leaf = 0.f;
        if (pr < npairs) {
            const int64_t row = pr / ng; const int g = (int)(pr - row * ng);
            const int n = g * 128 + 4 * cg;
            if (n < N) {                                   // N % 4 == 0: a leaf is inside or outside as a whole
                if constexpr (std::is_same<T, float>::value) leaf = ltx_rowsq_leaf(*reinterpret_cast<const f32x4*>(x + row * ld + n));
                else leaf = ltx_rowsq_leaf(*reinterpret_cast<const bf16x4*>(x + row * ld + n));
            }
        }
        leaves[wave][2 * st + half][cg] = leaf;
    }
    __syncthreads();
    const int64_t pr = p0 + lane;
    if (pr < npairs) {
        float s = leaves[wave][lane][0];
#pragma unroll
        for (int c = 1; c < 32; ++c) s += leaves[wave][lane][c];
        out[pr] = s;
    }
}

// Few rows (C1's 384 tokens, the 128 text rows of T5): ONE ROW PER BLOCK, a chunk (two beyond 2048 columns) per thread.  With a wave per
// row such a pass is 384 waves on 256 CUs, each walking a dependent chain of four chunks (and, finishing deferred K ranges, 1 KiB of
// loads per lane): pure latency - 7.0 us plain, 14.6 us with four ranges.  Four waves per row quarter the chain and put 4 x the loads
// in flight.  Row statistic: lane butterfly, then the four wave sums in wave order - the plain and the deferred form share it, so
// finishing the ranges here returns the bits of in-launch reduction + this norm (tests/test_gpu_defer.py).
template <typename T, int DEFER>          // DEFER: 0 no; 1: a.nparts ranges; 4: four ranges (unrolled)
__global__ __launch_bounds__(256) void rownorm_block_kernel(const RowNormArgs a) {
    constexpr int CH = ElemTraits<T>::CHUNK, NC = 2;
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int nch = a.D / CH;
    const float invD = 1.0f / (float)a.D;
    const T* x = reinterpret_cast<const T*>(a.x) + row * a.ldx;
    T* y = reinterpret_cast<T*>(a.y) + row * a.ldy;
    const int64_t b = row / a.rows_per_batch;
    const float* sc = a.scale ? a.scale + b * a.mod_stride : nullptr;
    const float* sh = a.shift ? a.shift + b * a.mod_stride : nullptr;
    Chunk16 v[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        v[i].u = (u32x4){0u, 0u, 0u, 0u};
        if (c < nch) {
            if constexpr (DEFER != 0) v[i] = deferred_chunk<T, DEFER == 4 ? 4 : 0>(a, row, b, c, true);
            else v[i].u = *reinterpret_cast<const u32x4*>(x + c * CH);
        }
    }
    auto block_sum = [&](float t, int slot) {
        t = group_sum(t, 64);
        if (lane == 0) red[slot][wave] = t;
        __syncthreads();
        return ((red[slot][0] + red[slot][1]) + red[slot][2]) + red[slot][3];
    };
    float mean = 0.f;
    if (a.kind == 1) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) { float f[CH]; chunk_to_f32<T>(v[i], f);
#pragma unroll
            for (int j = 0; j < CH; ++j) t += f[j]; }
        mean = block_sum(t, 0) * invD;
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (tid + i * 256 < nch) {
            float f[CH]; chunk_to_f32<T>(v[i], f);
#pragma unroll
            for (int j = 0; j < CH; ++j) { const float d = f[j] - mean; ss += d * d; }
        }
    }
    ss = block_sum(ss, 1);
    const float rinv = 1.0f / sqrtf(ss * invD + a.eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        if (c < nch) { float f[CH]; chunk_to_f32<T>(v[i], f); finish_chunk<T>(a, f, mean, rinv, c, sc, sh, y, true); }
    }
}

int pick_lpr(int nch) {
    int lpr = 1;
    while (lpr < nch && lpr < 64) lpr <<= 1;
    return lpr;
}

template <typename T>
void launch_rownorm_t(const RowNormArgs& a, int lpr, int nch, hipStream_t s) {
    const int rpw = 64 / lpr;
    if (nch <= lpr && a.rows >= (int64_t)rpw * NSLOT * 4 * 64) {       // narrow rows, enough of them: NSLOT rows per lane group
        const int64_t rows_per_block = (int64_t)4 * rpw * NSLOT;
        LTX_LAUNCH_TIMED((rownorm_kernel<T, 1>), dim3((unsigned)cdiv64(a.rows, rows_per_block)), dim3(256), 0, s, a, lpr);
    } else if (a.rows <= 512 && nch > 64 && nch <= 512) {                  // few wide rows: one row per block
        const dim3 grid((unsigned)a.rows);
        if (a.parts && a.nparts == 4) LTX_LAUNCH_TIMED((rownorm_block_kernel<T, 4>), grid, dim3(256), 0, s, a);
        else if (a.parts) LTX_LAUNCH_TIMED((rownorm_block_kernel<T, 1>), grid, dim3(256), 0, s, a);
        else LTX_LAUNCH_TIMED((rownorm_block_kernel<T, 0>), grid, dim3(256), 0, s, a);
    } else {
        // (measured and left out, round 3: four rows per wave with the scale / shift operands cached in registers - 312 blocks
        // instead of 1248 - ran the DiT's 4992 x 2048 rows in 15.9 us against 12.8 us: the pass is bound by how many row loads are
        // in flight across the chip, not by the operand traffic through the vector-memory path)
        const int64_t rows_per_block = 4 * rpw;
        dim3 grid((unsigned)cdiv64(a.rows, rows_per_block));
        // wide rows (one wave per row, <= 4 chunks per lane), enough of them: four rows per wave, two in flight (LTX_ROWNORM_ROWS=0: one row per wave)
#ifdef LTX_EXPERIMENTS     // x_rownorm_rows=1: measured 17.3 us against 12.7 us (lab notes R4.4)
        if (lpr == 64 && nch <= 4 * 64 && a.rows >= 2048 && ltx_exp("rownorm_rows", 0) == 1) {
            constexpr int R = 4;
            const int waves = (int)cdiv64(a.rows, R);
            const int blocks = (waves + 3) / 4;
            LTX_LAUNCH_TIMED((rownorm_rows_kernel<T, 4, R, 2>), dim3((unsigned)blocks), dim3(256), 0, s, a, blocks * 4);
            return;
        }
#endif
        // LTX_ROWNORM_OCC=n (experiment): at most n blocks per CU, by asking for 160 KiB / n of LDS the kernel never touches -
        // several generations of blocks instead of one, so that the stores of one overlap the loads of the next
        int shm = 0;
        { const int n = ltx_exp("rownorm_occ", 0); if (n >= 1 && n <= 8) shm = 163840 / n - 1024; }
        if (shm > 65536) {
            static std::atomic<unsigned long long> occ_devs{0};
            (void)ltx_set_max_dyn_smem(occ_devs, reinterpret_cast<const void*>(&rownorm_kernel<T, 0>), 163840);
        }
        if (a.parts && a.nparts == 4) LTX_LAUNCH_TIMED((rownorm_kernel<T, 0, 4>), grid, dim3(256), 0, s, a, lpr);       // (launcher-checked: WIDE rows)
        else if (a.parts) LTX_LAUNCH_TIMED((rownorm_kernel<T, 0, 1>), grid, dim3(256), 0, s, a, lpr);
        else if (nch <= NSLOT * lpr) LTX_LAUNCH_TIMED((rownorm_kernel<T, 0>), grid, dim3(256), shm, s, a, lpr);
        else LTX_LAUNCH_TIMED((rownorm_kernel<T, 2>), grid, dim3(256), 0, s, a, lpr);
    }
}

}  // namespace

int ltx_launch_rownorm(const RowNormArgs& a, int dtype, hipStream_t s) {
    const int ch = dtype == LTX_DT_BF16 ? 8 : 4;
    if (a.rows <= 0) return LTX_OK;
    if (a.D % ch != 0 || a.ldx % ch != 0 || a.ldy % ch != 0) LTX_FAIL(LTX_ERR_ARG, "rownorm: D/ld must be multiples of the 16-byte chunk");
    if ((a.scale == nullptr) != (a.shift == nullptr)) LTX_FAIL(LTX_ERR_ARG, "rownorm: scale and shift go together");
    const int nch = a.D / ch, lpr = pick_lpr(nch);
    if (a.parts) {
        const int rpw = 64 / lpr;
        const bool narrow = nch <= lpr && a.rows >= (int64_t)rpw * NSLOT * 4 * 64;
        const bool block_rows = a.rows <= 512 && nch > 64 && nch <= 512;          // one row per block (rownorm_block_kernel)
        if (a.presum || narrow || (nch > NSLOT * lpr && !block_rows) || a.nparts < 1 || a.nparts > 16 || !a.x_out || a.part_stride < a.rows * a.D || a.D % 8 ||
            ((uintptr_t)a.parts & 15) || (a.d_gate && (((uintptr_t)a.d_gate & 15) || a.d_gate_stride % 4)))
            LTX_FAIL(LTX_ERR_ARG, "rownorm: deferred rows need whole-row lanes (D of at most 64 x 8 chunks), 1..16 parts of [rows, D] f32 and an output for the finished rows");
    }
    // (every argument check sits in front of ltx_prof_begin: a failure after it would leave the timing record open)
    if (a.presum && (a.kind != 0 || a.presum_n < 4 || a.presum_n % 4 != 0 || nch > 256 || (nch & (nch - 1)) || a.rows >= 2147483647LL || a.rows_per_batch >= 2147483647LL))
        LTX_FAIL(LTX_ERR_ARG, "rownorm: presum serves RMS rows of a power-of-two number (<= 256) of 16-byte chunks with a multiple of 4 partials");
    void* tok = nullptr;
    ltx_prof_begin(LTX_PROF_ROWNORM, 2.0 * (double)a.rows * a.D * (dtype == LTX_DT_BF16 ? 2 : 4), s, &tok);
    if (a.presum) {
        const int rpb = 256 / nch;
        int R = 4;                                           // rows per thread (LTX_NORM_PRESUM_R = 2 / 4 / 8: tuning aid)
        { const int v = ltx_exp("norm_presum_r", 4); if (v == 2 || v == 4 || v == 8) R = v; }
        const dim3 grid((unsigned)cdiv64(a.rows, (int64_t)rpb * R));
        const bool wide = nch >= 64;
        // the DiT's case on the kernel that has nothing to decide (LTX_NORM_LEAN=0: the general one; same bits)
        const bool lean_on = ltx_opt().norm_lean != 0, nt = ltx_exp("norm_nt", 0) != 0;
        const bool lean = lean_on && dtype == LTX_DT_BF16 && wide && R == 4 && a.scale && !a.weight && a.act == 0 && a.presum_n <= 16 &&
                          a.rows % ((int64_t)rpb * 4) == 0 && a.rows_per_batch % 4 == 0 && !nt;
        if (lean) LTX_LAUNCH_TIMED((rownorm_presum_lean_kernel<4>), grid, dim3(256), 0, s, a, nch, rpb);
        else if (dtype == LTX_DT_BF16) {
            if (R == 2) { if (wide) LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 2, true>), grid, dim3(256), 0, s, a, nch, rpb); else LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 2, false>), grid, dim3(256), 0, s, a, nch, rpb); }
            else if (R == 4 && wide && nt) LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 4, true, true>), grid, dim3(256), 0, s, a, nch, rpb);
            else if (R == 4) { if (wide) LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 4, true>), grid, dim3(256), 0, s, a, nch, rpb); else LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 4, false>), grid, dim3(256), 0, s, a, nch, rpb); }
            else { if (wide) LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 8, true>), grid, dim3(256), 0, s, a, nch, rpb); else LTX_LAUNCH_TIMED((rownorm_presum_kernel<bf16_t, 8, false>), grid, dim3(256), 0, s, a, nch, rpb); }
        } else if (wide) LTX_LAUNCH_TIMED((rownorm_presum_kernel<float, 4, true>), dim3((unsigned)cdiv64(a.rows, (int64_t)rpb * 4)), dim3(256), 0, s, a, nch, rpb);
        else LTX_LAUNCH_TIMED((rownorm_presum_kernel<float, 4, false>), dim3((unsigned)cdiv64(a.rows, (int64_t)rpb * 4)), dim3(256), 0, s, a, nch, rpb);
    } else if (dtype == LTX_DT_BF16) launch_rownorm_t<bf16_t>(a, lpr, nch, s);
    else launch_rownorm_t<float>(a, lpr, nch, s);
    ltx_prof_end(tok, s);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

namespace {
// Few rows (C1's 384 tokens; the text rows of the cached cross-attention keys): one row per block, a chunk (two beyond 2048 columns)
// per thread and segment - the arithmetic of qknorm_rope_fused_kernel, the row statistics summed lane butterfly first, then the four
// wave sums in wave order (see rownorm_block_kernel): 6.6 -> ~4 us per launch at 384 rows.
template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_block_kernel(const QkNormRopeArgs a) {
    constexpr int CH = ElemTraits<T>::CHUNK, NC = 2;
    static_assert(CH == 8, "bf16 rows only");
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int nch = a.D / CH;
    const bool rope = a.cos != nullptr;
    const float* cs = rope ? a.cos + row * (a.D / 2) : nullptr;
    const float* sn = rope ? a.sin + row * (a.D / 2) : nullptr;
    T* x0 = reinterpret_cast<T*>(a.x) + row * a.ld;
    const int64_t sst = a.seg_stride ? a.seg_stride : a.D;
    Chunk16 v[2][NC];
    f32x4 co[NC], si[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        const bool in = c < nch;
        v[0][i].u = (u32x4){0u, 0u, 0u, 0u}; v[1][i].u = (u32x4){0u, 0u, 0u, 0u};
        if (in) v[0][i].u = *reinterpret_cast<const u32x4*>(x0 + c * CH);
        if (in && a.nseg == 2) v[1][i].u = *reinterpret_cast<const u32x4*>(x0 + sst + c * CH);
        if (in && rope) { co[i] = *reinterpret_cast<const f32x4*>(cs + c * 4); si[i] = *reinterpret_cast<const f32x4*>(sn + c * 4); }
    }
    float ss[2] = {0.f, 0.f};
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
#pragma unroll
        for (int i = 0; i < NC; ++i) { float f[CH]; chunk_to_f32<T>(v[seg][i], f);
#pragma unroll
            for (int j = 0; j < CH; ++j) ss[seg] += f[j] * f[j]; }
        ss[seg] = group_sum(ss[seg], 64);
        if (lane == 0) red[seg][wave] = ss[seg];
    }
    __syncthreads();
    float rinv[2];
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        const float t = ((red[seg][0] + red[seg][1]) + red[seg][2]) + red[seg][3];
        rinv[seg] = (seg == 0 ? a.out_scale0 : 1.0f) / sqrtf(t / (float)a.D + a.eps);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = tid + i * 256;
        if (c >= nch) continue;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            if (seg >= a.nseg) break;
            const T* w = reinterpret_cast<const T*>(seg == 0 ? a.w0 : a.w1);
            float f[CH]; chunk_to_f32<T>(v[seg][i], f);
            Chunk16 wc; wc.u = *reinterpret_cast<const u32x4*>(w + c * CH);
            float wv[CH]; chunk_to_f32<T>(wc, wv);
#pragma unroll
            for (int j = 0; j < CH; ++j) f[j] = f[j] * rinv[seg] * wv[j];
            if (seg == 0 && a.w0b) {
                Chunk16 wc2; wc2.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.w0b) + c * CH);
                float wv2[CH]; chunk_to_f32<T>(wc2, wv2);
#pragma unroll
                for (int j = 0; j < CH; ++j) f[j] *= wv2[j];
            }
            if (rope) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float re = f[2 * p], im = f[2 * p + 1];
                    f[2 * p] = re * co[i][p] - im * si[i][p];
                    f[2 * p + 1] = im * co[i][p] + re * si[i][p];
                }
            }
            Chunk16 o; f32_to_chunk<T>(f, o);
            *reinterpret_cast<u32x4*>(x0 + seg * sst + c * CH) = o.u;
        }
    }
}
}  // namespace

int ltx_launch_qknorm_rope(const QkNormRopeArgs& a, int dtype, hipStream_t s) {
    const int ch = dtype == LTX_DT_BF16 ? 8 : 4;
    if (a.rows <= 0) return LTX_OK;
    if (a.D % ch != 0 || a.ld % ch != 0) LTX_FAIL(LTX_ERR_ARG, "qknorm: D/ld must be multiples of the 16-byte chunk");
    if (a.nseg < 1 || a.nseg > 2 || !a.w0 || (a.nseg == 2 && !a.w1)) LTX_FAIL(LTX_ERR_ARG, "qknorm: bad segments/weights");
    const int nch = a.D / ch, lpr = pick_lpr(nch);
    const int64_t rows_per_block = 4 * (64 / lpr);
    dim3 grid((unsigned)cdiv64(a.rows, rows_per_block)), block(256);
    const bool cached = nch <= NSLOT * lpr;
    if (dtype == LTX_DT_BF16) {
        if (a.rows <= 512 && nch > 64 && nch <= 512) hipLaunchKernelGGL((qknorm_rope_block_kernel<bf16_t>), dim3((unsigned)a.rows), block, 0, s, a);      // few wide rows: one row per block
        else if (nch <= 4 * lpr) hipLaunchKernelGGL((qknorm_rope_fused_kernel<bf16_t, 4>), grid, block, 0, s, a, lpr);
        // the 13B model's 4096-wide rows: eight chunks per lane, the table still read once for q and k (LTX_QKNORM_FUSED8=0: the two-pass kernel)
        else if (nch <= 8 * lpr && ltx_exp("qknorm_fused8", 1)) hipLaunchKernelGGL((qknorm_rope_fused_kernel<bf16_t, 8, 2>), grid, block, 0, s, a, lpr);
        else if (cached) hipLaunchKernelGGL((qknorm_rope_kernel<bf16_t, true>), grid, block, 0, s, a, lpr);
        else hipLaunchKernelGGL((qknorm_rope_kernel<bf16_t, false>), grid, block, 0, s, a, lpr);
    } else {
        if (cached) hipLaunchKernelGGL((qknorm_rope_kernel<float, true>), grid, block, 0, s, a, lpr);
        else hipLaunchKernelGGL((qknorm_rope_kernel<float, false>), grid, block, 0, s, a, lpr);
    }
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

int ltx_launch_rowsq(const void* x, int dtype, int64_t rows, int N, int ld, float* out, hipStream_t s) {
    if (rows <= 0 || N <= 0) return LTX_OK;
    if (!x || !out || N % 4 != 0 || ld % 4 != 0 || ((uintptr_t)x & 7)) LTX_FAIL(LTX_ERR_ARG, "rowsq: N and ld must be multiples of 4 and x 8-byte aligned");
    const int64_t npairs = rows * ((N + 127) / 128);
    dim3 grid((unsigned)cdiv64(npairs, 256)), block(256);
    if (dtype == LTX_DT_BF16) hipLaunchKernelGGL(rowsq_kernel<bf16_t>, grid, block, 0, s, reinterpret_cast<const bf16_t*>(x), rows, N, ld, out);
    else hipLaunchKernelGGL(rowsq_kernel<float>, grid, block, 0, s, reinterpret_cast<const float*>(x), rows, N, ld, out);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

// cvec[b][n] = sum_k shift[b][k] * W[n][k] + bias[n] (norm fold, kernels.h): one wave per output column, lanes over 16-byte
// chunks of the weight row, f32 throughout, a fixed order of additions (per lane ascending k, then the lanes' sums by halving).
// Reads W once: a step's 2 x num_layers launches stream the q|k|v and ff1 weights at the HBM rate - once per distinct timestep.
namespace {
__global__ __launch_bounds__(256) void shift_gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias, const float* __restrict__ shift, int shift_stride,
                                                          int B, int N, int K, float* __restrict__ out, int out_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* wr = W + (int64_t)n * K;
    for (int c = lane; c < K / 8; c += 64) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wr + c * 8);
        float wf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wf[i] = (float)w[i];
        for (int b = 0; b < B; ++b) {
            const float* sp = shift + (int64_t)b * shift_stride + c * 8;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
            float a = acc[b];
#pragma unroll
            for (int i = 0; i < 4; ++i) a = __builtin_fmaf(wf[i], s0[i], a);
#pragma unroll
            for (int i = 0; i < 4; ++i) a = __builtin_fmaf(wf[4 + i], s1[i], a);
            acc[b] = a;
        }
    }
    const float bn = bias ? (float)bias[n] : 0.f;
    for (int b = 0; b < B; ++b) {
        float a = acc[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) out[(int64_t)b * out_stride + n] = a + bn;
    }
}
}  // namespace

int ltx_launch_shift_gemv(const void* W, const void* bias, const float* shift, int shift_stride, int B, int N, int K, float* cvec, int cvec_stride, hipStream_t s) {
    if (!W || !shift || !cvec || B < 1 || B > 8 || N < 1 || K < 8 || K % 8 || shift_stride % 4 || ((uintptr_t)W & 15) || ((uintptr_t)shift & 15))
        LTX_FAIL(LTX_ERR_ARG, "shift_gemv: bf16 weights [N, K] with K % 8 == 0, f32 shift rows 16-byte aligned, 1..8 batch rows");
    hipLaunchKernelGGL(shift_gemv_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(W), reinterpret_cast<const bf16_t*>(bias),
                       shift, shift_stride, B, N, K, cvec, cvec_stride);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}

int ltx_launch_rope_table(const RopeTableArgs& a, hipStream_t s) {
    if (a.D % 2 != 0 || a.D < 6) LTX_FAIL(LTX_ERR_ARG, "rope: dim must be even and >= 6");
    const int64_t total = (int64_t)a.B * a.F * a.H * a.W * (a.D / 2);
    int64_t blocks = cdiv64(total, 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    LTX_CHECK_LAUNCH();
    return LTX_OK;
}
