#!/usr/bin/env python3
"""Generates candle-video_amd/csrc/attn_q64_loop.inc: the hand-scheduled main loop of the 64-queries-per-wave
self-attention kernel (csrc/attn_q64.hip) as ONE inline-asm statement per variant, with every register named.

Why generated asm: the loop needs ~420 registers split by hand between the two halves of the unified file (S^T and P^T
in arch VGPRs for the VALU, O^T / Q^T / K / V^T fragments in AGPRs) and a fixed interleave of two v_exp + one v_cvt_pk
per MFMA; hipcc's allocator answers the same source with ~140 v_accvgpr copies per tile.

Register map (QB = 2, two 32-query column blocks per wave; QB = 1 uses the low half of each area):
  v[0:63]    S^T set X  (q0k0 | q0k1 | q1k0 | q1k1, 16 each)        a[0:63]    O^T  (q0d0 | q0d1 | q1d0 | q1d1)
  v[64:127]  S^T set Y                                               a[64:95]   Q^T fragments (q0 ks0..3 | q1 ks0..3)
  v[128:159] P^T operands ((qb, kb, s) -> 4 regs)                    a[96:127]  K fragments   (kb*4 + ks)
  v[160:191] -m tuples (q0 | q1)                                     a[128:159] V^T fragments (d*4 + j)
  v[192:199] row-sum accumulators (q0 | q1)                          a[160:163] 0/1 operand of the row-sum MFMA
  v[200:207] exp temporaries
  v[208:211] K read bases (ks), v[212:213] V^T read bases (d), v[214:215] K DMA offsets, v[216:217] V DMA offsets
QB = 1: S^T X = v[0:31] (k0 | k1), Y = v[32:63]; P^T sets v[128:143] / v[144:159]; -m v[160:175]; l v[192:195];
  O^T a[0:31]; Q^T a[64:79].

Iteration t (ring slot s = t & 3, CUR = S^T(t) - m, NXT receives S^T(t+1)), QB = 2:
  G1  8 x { MFMA S^T(t+1) q0 ; 2 exp + cvt of (CUR q0 k1)   ; one V^T(t) transpose read }
  G2  8 x { MFMA S^T(t+1) q1 ; 2 exp + cvt of (CUR q1 k0)   ; one V^T(t) transpose read }
  vmcnt(4) [tile t+2 landed], lgkmcnt(0), s_barrier
  G3  8 x { MFMA O^T q0 += V^T P^T ; 2 exp + cvt of (CUR q1 k1) ; K(t+2) fragment read ; row-sum MFMA (gaps 4..7) }
  G4  8 x { MFMA O^T q1            ; 2 exp + cvt of (NXT q0 k0) ; DMA piece of tile t+4 (even gaps) ; row-sum MFMA }
  lgkmcnt(0)
The loop is unrolled by four (static ring slots) and leaves after any iteration when the count runs out; tiles past
the last one are fetched out of the buffers' range (zeros, no memory traffic) and multiplied into the unused set."""
import os
import sys

TILE = 16384
KROW = VROW = 128
VBASE = 64 * KROW


def gen(QB):
    L = []
    emit = L.append
    SW = 32 * QB                      # registers per S^T set
    def S(st, qb, kb):                # first register of S^T block
        return st * SW + (qb * 2 + kb) * 16
    def P(ps, qb, kb, s):             # QB = 2: one set; QB = 1: two sets of 16
        if QB == 2:
            return 128 + ((qb * 2 + kb) * 2 + s) * 4
        return 128 + ps * 16 + (kb * 2 + s) * 4
    def MINIT(qb): return 160 + 16 * qb
    def LACC(qb): return 192 + 4 * qb
    def O(qb, d): return (qb * 2 + d) * 16
    def Q(qb, ks): return 64 + (qb * 4 + ks) * 4
    def KF(i): return 96 + 4 * i
    def VF(n): return 128 + 4 * n
    ONES = 160
    tmp = [0]
    def temps():
        a = 200 + 2 * (tmp[0] % 4); tmp[0] += 1
        return a, a + 1

    def vr(a, n): return f"v[{a}:{a + n - 1}]"
    def ar(a, n): return f"a[{a}:{a + n - 1}]"

    def exp_slice(sbase, pbase2, i, filler):
        """registers 2i, 2i+1 of the S^T block at sbase -> word (i & 3) of P operand s = i >> 2"""
        t0, t1 = temps()
        emit(f"v_exp_f32_e32 v{t0}, v{sbase + 2 * i}")
        emit(f"v_exp_f32_e32 v{t1}, v{sbase + 2 * i + 1}")
        for ins in filler: emit(ins)
        if not filler: emit("s_nop 0")                       # trans result -> VALU read needs one state
        emit(f"v_cvt_pk_bf16_f32 v{pbase2[i >> 2] + (i & 3)}, v{t0}, v{t1}")

    def body(slot, cur):
        nxt = cur ^ 1
        pc, pn = (cur, nxt) if QB == 1 else (0, 0)
        kslot = (slot + 2) & 3
        # ---- QK groups
        for qb in range(QB):
            for i in range(8):
                kb, ks = i >> 2, i & 3
                d = vr(S(nxt, qb, kb), 16)
                c = vr(MINIT(qb), 16) if ks == 0 else d
                emit(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(i), 4)}, {ar(Q(qb, ks), 4)}, {c}")
                if QB == 2:
                    sb = S(cur, 0, 1) if qb == 0 else S(cur, 1, 0)
                    pb = [P(pc, 0, 1, 0), P(pc, 0, 1, 1)] if qb == 0 else [P(pc, 1, 0, 0), P(pc, 1, 0, 1)]
                    halves = [qb * 8 + i]
                else:
                    sb = S(cur, 0, 1); pb = [P(pc, 0, 1, 0), P(pc, 0, 1, 1)]
                    halves = [2 * i, 2 * i + 1]
                fill = []
                for hidx in halves:
                    n, hf = hidx >> 1, hidx & 1
                    dd, j = n >> 2, n & 3
                    rowc = (j >> 1) * 32 + (j & 1) * 16 + 8 * hf
                    imm = slot * TILE + VBASE + rowc * VROW
                    fill.append(f"ds_read_b64_tr_b16 {ar(VF(n) + 2 * hf, 2)}, v{212 + dd} offset:{imm}")
                exp_slice(sb, pb, i, fill)
        # ---- tile t+2 landed for every wave, V^T(t) fragments in
        emit("s_waitcnt vmcnt(4)")
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
        # ---- PV groups
        for qb in range(QB):
            for i in range(8):
                d, j = i >> 2, i & 3
                o = ar(O(qb, d), 16)
                emit(f"v_mfma_f32_32x32x16_bf16 {o}, {ar(VF(i), 4)}, {vr(P(pc, qb, j >> 1, j & 1), 4)}, {o}")
                fill = []
                if QB == 2 and qb == 0:
                    sb = S(cur, 1, 1); pb = [P(pc, 1, 1, 0), P(pc, 1, 1, 1)]
                else:
                    sb = S(nxt, 0, 0); pb = [P(pn, 0, 0, 0), P(pn, 0, 0, 1)]
                if qb == 0:                                   # K(t+2) fragment i
                    kb, ks = i >> 2, i & 3
                    fill.append(f"ds_read_b128 {ar(KF(i), 4)}, v{208 + ks} offset:{kslot * TILE + kb * 32 * KROW}")
                if qb == QB - 1 and i % 2 == 0:               # DMA piece i/2 of tile t+4 into the slot of tile t
                    p = i >> 1
                    imm = slot * TILE + (VBASE if p >= 2 else 0) + (p & 1) * 1024
                    pre = f"s_add_u32 m0, %[ldsw], {imm}"
                    if p < 2: ld = f"buffer_load_dwordx4 v{214 + (p & 1)}, %[rk], %[koff] offen lds"
                    else: ld = f"buffer_load_dwordx4 v{216 + (p & 1)}, %[rv], %[voff] offen lds"
                    emit(pre)                                 # M0 written >= 2 instructions ahead of the load
                    fill.append(ld)
                if i >= 4:
                    l = vr(LACC(qb), 4)
                    fill.append(f"v_mfma_f32_16x16x32_bf16 {l}, {ar(ONES, 4)}, {vr(P(pc, qb, (i - 4) >> 1, (i - 4) & 1), 4)}, {l}")
                exp_slice(sb, pb, i, fill)
        emit("s_add_u32 %[koff], %[koff], %[kstep]")
        emit("s_add_u32 %[voff], %[voff], %[vstep]")
        emit("s_waitcnt lgkmcnt(0)")

    emit("s_nop 15")                                        # operands set up by compiler-scheduled VALU / accvgpr writes
    for it in range(4):
        if it == 0: emit("1:")
        body(it, it & 1)
        emit("s_add_i32 %[cnt], %[cnt], -1")
        emit("s_cmp_eq_u32 %[cnt], 0")
        if it < 3: emit("s_cbranch_scc1 2f")
        else: emit("s_cbranch_scc0 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    return L


def c_function(QB):
    lines = gen(QB)
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    if QB == 2:
        sig = ("f32x32& sx0, f32x32& sx1, f32x32& sy0, f32x32& sy1, u32x32& p, const f32x32& minit, f32x8& lacc, "
               "f32x32& o0, f32x32& o1, const u32x32& q, u32x32& kf, ")
        outs = ('"+{v[0:31]}"(sx0), "+{v[32:63]}"(sx1), "+{v[64:95]}"(sy0), "+{v[96:127]}"(sy1), "+{v[128:159]}"(p), "+{v[192:199]}"(lacc), '
                '"+{a[0:31]}"(o0), "+{a[32:63]}"(o1), "+{a[96:127]}"(kf), ')
        ins = '"{v[160:191]}"(minit), "{a[64:95]}"(q), '
    else:
        sig = ("f32x32& sx0, f32x32& sy0, u32x32& p, const f32x16& minit, f32x4& lacc, f32x32& o0, const u32x16& q, u32x32& kf, ")
        outs = ('"+{v[0:31]}"(sx0), "+{v[32:63]}"(sy0), "+{v[128:159]}"(p), "+{v[192:195]}"(lacc), "+{a[0:31]}"(o0), "+{a[96:127]}"(kf), ')
        ins = '"{v[160:175]}"(minit), "{a[64:79]}"(q), '
    clob = ['"v200"', '"v201"', '"v202"', '"v203"', '"v204"', '"v205"', '"v206"', '"v207"'] + [f'"a{i}"' for i in range(128, 160)] + ['"scc"', '"memory"']
    return f"""// GENERATED by tools/gen_attn_q64_asm.py - do not edit.  {len(lines)} instructions, QB = {QB}.
__device__ __forceinline__ void q64_loop_qb{QB}({sig}const u32x4& ones, const u32x4& kbase, const u32x2& trbase, const u32x4& dmaoff,
        const u32x4& rk, const u32x4& rv, int& cnt, uint32_t& koff, uint32_t kstep, uint32_t& voff, uint32_t vstep, uint32_t ldsw) {{
    asm volatile(
{text}        : {outs}[cnt] "+s"(cnt), [koff] "+s"(koff), [voff] "+s"(voff)
        : {ins}"{{a[160:163]}}"(ones), "{{v[208:211]}}"(kbase), "{{v[212:213]}}"(trbase), "{{v[214:217]}}"(dmaoff),
          [rk] "s"(rk), [rv] "s"(rv), [kstep] "s"(kstep), [vstep] "s"(vstep), [ldsw] "s"(ldsw)
        : {", ".join(clob)});
}}
"""


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "candle-video_amd", "csrc", "attn_q64_loop.inc")
    with open(out, "w") as f:
        f.write(c_function(2))
        f.write("\n")
        f.write(c_function(1))
    print("wrote", out)


if __name__ == "__main__":
    main()
