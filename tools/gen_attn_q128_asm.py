#!/usr/bin/env python3
"""Generates candle-video_amd/csrc/attn_q128_loop.inc: prologue + main loop + epilogue of the head_dim-128 self-attention
kernel (csrc/attn_q128.hip, the 13B model: BASELINE C5) as ONE inline-asm statement with every register named.

Same structure as tools/gen_attn_q64_asm.py (one wave per SIMD, software pipeline over 64-key tiles, fixed first-tile max,
row sums on the matrix pipe), 32 queries per wave:  per tile 16 S^T MFMAs (2 key blocks x 8 k-steps over d) + 16 P.V MFMAs
(4 d blocks x 4 key groups) of 32x32x16 - the 32 MFMA gaps of the 256-query form at head_dim 64 - with HALF its exponentials
(16 slices of two), so the loop is bound by the matrix pipe, not by the vector issue port.

Register map:
  v[0:31]    S^T set X (k0 | k1, 16 each)                 a[0:63]     O^T (d0 .. d3)
  v[32:63]   S^T set Y                                     a[64:95]    Q^T fragments (ks 0..7)
  v[128:143] P^T set 0, v[144:159] set 1 ((kb, s) -> 4)    a[96:159]   K fragments   (kb*8 + ks)
  v[160:175] -m tuple                                      a[160:223]  V^T fragments (d*4 + j)
  v176 -inf, v177 4*(lane>>5) (tail mask)                  a[224:227]  0/1 operand of the row-sum MFMA
  v[192:195] row-sum accumulator, v[200:207] exp temporaries
  v[208:215] K read bases (ks), v[216:219] V^T read bases (d)          - ring slots 0, 1 (immediate offsets < 64 KiB)
  v[236:243] K read bases + 64 KiB, v[232:235] V^T read bases + 64 KiB - ring slots 2, 3
  v[220:223] K DMA source offsets (4 pieces), v[224:227] V DMA offsets, v228 Q offset, v229 O offset, v230 l select

Iteration t (ring slot s = t & 3, CUR = S^T(t) - m with its k0 block already exponentiated, NXT receives S^T(t+1)):
  QK 16 x { MFMA S^T(t+1) ; ONE exp of (CUR k1) [+ the previous pair's cvt on even gaps] ; two V^T(t) transpose reads }
  vmcnt(8) [tile t+2 landed], lgkmcnt(0), s_barrier ; when two tiles are left: keys past Sk of S^T(t+1) -> -inf
  PV 16 x { MFMA O^T += V^T P^T ; ONE exp of (NXT k0) [+ cvt] ; K(t+2) fragment read ; row-sum MFMA (gaps 4..7) }
  LDS-DMA of tile t+4 (8 pieces of 1 KiB per wave): K pieces in QK gaps 1, 5, 9, 13, V pieces in PV gaps 1, 5, 9, 13 - a
  piece holds the CU's address path ~16 cycles and the four waves run in step; packed into the PV gaps they cost 30 %
Unrolled by four (static ring slots); leaves after any iteration when the count runs out; tiles past the last one are fetched
out of the buffers' range (zeros) and multiplied into the unused set."""
import os
import sys

TILE = 32768
KROW = VROW = 256
VBASE = 64 * KROW
HI = 65536


def gen(opt=None):
    o_ = dict(abl="", align=1, spread=1)
    o_.update(opt or {})
    SPREAD = int(o_["spread"])      # 1: tile t+4's K pieces in the QK gaps, V pieces in the PV gaps (one per four gaps); 0: all eight in the PV gaps
    abl = set(x for x in str(o_["abl"]).split("+") if x)
    L = []
    def emit(ins):
        op = ins.split()[0]
        if "nodma" in abl and (op.startswith("buffer_load") and " lds" in ins or ins.startswith("s_add_u32 m0")): return
        if "noexp" in abl and op.startswith("v_exp"): ins = ins.replace("v_exp_f32_e32", "v_mov_b32_e32")
        if "nobar" in abl and op == "s_barrier": return
        L.append(ins)
    def S(st, kb): return st * 32 + kb * 16
    def P(ps, kb, s): return 128 + ps * 16 + (kb * 2 + s) * 4
    MINIT, LACC, NEGINF, KEY0 = 160, 192, 176, 177
    def O(d): return d * 16
    def Q(ks): return 64 + 4 * ks
    def KF(i): return 96 + 4 * i
    def VF(n): return 160 + 4 * n
    ONES = 224
    tmp = [0]
    def temps():
        a = 200 + 2 * (tmp[0] % 4); tmp[0] += 1
        return a, a + 1
    def vr(a, n): return f"v[{a}:{a + n - 1}]"
    def ar(a, n): return f"a[{a}:{a + n - 1}]"
    def kbase(ks, slot): return (208 if slot < 2 else 236) + ks
    def vbase(d, slot): return (216 if slot < 2 else 232) + d

    pending = [None]
    cur_t = [None]
    def exp_half(sbase, pbase2, e, filler):
        """ONE exponential per MFMA gap (a transcendental holds the vector issue port for 16 cycles: two of them plus a convert
        and two LDS reads overran the 32-cycle MFMA of their gap): element e of the S^T block at sbase; the pair (2i, 2i+1)
        becomes word (i & 3) of P operand s = i >> 2 by a convert emitted with the NEXT pair's first exponential"""
        i = e >> 1
        if e & 1 == 0:
            cur_t[0] = temps()
            if pending[0]: emit("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % pending[0]); pending[0] = None
            emit(f"v_exp_f32_e32 v{cur_t[0][0]}, v{sbase + e}")
        else:
            emit(f"v_exp_f32_e32 v{cur_t[0][1]}, v{sbase + e}")
            pending[0] = (pbase2[i >> 2] + (i & 3), cur_t[0][0], cur_t[0][1])
        for ins in filler: emit(ins)

    def mask_tail(st):
        """keys past Sk of the tile in S^T set st (register i of key block kb = key kb*32 + (i&3) + 8*(i>>2) + 4h): -inf"""
        for kb in range(2):
            for i in range(16):
                c = kb * 32 + (i & 3) + 8 * (i >> 2)
                emit(f"s_sub_i32 %[stmp], %[rem], {c}")
                emit(f"v_cmp_le_i32_e32 vcc, %[stmp], v{KEY0}")
                emit(f"v_cndmask_b32_e32 v{S(st, kb) + i}, v{S(st, kb) + i}, v{NEGINF}, vcc")

    label = [10]
    def body(slot, cur):
        nxt = cur ^ 1
        kslot = (slot + 2) & 3
        for gp in range(32):
            if gp == 16:                                      # tile t+2 landed for every wave, V^T(t) fragments in
                emit(f"s_waitcnt vmcnt({12 if SPREAD else 8})")      # spread: the four K pieces of tile t+4 are already out
                emit("s_waitcnt lgkmcnt(0)")
                emit("s_barrier")
                lb = label[0]; label[0] += 1
                emit("s_cmp_eq_u32 %[cnt], 2")               # tile t+1 is the last one: mask its keys past Sk
                emit(f"s_cbranch_scc0 {lb}f")
                emit("s_nop 15")                              # the last S^T MFMA of the QK gaps above has left the pipe
                mask_tail(nxt)
                emit(f"{lb}:")
            pre, fill = [], []
            if gp < 16:
                kb, ks = gp >> 3, gp & 7
                d = vr(S(nxt, kb), 16)
                c = vr(MINIT, 16) if ks == 0 else d
                main = f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(gp), 4)}, {ar(Q(ks), 4)}, {c}"
                sb, pb = S(cur, 1), [P(cur, 1, 0), P(cur, 1, 1)]
                for hidx in (2 * gp, 2 * gp + 1):             # V^T(t) transpose reads: operand n = d*4 + j, half hf
                    n, hf = hidx >> 1, hidx & 1
                    d2, j2 = n >> 2, n & 3
                    rowc = (j2 >> 1) * 32 + (j2 & 1) * 16 + 8 * hf
                    fill.append(f"ds_read_b64_tr_b16 {ar(VF(n) + 2 * hf, 2)}, v{vbase(d2, slot)} offset:{(slot & 1) * TILE + VBASE + rowc * VROW}")
                if SPREAD and gp % 4 == 1:                    # K piece p of tile t+4 (the K half of this slot was last read two iterations ago);
                    p = gp // 4                               # M0 is written two instructions ahead of the load
                    pre.append(f"s_add_u32 m0, %[ldsw], {slot * TILE + p * 1024}")
                    fill.append(f"buffer_load_dwordx4 v{220 + p}, %[rk], %[koff] offen lds")
            else:
                i = gp - 16
                dd, j = i >> 2, i & 3
                o = ar(O(dd), 16)
                main = f"v_mfma_f32_32x32x16_bf16 {o}, {ar(VF(i), 4)}, {vr(P(cur, j >> 1, j & 1), 4)}, {o}"
                sb, pb = S(nxt, 0), [P(nxt, 0, 0), P(nxt, 0, 1)]
                fill.append(f"ds_read_b128 {ar(KF(i), 4)}, v{kbase(i & 7, kslot)} offset:{(kslot & 1) * TILE + (i >> 3) * 32 * KROW}")
                dma_v = None
                if SPREAD:
                    if i % 4 == 1:                            # V piece p of tile t+4 (V^T(t) was read before the barrier above)
                        p = i // 4
                        pre.append(f"s_add_u32 m0, %[ldsw], {slot * TILE + VBASE + p * 1024}")
                        dma_v = f"buffer_load_dwordx4 v{224 + p}, %[rv], %[voff] offen lds"
                elif i % 2 == 0:                              # all eight pieces of tile t+4 in the PV gaps
                    p = i // 2
                    imm = slot * TILE + (VBASE if p >= 4 else 0) + (p & 3) * 1024
                    pre.append(f"s_add_u32 m0, %[ldsw], {imm}")
                    if p < 4: fill.append(f"buffer_load_dwordx4 v{220 + (p & 3)}, %[rk], %[koff] offen lds")
                    else: fill.append(f"buffer_load_dwordx4 v{224 + (p & 3)}, %[rv], %[voff] offen lds")
                if 4 <= i < 8:                                # row-sum MFMA ri: operand (kb, s) = ri
                    ro = i - 4
                    l = vr(LACC, 4)
                    fill.append(f"v_mfma_f32_16x16x32_bf16 {l}, {ar(ONES, 4)}, {vr(P(cur, ro >> 1, ro & 1), 4)}, {l}")
                if dma_v:
                    if len(fill) < 2: fill.append("s_nop 0")
                    fill.append(dma_v)
            emit(main)
            for ins in pre: emit(ins)
            exp_half(sb, pb, gp & 15, fill)
        if "dmaoob" not in abl:
            emit("s_add_u32 %[koff], %[koff], %[kstep]")
            emit("s_add_u32 %[voff], %[voff], %[vstep]")
        if pending[0]:
            emit("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % pending[0]); pending[0] = None
        emit("s_waitcnt lgkmcnt(0)")

    def prologue():
        for ks in range(8):
            off = f" offset:{32 * ks}" if ks else ""
            emit(f"buffer_load_dwordx4 {ar(Q(ks), 4)}, v228, %[rq], 0 offen{off}")
        for tl in range(4):
            for p in range(8):
                imm = tl * TILE + (VBASE if p >= 4 else 0) + (p & 3) * 1024
                emit(f"s_add_u32 m0, %[ldsw], {imm}")
                emit("s_nop 0")
                if p < 4: emit(f"buffer_load_dwordx4 v{220 + (p & 3)}, %[rk], %[koff] offen lds")
                else: emit(f"buffer_load_dwordx4 v{224 + (p & 3)}, %[rv], %[voff] offen lds")
            emit("s_add_u32 %[koff], %[koff], %[kstep]")
            emit("s_add_u32 %[voff], %[voff], %[vstep]")
        emit(f"v_mov_b32_e32 v{NEGINF}, 0xff800000")
        emit("s_waitcnt vmcnt(16)")                          # Q^T and tiles 0, 1 landed; tiles 2, 3 stay in flight
        emit("s_barrier")
        for i in range(16):
            emit(f"ds_read_b128 {ar(KF(i), 4)}, v{kbase(i & 7, 0)} offset:{(i >> 3) * 32 * KROW}")
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(16):
            kb, ks = i >> 3, i & 7
            d = vr(S(0, kb), 16)
            emit(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(i), 4)}, {ar(Q(ks), 4)}, {'0' if ks == 0 else d}")
        for i in range(16):                                  # K(1) fragments (the MFMAs above have read theirs long before these land)
            emit(f"ds_read_b128 {ar(KF(i), 4)}, v{kbase(i & 7, 1)} offset:{TILE + (i >> 3) * 32 * KROW}")
        for i in range(64): emit(f"v_accvgpr_write_b32 a{i}, 0")
        for i in range(4): emit(f"v_mov_b32_e32 v{LACC + i}, 0")
        s0, m = S(0, 0), 200                                 # row maximum: 32 scores in the lane, then the other lane half
        emit(f"v_max3_f32 v{m}, v{s0}, v{s0 + 1}, v{s0 + 2}")
        for i in range(3, 31, 2): emit(f"v_max3_f32 v{m}, v{m}, v{s0 + i}, v{s0 + i + 1}")
        emit(f"v_max_f32_e32 v{m}, v{m}, v{s0 + 31}")
        emit(f"v_mov_b32_e32 v202, v{m}")
        emit("s_nop 1")
        emit("v_permlane32_swap_b32_e32 v200, v202")
        emit("v_max_f32_e32 v200, v200, v202")
        for i in range(16): emit(f"v_xor_b32_e32 v{MINIT + i}, 0x80000000, v200")
        for i in range(32): emit(f"v_sub_f32_e32 v{s0 + i}, v{s0 + i}, v200")
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(8):                                   # first exp slices: (tile 0, k0) -> P^T set 0
            t0, t1 = temps()
            emit(f"v_exp_f32_e32 v{t0}, v{S(0, 0) + 2 * i}")
            emit(f"v_exp_f32_e32 v{t1}, v{S(0, 0) + 2 * i + 1}")
            emit("s_nop 0")
            emit(f"v_cvt_pk_bf16_f32 v{P(0, 0, i >> 2) + (i & 3)}, v{t0}, v{t1}")
        tmp[0] = 0

    def epilogue():
        """O^T / l -> bf16 rows, 16-byte stores (lanes l, l^32 exchange column groups, attention.hip store_o_wide)"""
        emit("v_cmp_ne_u32_e32 vcc, 0, v230")
        emit(f"v_cndmask_b32_e32 v204, v{LACC}, v{LACC + 1}, vcc")
        emit("v_rcp_f32_e32 v206, v204")
        R = 0
        for d in range(4):
            for k in range(2):
                base = R; R = (R + 16) % 96
                for e in range(8): emit(f"v_accvgpr_read_b32 v{base + e}, a{O(d) + 8 * k + e}")
                for e in range(8): emit(f"v_mul_f32_e32 v{base + e}, v206, v{base + e}")
                for w in range(4): emit(f"v_cvt_pk_bf16_f32 v{base + 8 + w}, v{base + 2 * w}, v{base + 2 * w + 1}")
                emit("s_nop 1")
                emit(f"v_permlane32_swap_b32_e32 v{base + 8}, v{base + 10}")
                emit(f"v_permlane32_swap_b32_e32 v{base + 9}, v{base + 11}")
                emit(f"buffer_store_dwordx4 {vr(base + 8, 4)}, v229, %[ro], 0 offen offset:{d * 64 + 32 * k}")

    emit("s_nop 15")
    prologue()
    if "dmaoob" in abl:                                      # timing ablation: every in-loop piece out of range (issued, zero-filled, no memory traffic)
        emit("s_mov_b32 %[koff], 0x80000000"); emit("s_mov_b32 %[voff], 0x80000000")
    for it in range(4):
        if int(o_["align"]): emit(".p2align 6" if it == 0 else ".p2align 3")
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
    epilogue()
    return L


def c_function(opt=None):
    lines = gen(opt)
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    used_v = list(range(0, 64)) + list(range(128, 177)) + list(range(196, 208))
    clob = [f'"v{i}"' for i in used_v] + [f'"a{i}"' for i in range(0, 224)] + ['"vcc"', '"scc"', '"memory"']
    return f"""// GENERATED by tools/gen_attn_q128_asm.py - do not edit.  {len(lines)} instructions: head_dim 128, 32 queries per wave, prologue + loop + epilogue.
__device__ __forceinline__ void q128_full(f32x4& lacc, const u32x4& ones, const u32x8& kbase, const u32x4& trbase, const u32x8& kbase_hi, const u32x4& trbase_hi,
        const u32x8& dmaoff, uint32_t qoff, uint32_t ooff, uint32_t sel, uint32_t key0, const u32x4& rk, const u32x4& rv, const u32x4& rq, const u32x4& ro,
        int cnt, int rem, uint32_t kstep, uint32_t vstep, uint32_t ldsw) {{
    uint32_t koff = 0, voff = 0;
    int stmp;
    asm volatile(
{text}        : "={{v[192:195]}}"(lacc), [cnt] "+s"(cnt), [koff] "+s"(koff), [voff] "+s"(voff), [stmp] "=&s"(stmp)
        : "{{a[224:227]}}"(ones), "{{v[208:215]}}"(kbase), "{{v[216:219]}}"(trbase), "{{v[236:243]}}"(kbase_hi), "{{v[232:235]}}"(trbase_hi),
          "{{v[220:227]}}"(dmaoff), "{{v228}}"(qoff), "{{v229}}"(ooff), "{{v230}}"(sel), "{{v177}}"(key0),
          [rk] "s"(rk), [rv] "s"(rv), [rq] "s"(rq), [ro] "s"(ro), [rem] "s"(rem), [kstep] "s"(kstep), [vstep] "s"(vstep), [ldsw] "s"(ldsw)
        : {", ".join(clob)});
}}
"""


# ======================================================================================================================
# 64 queries per wave (256 per workgroup): half the K/V bytes per FLOP of the form above, which measured bound by them
# (every LDS-DMA piece out of range: +28 %).  Two 32-query blocks per wave share every K and V^T fragment, so the fragments
# are STREAMED through four-deep register windows (read once, used by two back-to-back MFMAs) instead of being held whole.
#
# Register map:
#   v[0:63]    S^T set X ((qb, kb) -> 16)      a[0:127]    O^T ((qb, d) -> 16)
#   v[64:127]  S^T set Y                        a[128:191]  Q^T fragments ((qb, ks) -> 4)
#   v[128:159] P^T operands ((qb, kb, s) -> 4)  a[192:207]  K window (4 fragments), a[208:223] V^T window (4 operands)
#   v[160:191] -m tuples (q0 | q1)              a[224:227]  0/1 operand of the row-sum MFMA
#   v[192:199] row sums (q0 | q1), v[200:207] exp temporaries
#   v[208:215] / v[216:223] K read bases (slots 0,1 / 2,3), v[224:227] / v[228:231] V^T read bases
#   v[232:239] DMA source offsets (K 4 | V 4), v[240:241] Q offsets, v[242:243] O offsets, v244 l select, v245 4*(lane>>5), v246 -inf
#
# Iteration t (slot s = t & 3; CUR = S^T(t) - m with (q0, k0) already exponentiated; NXT receives S^T(t+1)):
#   QK 32 gaps, fragment f = kb*8 + ks of K(t+1), two gaps each: { MFMA S^T(t+1) q0 | q1 ; one exp of (CUR q0 k1) [f < 8] or
#      (CUR q1 k0) [f >= 8] ; K read of fragment f+3 (even gaps) ; K piece of tile t+3 (4 of the gaps) }
#   last QK gaps also start the V^T(t) window; vmcnt(4) [tile t+2 whole], lgkmcnt(0), s_barrier, tail mask when two tiles are left
#   PV 32 gaps, operand n = j*4 + d (key group j major), two gaps each: { MFMA O^T q0 | q1 ; one exp of (CUR q1 k1) [first 16]
#      or (NXT q0 k0) [last 16] ; two transpose reads of operand n+3 (even gaps) ; V piece of tile t+3 ; row-sum MFMAs }
#   the last PV gaps start the K(t+2) window of the next iteration.
def gen2(opt=None):
    o_ = dict(abl="", align=1)
    o_.update(opt or {})
    abl = set(x for x in str(o_["abl"]).split("+") if x)
    L = []
    def emit(ins):
        op = ins.split()[0]
        if "nodma" in abl and (op.startswith("buffer_load") and " lds" in ins or ins.startswith("s_add_u32 m0")): return
        if "noexp" in abl and op.startswith("v_exp"): ins = ins.replace("v_exp_f32_e32", "v_mov_b32_e32")
        if "nobar" in abl and op == "s_barrier": return
        L.append(ins)
    def S(st, qb, kb): return st * 64 + (qb * 2 + kb) * 16
    def P(qb, kb, s): return 128 + ((qb * 2 + kb) * 2 + s) * 4
    def MINIT(qb): return 160 + 16 * qb
    def LACC(qb): return 192 + 4 * qb
    NEGINF, KEY0 = 246, 245
    def O(qb, d): return (qb * 4 + d) * 16
    def Q(qb, ks): return 128 + (qb * 8 + ks) * 4
    def KW(f): return 192 + 4 * (f & 3)
    def VW(n): return 208 + 4 * (n & 3)
    ONES = 224
    tmp = [0]
    def temps():
        a = 200 + 2 * (tmp[0] % 4); tmp[0] += 1
        return a, a + 1
    def vr(a, n): return f"v[{a}:{a + n - 1}]"
    def ar(a, n): return f"a[{a}:{a + n - 1}]"
    def kbase(ks, slot): return (208 if slot < 2 else 216) + ks
    def vbase(d, slot): return (224 if slot < 2 else 228) + d
    def kread(f, slot):                                       # K fragment f = kb*8 + ks of the tile in ring slot `slot`
        return f"ds_read_b128 {ar(KW(f), 4)}, v{kbase(f & 7, slot)} offset:{(slot & 1) * TILE + (f >> 3) * 32 * KROW}"
    def vreads(n, slot):                                      # V^T operand n = j*4 + d: two transpose reads
        j, d = n >> 2, n & 3
        out = []
        for hf in range(2):
            rowc = (j >> 1) * 32 + (j & 1) * 16 + 8 * hf
            out.append(f"ds_read_b64_tr_b16 {ar(VW(n) + 2 * hf, 2)}, v{vbase(d, slot)} offset:{(slot & 1) * TILE + VBASE + rowc * VROW}")
        return out

    pending = [None]
    cur_t = [None]
    def exp_half(sbase, pbase2, e):
        i = e >> 1
        if e & 1 == 0:
            cur_t[0] = temps()
            if pending[0]: emit("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % pending[0]); pending[0] = None
            emit(f"v_exp_f32_e32 v{cur_t[0][0]}, v{sbase + e}")
        else:
            emit(f"v_exp_f32_e32 v{cur_t[0][1]}, v{sbase + e}")
            pending[0] = (pbase2[i >> 2] + (i & 3), cur_t[0][0], cur_t[0][1])

    def mask_tail(st):
        for qb in range(2):
            for kb in range(2):
                for i in range(16):
                    c = kb * 32 + (i & 3) + 8 * (i >> 2)
                    emit(f"s_sub_i32 %[stmp], %[rem], {c}")
                    emit(f"v_cmp_le_i32_e32 vcc, %[stmp], v{KEY0}")
                    emit(f"v_cndmask_b32_e32 v{S(st, qb, kb) + i}, v{S(st, qb, kb) + i}, v{NEGINF}, vcc")

    def rowsum(rq, ro):                                        # row sums of tile t on the matrix pipe: P operand (kb, s) = ro of query block rq
        l = vr(LACC(rq), 4)
        emit(f"v_mfma_f32_16x16x32_bf16 {l}, {ar(ONES, 4)}, {vr(P(rq, ro >> 1, ro & 1), 4)}, {l}")

    label = [20]
    def body(slot, cur):
        nxt = cur ^ 1
        ks1 = (slot + 1) & 3                                   # K(t+1)
        ks2 = (slot + 2) & 3                                   # K(t+2): the next iteration's window starts here
        ds = (slot + 3) & 3                                    # tile t+3 goes where tile t-1 was
        # ---- QK phase: 16 fragments x (q0, q1).  Reads of fragments 0..2 were issued at the end of the previous iteration.
        for f in range(16):
            kb, ks = f >> 3, f & 7
            for qb in range(2):
                g = 2 * f + qb
                if qb == 0:
                    if f + 3 < 16: emit(kread(f + 3, ks1))
                    emit(f"s_waitcnt lgkmcnt({min(3, 15 - f)})")
                d = vr(S(nxt, qb, kb), 16)
                c = vr(MINIT(qb), 16) if ks == 0 else d
                emit(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KW(f), 4)}, {ar(Q(qb, ks), 4)}, {c}")
                if g % 8 == 1:                                 # K piece p of tile t+4: the K half of THIS slot (K(t) was read an iteration ago)
                    p = g // 8
                    emit(f"s_add_u32 m0, %[ldsw], {slot * TILE + p * 1024}")
                if g < 16: exp_half(S(cur, 0, 1), [P(0, 1, 0), P(0, 1, 1)], g)
                else: exp_half(S(cur, 1, 0), [P(1, 0, 0), P(1, 0, 1)], g - 16)
                if g % 8 == 1: emit(f"buffer_load_dwordx4 v{232 + g // 8}, %[rk], %[koff] offen lds")
                if 24 <= g < 28: rowsum(0, g - 24)             # q0: (k0, s) final since the last iteration, (k1, s) since gap 16
        for n in range(3):                                     # V^T(t) window: operands 0..2
            for ins in vreads(n, slot): emit(ins)
        emit("s_waitcnt vmcnt(12)")                            # in flight: K(t+4), V(t+2), K(t+3); landed: K(t+2), V(t+1) and older
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
        lb = label[0]; label[0] += 1
        emit("s_cmp_eq_u32 %[cnt], 2")
        emit(f"s_cbranch_scc0 {lb}f")
        emit("s_nop 15")
        mask_tail(nxt)
        emit(f"{lb}:")
        # ---- PV phase: 16 operands (key group j major) x (q0, q1)
        for n in range(16):
            j, d = n >> 2, n & 3
            for qb in range(2):
                g = 2 * n + qb
                if qb == 0:
                    if n + 3 < 16:
                        for ins in vreads(n + 3, slot): emit(ins)
                    emit(f"s_waitcnt lgkmcnt({2 * min(3, 15 - n)})")
                o = ar(O(qb, d), 16)
                emit(f"v_mfma_f32_32x32x16_bf16 {o}, {ar(VW(n), 4)}, {vr(P(qb, j >> 1, j & 1), 4)}, {o}")
                if g % 8 == 1:                                 # V piece p of tile t+3 (V of tile t-1 was last read an iteration ago)
                    p = g // 8
                    emit(f"s_add_u32 m0, %[ldsw], {ds * TILE + VBASE + p * 1024}")
                if g < 16: exp_half(S(cur, 1, 1), [P(1, 1, 0), P(1, 1, 1)], g)
                else: exp_half(S(nxt, 0, 0), [P(0, 0, 0), P(0, 0, 1)], g - 16)
                if g % 8 == 1: emit(f"buffer_load_dwordx4 v{236 + g // 8}, %[rv], %[voff] offen lds")
                if 4 <= g < 6: rowsum(1, g - 4)                # q1: (k0, s) final since PV gap 0
                if 20 <= g < 22: rowsum(1, 2 + g - 20)         # q1: (k1, s) final since PV gap 16 (P(q0, k0) is being rewritten for tile t+1 by now)
        if "dmaoob" not in abl:
            emit("s_add_u32 %[koff], %[koff], %[kstep]")
            emit("s_add_u32 %[voff], %[voff], %[vstep]")
        if pending[0]:
            emit("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % pending[0]); pending[0] = None
        for f in range(3): emit(kread(f, ks2))                # K(t+2) window for the next iteration's QK phase

    def prologue():
        for qb in range(2):
            for ks in range(8):
                off = f" offset:{32 * ks}" if ks else ""
                emit(f"buffer_load_dwordx4 {ar(Q(qb, ks), 4)}, v{240 + qb}, %[rq], 0 offen{off}")
        for tl in range(4):                                   # tiles 0, 1, 2 and the K half of tile 3
            for p in range(8 if tl < 3 else 4):
                imm = tl * TILE + (VBASE if p >= 4 else 0) + (p & 3) * 1024
                emit(f"s_add_u32 m0, %[ldsw], {imm}")
                emit("s_nop 0")
                if p < 4: emit(f"buffer_load_dwordx4 v{232 + (p & 3)}, %[rk], %[koff] offen lds")
                else: emit(f"buffer_load_dwordx4 v{236 + (p & 3)}, %[rv], %[voff] offen lds")
            emit("s_add_u32 %[koff], %[koff], %[kstep]")
            if tl < 3: emit("s_add_u32 %[voff], %[voff], %[vstep]")
        emit(f"v_mov_b32_e32 v{NEGINF}, 0xff800000")
        emit("s_waitcnt vmcnt(12)")                          # Q^T and tiles 0, 1 landed; tile 2 and K(3) stay in flight
        emit("s_barrier")
        for f in range(3): emit(kread(f, 0))
        for f in range(16):                                   # S^T(0) = K(0) . Q^T
            kb, ks = f >> 3, f & 7
            if f + 3 < 16: emit(kread(f + 3, 0))
            emit(f"s_waitcnt lgkmcnt({min(3, 15 - f)})")
            for qb in range(2):
                d = vr(S(0, qb, kb), 16)
                emit(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KW(f), 4)}, {ar(Q(qb, ks), 4)}, {'0' if ks == 0 else d}")
        for i in range(128): emit(f"v_accvgpr_write_b32 a{i}, 0")
        for i in range(8): emit(f"v_mov_b32_e32 v{192 + i}, 0")
        emit("s_nop 15")
        for qb in range(2):                                   # row maxima over tile 0
            s0 = S(0, qb, 0); m = 200 + qb
            emit(f"v_max3_f32 v{m}, v{s0}, v{s0 + 1}, v{s0 + 2}")
            for i in range(3, 31, 2): emit(f"v_max3_f32 v{m}, v{m}, v{s0 + i}, v{s0 + i + 1}")
            emit(f"v_max_f32_e32 v{m}, v{m}, v{s0 + 31}")
            emit(f"v_mov_b32_e32 v{202 + qb}, v{m}")
        emit("s_nop 1")
        for qb in range(2): emit(f"v_permlane32_swap_b32_e32 v{200 + qb}, v{202 + qb}")
        for qb in range(2): emit(f"v_max_f32_e32 v{200 + qb}, v{200 + qb}, v{202 + qb}")
        for qb in range(2):
            for i in range(16): emit(f"v_xor_b32_e32 v{MINIT(qb) + i}, 0x80000000, v{200 + qb}")
            for i in range(32): emit(f"v_sub_f32_e32 v{S(0, qb, 0) + i}, v{S(0, qb, 0) + i}, v{200 + qb}")
        for i in range(8):                                   # first exp slices: (tile 0, q0, k0)
            t0, t1 = temps()
            emit(f"v_exp_f32_e32 v{t0}, v{S(0, 0, 0) + 2 * i}")
            emit(f"v_exp_f32_e32 v{t1}, v{S(0, 0, 0) + 2 * i + 1}")
            emit("s_nop 0")
            emit(f"v_cvt_pk_bf16_f32 v{P(0, 0, i >> 2) + (i & 3)}, v{t0}, v{t1}")
        tmp[0] = 0
        for f in range(3): emit(kread(f, 1))                  # K(1) window for iteration 0

    def epilogue():
        emit("v_cmp_ne_u32_e32 vcc, 0, v244")
        for qb in range(2): emit(f"v_cndmask_b32_e32 v{204 + qb}, v{LACC(qb)}, v{LACC(qb) + 1}, vcc")
        for qb in range(2): emit(f"v_rcp_f32_e32 v{206 + qb}, v{204 + qb}")
        R = 0
        for qb in range(2):
            for d in range(4):
                for k in range(2):
                    base = R; R = (R + 16) % 96
                    for e in range(8): emit(f"v_accvgpr_read_b32 v{base + e}, a{O(qb, d) + 8 * k + e}")
                    for e in range(8): emit(f"v_mul_f32_e32 v{base + e}, v{206 + qb}, v{base + e}")
                    for w in range(4): emit(f"v_cvt_pk_bf16_f32 v{base + 8 + w}, v{base + 2 * w}, v{base + 2 * w + 1}")
                    emit("s_nop 1")
                    emit(f"v_permlane32_swap_b32_e32 v{base + 8}, v{base + 10}")
                    emit(f"v_permlane32_swap_b32_e32 v{base + 9}, v{base + 11}")
                    emit(f"buffer_store_dwordx4 {vr(base + 8, 4)}, v{242 + qb}, %[ro], 0 offen offset:{d * 64 + 32 * k}")

    emit("s_nop 15")
    prologue()
    if "dmaoob" in abl:
        emit("s_mov_b32 %[koff], 0x80000000"); emit("s_mov_b32 %[voff], 0x80000000")
    for it in range(4):
        if int(o_["align"]): emit(".p2align 6" if it == 0 else ".p2align 3")
        if it == 0: emit("1:")
        body(it, it & 1)
        emit("s_add_i32 %[cnt], %[cnt], -1")
        emit("s_cmp_eq_u32 %[cnt], 0")
        if it < 3: emit("s_cbranch_scc1 2f")
        else: emit("s_cbranch_scc0 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0)")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    epilogue()
    return L


def c_function2(opt=None):
    lines = gen2(opt)
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    used_v = list(range(0, 192)) + list(range(200, 208)) + [246]
    clob = [f'"v{i}"' for i in used_v] + [f'"a{i}"' for i in range(0, 224)] + ['"vcc"', '"scc"', '"memory"']
    return f"""// GENERATED by tools/gen_attn_q128_asm.py - do not edit.  {len(lines)} instructions: head_dim 128, 64 queries per wave (two 32-query blocks), prologue + loop + epilogue.
__device__ __forceinline__ void q128_full_qb2(f32x8& lacc, const u32x4& ones, const u32x8& kbase, const u32x8& kbase_hi, const u32x4& trbase, const u32x4& trbase_hi,
        const u32x8& dmaoff, const u32x2& qoff, const u32x2& ooff, uint32_t sel, uint32_t key0, const u32x4& rk, const u32x4& rv, const u32x4& rq, const u32x4& ro,
        int cnt, int rem, uint32_t kstep, uint32_t vstep, uint32_t ldsw) {{
    uint32_t koff = 0, voff = 0;
    int stmp;
    asm volatile(
{text}        : "={{v[192:199]}}"(lacc), [cnt] "+s"(cnt), [koff] "+s"(koff), [voff] "+s"(voff), [stmp] "=&s"(stmp)
        : "{{a[224:227]}}"(ones), "{{v[208:215]}}"(kbase), "{{v[216:223]}}"(kbase_hi), "{{v[224:227]}}"(trbase), "{{v[228:231]}}"(trbase_hi),
          "{{v[232:239]}}"(dmaoff), "{{v[240:241]}}"(qoff), "{{v[242:243]}}"(ooff), "{{v244}}"(sel), "{{v245}}"(key0),
          [rk] "s"(rk), [rv] "s"(rv), [rq] "s"(rq), [ro] "s"(ro), [rem] "s"(rem), [kstep] "s"(kstep), [vstep] "s"(vstep), [ldsw] "s"(ldsw)
        : {", ".join(clob)});
}}
"""


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "candle-video_amd", "csrc", "attn_q128_loop.inc")
    opt = {}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--out": out = args.pop(0)
        else:
            k, v = a.split("="); opt[k] = v
    with open(out, "w") as f:
        f.write(c_function(opt))
        f.write("\n")
        f.write(c_function2(opt))
    print("wrote", out, opt)


if __name__ == "__main__":
    main()


