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
    print("wrote", out, opt)


if __name__ == "__main__":
    main()
