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


DEFAULTS = dict(align=1, phase=0, abl="", cvt_lag=1, v_early=1, stamp=0, split_wait=0, v_gaps="", k_gaps="", dma_gaps="", row_gaps="", v_gaps1="", k_gaps1="", dma_gaps1="", row_gaps1="")


def gen(QB, opt=None, full=False, part=False, stream=False):
    o_ = dict(DEFAULTS); o_.update(opt or {})
    abl = set(x for x in str(o_["abl"]).split("+") if x)
    lag = int(o_["cvt_lag"])
    L = []
    # stream flavour (round 6): every scalar lives in a NAMED SGPR - s[64:79] the item record (host-built, attn_q64.hip Q64StreamItem),
    # s[80:87] scratch - so that the statement stays under the 30-operand limit of inline asm
    KOFF, VOFF, CNT = ("s80", "s81", "s82") if stream else ("%[koff]", "%[voff]", "%[cnt]")
    def emit(ins):
        op = ins.split()[0]
        if "nodma" in abl and (op.startswith("buffer_load") or ins.startswith("s_add_u32 m0")): return
        if "noexp" in abl and op.startswith("v_exp"): ins = ins.replace("v_exp_f32_e32", "v_mov_b32_e32")
        if "dropexp" in abl and (op.startswith("v_exp") or op.startswith("v_cvt_pk")): return
        if "nobar" in abl and op == "s_barrier": return
        if "norow" in abl and op.startswith("v_mfma_f32_16x16x32"): return
        if "nolds" in abl and op.startswith("ds_read"): return
        if "nokread" in abl and op.startswith("ds_read_b128"): return
        if "novread" in abl and op.startswith("ds_read_b64_tr"): return
        if "nosadd" in abl and ins.startswith("s_add_u32 m0"): return
        if "nomfma" in abl and op.startswith("v_mfma_f32_32x32"): return
        if "halfexp" in abl and op.startswith("v_exp") and (int(ins.split()[1].strip("v,")) & 1): return
        if "nocvt" in abl and op.startswith("v_cvt_pk"): return
        L.append(ins)
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

    pending = [None]                  # cvt_lag = 1: (dst, t0, t1) of the previous gap's pair
    def exp_slice(sbase, pbase2, i, filler):
        """registers 2i, 2i+1 of the S^T block at sbase -> word (i & 3) of P operand s = i >> 2"""
        t0, t1 = temps()
        dst = pbase2[i >> 2] + (i & 3)
        if lag:
            if pending[0]: emit("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % pending[0])
            emit(f"v_exp_f32_e32 v{t0}, v{sbase + 2 * i}")
            emit(f"v_exp_f32_e32 v{t1}, v{sbase + 2 * i + 1}")
            for ins in filler: emit(ins)
            pending[0] = (dst, t0, t1)
            return
        emit(f"v_exp_f32_e32 v{t0}, v{sbase + 2 * i}")
        emit(f"v_exp_f32_e32 v{t1}, v{sbase + 2 * i + 1}")
        for ins in filler: emit(ins)
        if not filler: emit("s_nop 0")                       # trans result -> VALU read needs one state
        emit(f"v_cvt_pk_bf16_f32 v{dst}, v{t0}, v{t1}")

    NG = 16 * QB                      # MFMA gaps per iteration: QK groups then PV groups, 8 gaps each

    def parse(key, default):
        v = o_.get(key + ("1" if QB == 1 else "")) or default
        return [int(x) for x in str(v).split(",")] if isinstance(v, str) else list(v)
    # ---- placement tables (gap indices inside one iteration; QK gaps 0 .. 8*QB-1, PV gaps 8*QB .. 16*QB-1) ----
    PV0 = 8 * QB
    if QB == 2:
        v_def = [i // 2 for i in range(16)] if int(o_["v_early"]) else list(range(16))
        k_def = [PV0 + i for i in range(8)]
        d_def = [PV0 + 8 + 2 * p for p in range(4)]
        r_def = [PV0 + 4 + i for i in range(4)] + [PV0 + 12 + i for i in range(4)]
    else:
        v_def = [i // 2 for i in range(16)]
        k_def = [PV0 + i for i in range(8)]
        d_def = [PV0 + 2 * p for p in range(4)]
        r_def = [PV0 + 4 + i for i in range(4)]
    v_gaps, k_gaps = parse("v_gaps", v_def), parse("k_gaps", k_def)
    d_gaps, r_gaps = parse("dma_gaps", d_def), parse("row_gaps", r_def)
    assert len(v_gaps) == 16 and len(k_gaps) == 8 and len(d_gaps) == 4 and len(r_gaps) == 4 * QB
    assert all(0 <= x < PV0 for x in v_gaps) and all(PV0 <= x < NG for x in k_gaps + d_gaps)

    def body(slot, cur):
        nxt = cur ^ 1
        pc, pn = (cur, nxt) if QB == 1 else (0, 0)
        kslot = (slot + 2) & 3
        for gp in range(NG):
            grp, i = gp // 8, gp % 8
            if gp == PV0:                                     # tile t+2 landed for every wave, V^T(t) fragments in
                emit("s_waitcnt vmcnt(4)")
                emit("s_waitcnt lgkmcnt(0)")
                emit("s_barrier")
            pre = []
            fill = []
            if gp < PV0:                                      # ---- QK gap
                qb, kb, ks = grp, i >> 2, i & 3
                d = vr(S(nxt, qb, kb), 16)
                c = vr(MINIT(qb), 16) if ks == 0 else d
                main = f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(i), 4)}, {ar(Q(qb, ks), 4)}, {c}"
                if QB == 2 and qb == 1: sb, pb = S(cur, 1, 0), [P(pc, 1, 0, 0), P(pc, 1, 0, 1)]
                else: sb, pb = S(cur, 0, 1), [P(pc, 0, 1, 0), P(pc, 0, 1, 1)]
            else:                                             # ---- PV gap
                qb, dd, j = grp - QB, i >> 2, i & 3
                o = ar(O(qb, dd), 16)
                main = f"v_mfma_f32_32x32x16_bf16 {o}, {ar(VF(i), 4)}, {vr(P(pc, qb, j >> 1, j & 1), 4)}, {o}"
                if QB == 2 and qb == 0: sb, pb = S(cur, 1, 1), [P(pc, 1, 1, 0), P(pc, 1, 1, 1)]
                else: sb, pb = S(nxt, 0, 0), [P(pn, 0, 0, 0), P(pn, 0, 0, 1)]
            for hidx, vg in enumerate(v_gaps):                # V^T(t) transpose reads
                if vg != gp: continue
                n, hf = hidx >> 1, hidx & 1
                d2, j2 = n >> 2, n & 3
                rowc = (j2 >> 1) * 32 + (j2 & 1) * 16 + 8 * hf
                fill.append(f"ds_read_b64_tr_b16 {ar(VF(n) + 2 * hf, 2)}, v{212 + d2} offset:{slot * TILE + VBASE + rowc * VROW}")
            for ki, kg in enumerate(k_gaps):                  # K(t+2) fragments
                if kg != gp: continue
                fill.append(f"ds_read_b128 {ar(KF(ki), 4)}, v{208 + (ki & 3)} offset:{kslot * TILE + (ki >> 2) * 32 * KROW}")
            for p, dg in enumerate(d_gaps):                   # DMA piece p of tile t+4 into the slot of tile t
                if dg != gp: continue
                imm = slot * TILE + (VBASE if p >= 2 else 0) + (p & 1) * 1024
                pre.append(f"s_add_u32 m0, %[ldsw], {imm}")   # M0 written >= 2 instructions ahead of the load
                if p < 2: fill.append(f"buffer_load_dwordx4 v{214 + (p & 1)}, %[rk], {KOFF} offen lds")
                else: fill.append(f"buffer_load_dwordx4 v{216 + (p & 1)}, %[rv], {VOFF} offen lds")
            for ri, rg in enumerate(r_gaps):                  # row-sum MFMA ri: query block ri / 4, operand (kb, s) = ri % 4
                if rg != gp: continue
                rq, ro = ri // 4, ri % 4
                l = vr(LACC(rq), 4)
                fill.append(f"v_mfma_f32_16x16x32_bf16 {l}, {ar(ONES, 4)}, {vr(P(pc, rq, ro >> 1, ro & 1), 4)}, {l}")
            emit(main)
            for ins in pre: emit(ins)
            exp_slice(sb, pb, i, fill)
        emit(f"s_add_u32 {KOFF}, {KOFF}, %[kstep]")
        emit(f"s_add_u32 {VOFF}, {VOFF}, %[vstep]")
        if pending[0]:                                        # cvt_lag: the body's last pair (two SALU since its exps)
            emit("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % pending[0]); pending[0] = None
        emit("s_waitcnt lgkmcnt(0)")

    def stamp(a, b_):
        if int(o_["stamp"]):
            emit(f"s_memtime %[{a}]"); emit(f"s_memrealtime %[{b_}]"); emit("s_waitcnt lgkmcnt(0)")

    def prologue():
        """full flavour: Q^T fragments, first four tiles in flight, S^T(0), row maxima over tile 0, -m tuples, K(1)
        fragments, first exp slice, O^T / l zeroed.  Tiles past the last one are out of the buffers' range (zeros)."""
        for qb in range(QB):
            for ks in range(4):
                off = f" offset:{32 * ks}" if ks else ""
                emit(f"buffer_load_dwordx4 {ar(Q(qb, ks), 4)}, v{218 + qb}, %[rq], 0 offen{off}")
        for tl in range(4):
            for p in range(4):
                imm = tl * TILE + (VBASE if p >= 2 else 0) + (p & 1) * 1024
                emit(f"s_add_u32 m0, %[ldsw], {imm}")
                emit("s_nop 0")
                if p < 2: emit(f"buffer_load_dwordx4 v{214 + (p & 1)}, %[rk], %[koff] offen lds")
                else: emit(f"buffer_load_dwordx4 v{216 + (p & 1)}, %[rv], %[voff] offen lds")
            emit("s_add_u32 %[koff], %[koff], %[kstep]")
            emit("s_add_u32 %[voff], %[voff], %[vstep]")
        if int(o_["stamp"]): emit("s_memtime %[sa1]")
        # O^T and the row sums are zeroed HERE, under the flight of the loads (round 4; they used to sit behind the S^T(0) MFMAs,
        # on the exposed path of every block): 72 independent writes, no result changes
        for i in range(32 * QB): emit(f"v_accvgpr_write_b32 a{i}, 0")
        for i in range(4 * QB): emit(f"v_mov_b32_e32 v{192 + i}, 0")
        # round 5 experiment (split_wait=1, not the default): S^T(0) needs Q^T and K(0) only - 40 of the block's first 96 KB - so the
        # first wait would cover those (the loads return in order: Q, K(0), V(0), K(1), ...) and V(0), K(1), V(1) land under the first
        # sixteen MFMAs.  Measured on MI355X, same box, alternating fresh processes: 183.9 / 184.1 us per C2 launch against 183.4 / 183.4,
        # per-block constant 5.72 / 5.61 against 5.51 / 5.59 us (profiles/r5_attn_split_wait_ab.json): the prologue is bound by the
        # burst's bandwidth, not by what the first wait covers
        split = int(o_["split_wait"])
        emit("s_waitcnt vmcnt(14)" if split else "s_waitcnt vmcnt(8)")   # Q^T and K(0) [split] / Q^T and tiles 0, 1 landed; the rest stays in flight
        if int(o_["stamp"]): emit("s_memtime %[sa2]")
        emit("s_barrier")
        if int(o_["stamp"]): emit("s_memtime %[sa3]")
        for i in range(8):
            emit(f"ds_read_b128 {ar(KF(i), 4)}, v{208 + (i & 3)} offset:{(i >> 2) * 32 * KROW}")
        emit("s_waitcnt lgkmcnt(0)")
        for qb in range(QB):
            for i in range(8):
                kb, ks = i >> 2, i & 3
                d = vr(S(0, qb, kb), 16)
                emit(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(i), 4)}, {ar(Q(qb, ks), 4)}, {'0' if ks == 0 else d}")
        if split:
            emit("s_waitcnt vmcnt(8)")                       # V(0), K(1), V(1) of this wave landed; the barrier makes that true of every wave's pieces
            emit("s_barrier")
        for i in range(8):                                   # K(1) fragments (the MFMAs above have read theirs long before these land)
            emit(f"ds_read_b128 {ar(KF(i), 4)}, v{208 + (i & 3)} offset:{TILE + (i >> 2) * 32 * KROW}")
        for qb in range(QB):                                 # row maxima: 32 scores in the lane, then the other lane half
            s0 = S(0, qb, 0); m = 200 + qb
            emit(f"v_max3_f32 v{m}, v{s0}, v{s0 + 1}, v{s0 + 2}")
            for i in range(3, 31, 2): emit(f"v_max3_f32 v{m}, v{m}, v{s0 + i}, v{s0 + i + 1}")
            emit(f"v_max_f32_e32 v{m}, v{m}, v{s0 + 31}")
            emit(f"v_mov_b32_e32 v{202 + qb}, v{m}")
        emit("s_nop 1")
        for qb in range(QB): emit(f"v_permlane32_swap_b32_e32 v{200 + qb}, v{202 + qb}")
        for qb in range(QB): emit(f"v_max_f32_e32 v{200 + qb}, v{200 + qb}, v{202 + qb}")
        for qb in range(QB):
            for i in range(16): emit(f"v_xor_b32_e32 v{MINIT(qb) + i}, 0x80000000, v{200 + qb}")
            for i in range(32): emit(f"v_sub_f32_e32 v{S(0, qb, 0) + i}, v{S(0, qb, 0) + i}, v{200 + qb}")
        emit("s_waitcnt lgkmcnt(0)")
        if int(o_["stamp"]): emit("s_memtime %[sa4]")
        for i in range(8):                                   # first exp slice: (tile 0, q0, k0)
            t0, t1 = temps()
            emit(f"v_exp_f32_e32 v{t0}, v{S(0, 0, 0) + 2 * i}")
            emit(f"v_exp_f32_e32 v{t1}, v{S(0, 0, 0) + 2 * i + 1}")
            emit("s_nop 0")
            emit(f"v_cvt_pk_bf16_f32 v{P(0, 0, 0, i >> 2) + (i & 3)}, v{t0}, v{t1}")
        tmp[0] = 0

    def prologue_stream():
        """stream flavour (QB = 2): one item of a persistent workgroup's list.  Item record s[64:79]:
          s64 K offset of the item's first key tile   s65 V offset   s66 Q offset of the item's first query row   s67 O offset (or slab: unused)
          s68 next item's K offset   s69 its V offset   s70 its Q offset   (0x80000000 = none: out of range, zeros, no traffic)
          s71 key tiles - 4   s72 ring slot of tile 0 (0 | 2)   s73 ring slot of the tile after the last (0 | 2)   s74 s72 * 16384
          s75 cold (1: nothing of this item was requested yet)   s76 vector-memory stores the previous item's epilogue issued (8 | 17)
          s[76:79] part flavour: words 1 .. 3 are unused here (the slab descriptor is an operand)
        A WARM item finds Q^T in the wave's 8-KiB staging area (K-tile image: the fragment reads are K reads at another base) and its
        first four tiles requested - by the previous item's last four loop iterations, into the ring slots those freed - so the block
        start costs the S^T(0) chain, not a cold 96-KiB burst (5.5 us per block)."""
        assert QB == 2
        emit("s_mov_b32 s80, s64"); emit("s_mov_b32 s81, s65")
        emit("s_cmp_eq_u32 s75, 0")
        emit("s_cbranch_scc1 5f")
        # ---- cold: Q^T pieces (8 per wave: 8 rows x 128 B each) and the first four tiles, as the full flavour issues them
        emit("s_mov_b32 s83, s66")
        for j in range(8):
            emit(f"s_add_u32 m0, %[ldsq], {j * 1024}")
            emit("s_nop 0")
            emit(f"buffer_load_dwordx4 v{218 + (j & 1)}, %[rq], s83 offen lds")
            emit("s_add_u32 s83, s83, %[qstep8]")
        for tl in range(4):
            emit(f"s_add_u32 s84, s72, {tl}"); emit("s_and_b32 s84, s84, 3"); emit("s_lshl_b32 s84, s84, 14"); emit("s_add_u32 s84, s84, %[ldsw]")
            for p in range(4):
                imm = (VBASE if p >= 2 else 0) + (p & 1) * 1024
                emit(f"s_add_u32 m0, s84, {imm}")
                emit("s_nop 0")
                if p < 2: emit(f"buffer_load_dwordx4 v{214 + (p & 1)}, %[rk], s80 offen lds")
                else: emit(f"buffer_load_dwordx4 v{216 + (p & 1)}, %[rv], s81 offen lds")
            emit("s_add_u32 s80, s80, %[kstep]"); emit("s_add_u32 s81, s81, %[vstep]")
        for i in range(32 * QB): emit(f"v_accvgpr_write_b32 a{i}, 0")
        for i in range(4 * QB): emit(f"v_mov_b32_e32 v{192 + i}, 0")
        emit("s_waitcnt vmcnt(8)")                          # Q^T and tiles 0, 1 landed; 2, 3 stay in flight
        emit("s_branch 6f")
        # ---- warm: everything was requested by the previous item; its epilogue's stores are the only younger operations
        emit("5:")
        emit("s_lshl_b32 s83, %[kstep], 2"); emit("s_add_u32 s80, s80, s83")
        emit("s_lshl_b32 s83, %[vstep], 2"); emit("s_add_u32 s81, s81, s83")
        for i in range(32 * QB): emit(f"v_accvgpr_write_b32 a{i}, 0")
        for i in range(4 * QB): emit(f"v_mov_b32_e32 v{192 + i}, 0")
        emit("s_cmp_eq_u32 s76, 8")
        emit("s_cbranch_scc1 7f")
        emit("s_waitcnt vmcnt(17)")
        emit("s_branch 6f")
        emit("7:")
        emit("s_waitcnt vmcnt(8)")
        emit("6:")
        emit("s_barrier")
        # Q^T fragments from the staging area, K(0) fragments from ring slot s72
        for qb in range(QB):
            for ks in range(4):
                emit(f"ds_read_b128 {ar(Q(qb, ks), 4)}, v{224 + ks} offset:{qb * 32 * KROW}")
        for ks in range(4): emit(f"v_add_u32_e32 v{200 + ks}, s74, v{208 + ks}")
        for i in range(8):
            emit(f"ds_read_b128 {ar(KF(i), 4)}, v{200 + (i & 3)} offset:{(i >> 2) * 32 * KROW}")
        emit("s_waitcnt lgkmcnt(0)")
        for qb in range(QB):
            for i in range(8):
                kb, ks = i >> 2, i & 3
                d = vr(S(0, qb, kb), 16)
                emit(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(i), 4)}, {ar(Q(qb, ks), 4)}, {'0' if ks == 0 else d}")
        for i in range(8):                                   # K(1) fragments: slot s72 + 1 (s72 is 0 or 2: no wrap)
            emit(f"ds_read_b128 {ar(KF(i), 4)}, v{200 + (i & 3)} offset:{TILE + (i >> 2) * 32 * KROW}")
        for qb in range(QB):
            s0 = S(0, qb, 0); m = 204 + qb                   # (v200 .. v203 hold the read bases until the reads above have issued)
            emit(f"v_max3_f32 v{m}, v{s0}, v{s0 + 1}, v{s0 + 2}")
            for i in range(3, 31, 2): emit(f"v_max3_f32 v{m}, v{m}, v{s0 + i}, v{s0 + i + 1}")
            emit(f"v_max_f32_e32 v{m}, v{m}, v{s0 + 31}")
            emit(f"v_mov_b32_e32 v{206 + qb}, v{m}")
        emit("s_nop 1")
        for qb in range(QB): emit(f"v_permlane32_swap_b32_e32 v{204 + qb}, v{206 + qb}")
        for qb in range(QB): emit(f"v_max_f32_e32 v{204 + qb}, v{204 + qb}, v{206 + qb}")
        for qb in range(QB):
            for i in range(16): emit(f"v_xor_b32_e32 v{MINIT(qb) + i}, 0x80000000, v{204 + qb}")
            for i in range(32): emit(f"v_sub_f32_e32 v{S(0, qb, 0) + i}, v{S(0, qb, 0) + i}, v{204 + qb}")
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(8):                                   # first exp slice: (tile 0, q0, k0)
            t0, t1 = temps()
            emit(f"v_exp_f32_e32 v{t0}, v{S(0, 0, 0) + 2 * i}")
            emit(f"v_exp_f32_e32 v{t1}, v{S(0, 0, 0) + 2 * i + 1}")
            emit("s_nop 0")
            emit(f"v_cvt_pk_bf16_f32 v{P(0, 0, 0, i >> 2) + (i & 3)}, v{t0}, v{t1}")
        tmp[0] = 0
        # loop control: phase A = the item's tiles but the last four (the loads issued are its own), phase B = the last four
        # (the loads issued are the NEXT item's first four tiles, into the slots these iterations free)
        emit("s_mov_b32 s82, s71")
        emit("s_mov_b32 s85, 0")                             # phase
        emit("s_mov_b32 s86, s72")                           # entry body
        emit("8:")
        emit("s_cmp_eq_u32 s86, 0")
        emit("s_cbranch_scc1 1f")
        emit("s_branch 3f")

    def next_q_stream():
        """stream flavour: the next item's Q^T into the staging area (this item's fragments left it in the prologue), issued where the
        loop ends: the epilogue and the next prologue's zeroing run under its flight"""
        emit("s_mov_b32 s83, s70")
        for j in range(8):
            emit(f"s_add_u32 m0, %[ldsq], {j * 1024}")
            emit("s_nop 0")
            emit(f"buffer_load_dwordx4 v{218 + (j & 1)}, %[rq], s83 offen lds")
            emit("s_add_u32 s83, s83, %[qstep8]")

    def epilogue():
        """full flavour: O^T / l -> bf16 rows, 16-byte stores (lanes l, l^32 exchange column groups, attention.hip store_o_wide)"""
        emit("v_cmp_ne_u32_e32 vcc, 0, v222")
        for qb in range(QB):
            emit(f"v_cndmask_b32_e32 v{204 + qb}, v{LACC(qb)}, v{LACC(qb) + 1}, vcc")
        for qb in range(QB): emit(f"v_rcp_f32_e32 v{206 + qb}, v{204 + qb}")
        R = 0                                                # S^T registers are free now
        for qb in range(QB):
            for d in range(2):
                for k in range(2):
                    base = R; R = (R + 16) % 96
                    for e in range(8): emit(f"v_accvgpr_read_b32 v{base + e}, a{O(qb, d) + 8 * k + e}")
                    for e in range(8): emit(f"v_mul_f32_e32 v{base + e}, v{206 + qb}, v{base + e}")
                    for w in range(4): emit(f"v_cvt_pk_bf16_f32 v{base + 8 + w}, v{base + 2 * w}, v{base + 2 * w + 1}")
                    emit("s_nop 1")
                    emit(f"v_permlane32_swap_b32_e32 v{base + 8}, v{base + 10}")
                    emit(f"v_permlane32_swap_b32_e32 v{base + 9}, v{base + 11}")
                    emit(f"buffer_store_dwordx4 {vr(base + 8, 4)}, v{220 + qb}, %[ro], {'s67' if stream else '0'} offen offset:{d * 64 + 32 * k}")

    def epilogue_part():
        """part flavour (a key range of a split block, attn_q64.hip): the un-normalised O^T, the row sums and -m leave
        as write-through (sc1) 16-byte stores, fragment-major (1 KiB per wave instruction), for the in-launch merge:
        groups g = 0 .. 16*QB-1 hold a[4g:4g+3], the last group (own l of q0, of q1, -m of q0, of q1)"""
        emit("v_cmp_ne_u32_e32 vcc, 0, v222")
        for qb in range(QB): emit(f"v_cndmask_b32_e32 v{200 + qb}, v{LACC(qb)}, v{LACC(qb) + 1}, vcc")
        if QB == 1: emit("v_mov_b32_e32 v201, 0")
        emit(f"v_mov_b32_e32 v202, v{MINIT(0)}")
        emit(f"v_mov_b32_e32 v203, v{MINIT(QB - 1)}")
        ng = 8 * QB
        for g in range(ng + 1):
            if g % 4 == 0:
                emit(f"s_mov_b32 {KOFF}, {(g // 4) * 4096}")
                emit("s_nop 0")
            src = ar(4 * g, 4) if g < ng else vr(200, 4)
            off = f" offset:{(g % 4) * 1024}" if g % 4 else ""
            emit(f"buffer_store_dwordx4 {src}, v220, %[ro], {KOFF} offen{off} sc1")

    emit("s_nop 15")                                        # operands set up by compiler-scheduled VALU / accvgpr writes
    if stream:
        prologue_stream()
    elif full:
        stamp("sp0", "sq0")
        prologue()
    stamp("st0", "sr0")
    # Code placement (MI355X guide, two waves per SIMD item 8): a hand-scheduled stream shifted to 4 mod 8 bytes runs up to 13 %
    # slower, and where hipcc puts this statement inside the kernel changes with every edit of the C++ around it.  Every
    # unrolled body therefore starts on an 8-byte boundary (the loop head on a 64-byte one); phase=1 shifts them all by one
    # s_nop (A/B aid).
    for it in range(4):
        if int(o_["align"]):
            emit(".p2align 6" if it == 0 else ".p2align 3")
            if int(o_["phase"]): emit("s_nop 0")
        if it == 0: emit("1:")
        if it == 2 and stream: emit("3:")
        body(it, it & 1)
        emit(f"s_add_i32 {CNT}, {CNT}, -1")
        emit(f"s_cmp_eq_u32 {CNT}, 0")
        if it < 3: emit("s_cbranch_scc1 2f")
        else: emit("s_cbranch_scc0 1b")
    emit("2:")
    stamp("st1", "sr1")
    if stream:
        emit("s_cmp_eq_u32 s85, 1")
        emit("s_cbranch_scc1 9f")
        emit("s_mov_b32 s85, 1")
        emit("s_mov_b32 s80, s68"); emit("s_mov_b32 s81, s69")
        emit("s_mov_b32 s82, 4"); emit("s_mov_b32 s86, s73")
        emit("s_branch 8b")
        emit("9:")
        next_q_stream()                                     # (no drain: the next item's tiles and Q^T stay in flight across the epilogue)
    else:
        emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    if full or stream:
        if part: epilogue_part()
        else: epilogue()
        stamp("sp1", "sq1")
    return L


def c_function(QB, opt=None):
    lines = gen(QB, opt)
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
    stamp = int((dict(DEFAULTS, **(opt or {})))["stamp"])
    st_sig = ", unsigned long long& st0, unsigned long long& sr0, unsigned long long& st1, unsigned long long& sr1" if stamp else ""
    st_out = ', [st0] "=&s"(st0), [sr0] "=&s"(sr0), [st1] "=&s"(st1), [sr1] "=&s"(sr1)' if stamp else ""
    clob = ['"v200"', '"v201"', '"v202"', '"v203"', '"v204"', '"v205"', '"v206"', '"v207"'] + [f'"a{i}"' for i in range(128, 160)] + ['"scc"', '"memory"']
    return f"""// GENERATED by tools/gen_attn_q64_asm.py - do not edit.  {len(lines)} instructions, QB = {QB}.
__device__ __forceinline__ void q64_loop_qb{QB}({sig}const u32x4& ones, const u32x4& kbase, const u32x2& trbase, const u32x4& dmaoff,
        const u32x4& rk, const u32x4& rv, int& cnt, uint32_t& koff, uint32_t kstep, uint32_t& voff, uint32_t vstep, uint32_t ldsw{st_sig}) {{
    asm volatile(
{text}        : {outs}[cnt] "+s"(cnt), [koff] "+s"(koff), [voff] "+s"(voff){st_out}
        : {ins}"{{a[160:163]}}"(ones), "{{v[208:211]}}"(kbase), "{{v[212:213]}}"(trbase), "{{v[214:217]}}"(dmaoff),
          [rk] "s"(rk), [rv] "s"(rv), [kstep] "s"(kstep), [vstep] "s"(vstep), [ldsw] "s"(ldsw)
        : {", ".join(clob)});
}}
"""


def c_function_full(QB, opt=None, part=False):
    lines = gen(QB, opt, full=True, part=part)
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    stamp = int((dict(DEFAULTS, **(opt or {})))["stamp"])
    names = ["sp0", "sq0", "st0", "sr0", "st1", "sr1", "sp1", "sq1", "sa1", "sa2", "sa3", "sa4"]
    st_sig = "".join(f", unsigned long long& {n}" for n in names) if stamp else ""
    st_out = "".join(f', [{n}] "=&s"({n})' for n in names) if stamp else ""
    lout = '"={v[192:199]}"(lacc)' if QB == 2 else '"={v[192:195]}"(lacc)'
    ltype = "f32x8" if QB == 2 else "f32x4"
    used_v = [i for i in range(0, 192)] + [i for i in range(192 + 4 * QB, 208)]
    clob = [f'"v{i}"' for i in used_v] + [f'"a{i}"' for i in range(0, 160)] + ['"vcc"', '"scc"', '"memory"']
    return f"""// GENERATED by tools/gen_attn_q64_asm.py - do not edit.  {len(lines)} instructions, QB = {QB}: prologue + loop + {"slab-publishing epilogue (key-range part of a split block)" if part else "epilogue"}.
__device__ __forceinline__ void q64_{"part" if part else "full"}_qb{QB}({ltype}& lacc, const u32x4& ones, const u32x4& kbase, const u32x2& trbase, const u32x4& dmaoff,
        const u32x2& qoff, const u32x2& ooff, uint32_t sel, const u32x4& rk, const u32x4& rv, const u32x4& rq, const u32x4& ro,
        int cnt, uint32_t kstep, uint32_t vstep, uint32_t ldsw{st_sig}) {{
    uint32_t koff = 0, voff = 0;
    asm volatile(
{text}        : {lout}, [cnt] "+s"(cnt), [koff] "+s"(koff), [voff] "+s"(voff){st_out}
        : "{{a[160:163]}}"(ones), "{{v[208:211]}}"(kbase), "{{v[212:213]}}"(trbase), "{{v[214:217]}}"(dmaoff), "{{v[218:219]}}"(qoff),
          "{{v[220:221]}}"(ooff), "{{v222}}"(sel), [rk] "s"(rk), [rv] "s"(rv), [rq] "s"(rq), [ro] "s"(ro),
          [kstep] "s"(kstep), [vstep] "s"(vstep), [ldsw] "s"(ldsw)
        : {", ".join(clob)});
}}
"""


def c_function_stream(opt=None, part=False):
    """one item of a persistent workgroup (QB = 2): item record in s[64:79], lane vectors in v[208:227] (see prologue_stream)"""
    lines = gen(2, opt, full=False, part=part, stream=True)
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    used_v = [i for i in range(0, 192)] + [i for i in range(200, 208)]
    clob = [f'"v{i}"' for i in used_v] + [f'"a{i}"' for i in range(0, 160)] + [f'"s{i}"' for i in range(80, 88)] + ['"vcc"', '"scc"', '"memory"']
    name = "q64_stream_part_qb2" if part else "q64_stream_qb2"
    return f"""// GENERATED by tools/gen_attn_q64_asm.py - do not edit.  {len(lines)} instructions, QB = 2, stream flavour: warm / cold prologue + two-phase loop + next item's Q^T request + {"slab-publishing epilogue (key-range part of a split block)" if part else "epilogue"}.
__device__ __forceinline__ void {name}(f32x8& lacc, const u32x4& ones, const u32x16& lanes, const u32x4& qbase, const u32x16& item,
        const u32x4& rk, const u32x4& rv, const u32x4& rq, const u32x4& ro, uint32_t kstep, uint32_t vstep, uint32_t ldsw, uint32_t ldsq, uint32_t qstep8) {{
    asm volatile(
{text}        : "={{v[192:199]}}"(lacc)
        : "{{a[160:163]}}"(ones), "{{v[208:223]}}"(lanes), "{{v[224:227]}}"(qbase), "{{s[64:79]}}"(item),
          [rk] "s"(rk), [rv] "s"(rv), [rq] "s"(rq), [ro] "s"(ro), [kstep] "s"(kstep), [vstep] "s"(vstep), [ldsw] "s"(ldsw), [ldsq] "s"(ldsq), [qstep8] "s"(qstep8)
        : {", ".join(clob)});
}}
"""


def main():
    """usage: gen_attn_q64_asm.py [--out FILE] [key=value ...]   (keys: see DEFAULTS; the shipped file uses the defaults)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "candle-video_amd", "csrc", "attn_q64_loop.inc")
    opt = {}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--out": out = args.pop(0)
        else:
            k, v = a.split("="); assert k in DEFAULTS, k; opt[k] = v
    with open(out, "w") as f:
        f.write(c_function(2, opt))
        f.write("\n")
        f.write(c_function(1, opt))
        f.write("\n")
        f.write(c_function_full(2, opt))
        f.write("\n")
        f.write(c_function_full(1, opt))
        f.write("\n")
        f.write(c_function_full(2, opt, part=True))
        f.write("\n")
        f.write(c_function_stream(opt))
        f.write("\n")
        f.write(c_function_stream(opt, part=True))
    print("wrote", out, opt)


if __name__ == "__main__":
    main()
