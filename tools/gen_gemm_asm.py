#!/usr/bin/env python3
"""Generates candle-video_amd/csrc/gemm_asm_loop.inc: the hand-scheduled K loop of the one-wave-per-SIMD bf16 GEMM
(csrc/gemm_asm.hip), one inline-asm statement per tile shape, every register named.

Structure (per workgroup = 4 waves, one per SIMD, tile BM x BN, waves laid out WGM x WGN, wave tile WM x WN):
  * LDS: two stages of (BM + BN) rows x 128 B (64 bf16 of K), the image of gemm_big.hip (16-byte chunk c of row r at chunk
    c ^ ((r >> 1) & 7)); operands arrive by buffer LDS-DMA, 1 KiB per wave-instruction, the swizzle applied to the source chunk;
  * MFMA v_mfma_f32_32x32x16_bf16, D = W_frag x A_frag (a lane owns 4 consecutive output columns), accumulators in AGPRs
    (tiles beyond 256 registers in arch VGPRs), fragments in two register sets: while the MB x NB MFMAs of one 16-deep step
    run, the fragments of the next step are read (ds_read_b128, one per MFMA gap) and the DMA pieces of the NEXT K-step are
    issued (one per gap, all of them within the first three of the four steps so that they have >= one step to land);
  * one vmcnt(0) + s_barrier per K-step, placed in the middle of the last step's MFMAs; the first fragment reads of the
    next stage follow it under the remaining MFMAs.
Why asm: with one wave per SIMD the interleave of the 16 DMA pieces and 32 fragment reads of a K-step with its 64 MFMAs IS
the kernel; hipcc issues the pieces back to back at the top of the K-step (the one-wave-per-SIMD C++ kernel of round 1 lost
8-25 % to exactly that).

Register map: v[0:71] fragment sets (set s at 36 s; W fragments nb, then A fragments mb, 4 registers each);
v[72:79] W read bases [stage][k16 step], v[80:87] A read bases; v[88:105] DMA source offsets (A pieces, then W pieces);
v[112:175] accumulator tiles 16 .. 19 (320-row tiles only); a[0:255] accumulator tiles 0 .. 15 (tile = nb * MB + mb)."""
import os
import sys

TILES = {            # name: (BM, BN, WGM, WGN)
    "256x256": (256, 256, 2, 2),
    "320x256": (320, 256, 2, 2),
    "160x256": (160, 256, 1, 4),
}


def gen(BM, BN, WGM, WGN, opt=None):
    opt = opt or {}
    WM, WN = BM // WGM, BN // WGN
    MB, NB = WM // 32, WN // 32
    NT = MB * NB
    AI, BI = BM // 32, BN // 32            # DMA pieces per wave and K-step (A rows, W rows): 8 rows per piece, 4 waves
    STAGE = (BM + BN) * 128
    assert WM % 32 == 0 and WN % 32 == 0 and NB + MB <= 9 and NT <= 20 and AI + BI <= 18
    L = []
    emit = L.append
    abl = set(x for x in str(opt.get("abl", "")).split("+") if x)

    def acc(nb, mb):
        t = nb * MB + mb
        return f"a[{16 * t}:{16 * t + 15}]" if t < 16 else f"v[{112 + 16 * (t - 16)}:{112 + 16 * (t - 16) + 15}]"
    def wfrag(s, nb): return f"v[{36 * s + 4 * nb}:{36 * s + 4 * nb + 3}]"
    def afrag(s, mb): return f"v[{36 * s + 4 * NB + 4 * mb}:{36 * s + 4 * NB + 4 * mb + 3}]"

    def reads(stage, ks, s):
        out = []
        for nb in range(NB): out.append(f"ds_read_b128 {wfrag(s, nb)}, v{72 + 4 * stage + ks} offset:{nb * 4096}")
        for mb in range(MB): out.append(f"ds_read_b128 {afrag(s, mb)}, v{80 + 4 * stage + ks} offset:{mb * 4096}")
        return [] if "nolds" in abl else out

    def dma(stage, p):
        """piece p of the next K-step into `stage`: (M0 write, load)"""
        if "nodma" in abl: return None
        if p < AI:
            imm = stage * STAGE + p * 4096
            return (f"s_add_u32 m0, %[ldsw], {imm}", f"buffer_load_dwordx4 v{88 + p}, %[ra], %[asoff] offen lds")
        imm = stage * STAGE + BM * 128 + (p - AI) * 4096
        return (f"s_add_u32 m0, %[ldsw], {imm}", f"buffer_load_dwordx4 v{88 + p}, %[rw], %[bsoff] offen lds")

    def kstep(stage):
        nxt_stage = stage ^ 1
        npieces = AI + BI
        # pieces over the first three 16-deep steps, most of them early: the last one gets >= 1.5 steps to land
        p0 = (2 * npieces + 4) // 5
        per = [p0, p0, npieces - 2 * p0]
        if int(opt.get("dma_steps", 3)) == 1: per = [npieces, 0, 0]
        if int(opt.get("dma_steps", 3)) == 2: per = [(npieces + 1) // 2, npieces // 2, 0]
        piece = 0
        # the K position of the next K-step; out of range (zeros, no traffic) when this is the last one
        emit("s_cmp_eq_u32 %[cnt], 1")
        emit("s_cselect_b32 %[asoff], 0x80000000, %[ak]")
        emit("s_cselect_b32 %[bsoff], 0x80000000, %[bk]")
        for s16 in range(4):
            cur, nxt = s16 & 1, (s16 & 1) ^ 1
            emit("s_waitcnt lgkmcnt(0)")
            fill = []                                       # (position hint, [instructions]) in issue order
            if s16 < 3:
                rd = reads(stage, s16 + 1, nxt)
                pcs = []
                for _ in range(per[s16]):
                    d = dma(nxt_stage, piece); piece += 1
                    if d: pcs.append(d)
                # interleave: a read every gap first, DMA pieces every other gap
                slots = [[] for _ in range(NT)]
                for i, r in enumerate(rd): slots[min(i, NT - 1)].append(r)
                for i, (m0, ld) in enumerate(pcs):
                    g = min(1 + (i * (NT - 1)) // max(len(pcs), 1), NT - 1)
                    if len(pcs) >= NT: g = min(i, NT - 1)
                    slots[max(g - 1, 0)].append(m0)         # M0 one gap ahead of its load
                    slots[g].append(ld)
            else:
                rd = reads(nxt_stage, 0, nxt)
                half = NT // 2 if not int(opt.get("wait_late", 0)) else NT - 2
                slots = [[] for _ in range(NT)]
                slots[half - 1] += [f"s_waitcnt vmcnt({npieces * int(opt.get('vm_lag', 0))})"] + ([] if "nobar" in abl else ["s_barrier"])
                for i, r in enumerate(rd): slots[min(half + i // 2, NT - 1)].append(r)     # two per gap: done well before the step ends
            i = 0
            for nb in range(NB):
                for mb in range(MB):
                    if "nomfma" not in abl:
                        emit(f"v_mfma_f32_32x32x16_bf16 {acc(nb, mb)}, {wfrag(cur, nb)}, {afrag(cur, mb)}, {acc(nb, mb)}")
                    for ins in slots[i]: emit(ins)
                    i += 1
        emit("s_add_u32 %[ak], %[ak], 128")
        emit("s_add_u32 %[bk], %[bk], 128")

    REG = str(opt.get("stage", "dma")) == "reg"
    if REG: assert NT <= 16 and AI + BI == 16, "register staging: 128 staging registers = two K-steps of 16 pieces"

    def RS(r, p): return 112 + 64 * r + 4 * p

    def gload(r, p):
        """piece p of a K-step into register set r (coalesced 128-byte rows, no swizzle: that is applied by the LDS write)"""
        if "nodma" in abl: return None
        if p < AI: return f"buffer_load_dwordx4 v[{RS(r, p)}:{RS(r, p) + 3}], v{88 + p}, %[ra], %[asoff] offen"
        return f"buffer_load_dwordx4 v[{RS(r, p)}:{RS(r, p) + 3}], v{88 + p}, %[rw], %[bsoff] offen"

    def lwrite(stage, r, p):
        if "nodma" in abl or "nowrite" in abl: return None
        imm = p * 4096 if p < AI else BM * 128 + (p - AI) * 4096
        return f"ds_write_b128 v{104 + stage}, v[{RS(r, p)}:{RS(r, p) + 3}] offset:{imm}"

    def kstep_reg(stage, wr):
        """K-step reading LDS `stage`; register set `wr` (the NEXT K-step's operands, loaded one K-step ago) goes to LDS stage ^ 1,
        set wr ^ 1 receives the operands of the K-step after that.  Per sixteen-deep step: the next step's 8 fragment reads
        in the first gaps, then global loads and LDS writes; the step boundary waits with a COUNTED lgkmcnt for the reads
        only (LDS operations retire in order: the writes issued after them stay in flight); the K-step ends with
        lgkmcnt(0) + barrier and NO vmcnt wait."""
        ld = wr ^ 1
        emit("s_cmp_le_u32 %[cnt], 2")
        emit("s_cselect_b32 %[asoff], 0x80000000, %[ak]")
        emit("s_cselect_b32 %[bsoff], 0x80000000, %[bk]")
        lper = [int(x) for x in str(opt.get("reg_loads", "4,4,4,4")).split(",")]
        wper = [int(x) for x in str(opt.get("reg_writes", "6,5,5,0")).split(",")]
        assert sum(lper) == 16 and sum(wper) == 16
        counted = int(opt.get("lgkm_counted", 1))
        lp = wp = 0                                    # pieces loaded / written so far in this K-step
        pending_w = 0                                  # LDS writes issued after the last fragment read
        for s16 in range(4):
            cur, nxt = s16 & 1, (s16 & 1) ^ 1
            emit(f"s_waitcnt lgkmcnt({min(pending_w, 15) if counted else 0})")
            pending_w = 0
            slots = [[] for _ in range(NT)]
            if s16 < 3:
                rd = reads(stage, s16 + 1, nxt)
                for i, r in enumerate(rd): slots[min(i, NT - 1)].append(r)
                free = list(range(len(rd), NT)) or [NT - 1]
            else:
                half = NT // 2
                slots[half - 1] += ["s_waitcnt lgkmcnt(0)"] + ([] if "nobar" in abl else ["s_barrier"])
                rd = reads(stage ^ 1, 0, nxt)
                for i, r in enumerate(rd): slots[min(half + i // 2, NT - 1)].append(r)
                free = list(range(0, half - 1)) or [0]
            seq = []                                   # loads and writes of this step, alternating
            nl, nw = lper[s16], wper[s16]
            for i in range(max(nl, nw)):
                if i < nl: seq.append("L")
                if i < nw: seq.append("W")
            for i, kind in enumerate(seq):
                g = free[(i * len(free)) // len(seq)]
                if kind == "L":
                    l = gload(ld, lp); lp += 1
                    if l: slots[g].append(l)
                else:
                    w = lwrite(stage ^ 1, wr, wp)
                    if w:
                        q = lp if "nodma" not in abl else 0
                        slots[g].append(f"s_waitcnt vmcnt({15 - wp + q})")          # load `wp` of the older batch has landed
                        slots[g].append(w); pending_w += 1
                    wp += 1
            i = 0
            for nb in range(NB):
                for mb in range(MB):
                    if "nomfma" not in abl:
                        emit(f"v_mfma_f32_32x32x16_bf16 {acc(nb, mb)}, {wfrag(cur, nb)}, {afrag(cur, mb)}, {acc(nb, mb)}")
                    for ins in slots[i]: emit(ins)
                    i += 1
        assert lp == 16 and wp == 16
        emit("s_add_u32 %[ak], %[ak], 128")
        emit("s_add_u32 %[bk], %[bk], 128")

    if REG:
        emit("s_nop 15")
        emit("s_mov_b32 %[asoff], %[ak]")
        emit("s_mov_b32 %[bsoff], %[bk]")
        for p in range(16):
            l = gload(1, p)
            if l: emit(l)
        emit("s_add_u32 %[asoff], %[ak], 128")
        emit("s_add_u32 %[bsoff], %[bk], 128")
        for p in range(16):
            l = gload(0, p)
            if l: emit(l)
        emit("s_add_u32 %[ak], %[ak], 256")               # the K position of the loads issued inside K-step 0
        emit("s_add_u32 %[bk], %[bk], 256")
        for t in range(NT):
            for r in range(16): emit(f"v_accvgpr_write_b32 a{16 * t + r}, 0")
        emit("s_waitcnt vmcnt(16)")
        for p in range(16):
            w = lwrite(0, 1, p)
            if w: emit(w)
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
        for ins in reads(0, 0, 0): emit(ins)
        emit("1:")
        kstep_reg(0, 0)
        emit("s_add_i32 %[cnt], %[cnt], -1")
        emit("s_cmp_eq_u32 %[cnt], 0")
        emit("s_cbranch_scc1 2f")
        kstep_reg(1, 1)
        emit("s_add_i32 %[cnt], %[cnt], -1")
        emit("s_cmp_eq_u32 %[cnt], 0")
        emit("s_cbranch_scc0 1b")
        emit("2:")
        emit("s_waitcnt vmcnt(0)")
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_nop 15")
        emit("s_nop 15")
        return L, dict(MB=MB, NB=NB, NT=NT, AI=AI, BI=BI, REG=True)

    # ---- prologue: first K-step's operands, zeroed accumulators under their flight, first fragments
    emit("s_nop 15")
    emit("s_mov_b32 %[asoff], %[ak]")
    emit("s_mov_b32 %[bsoff], %[bk]")
    for p in range(AI + BI):
        d = dma(0, p)
        if d:
            emit(d[0]); emit("s_nop 0"); emit(d[1])
    emit("s_add_u32 %[ak], %[ak], 128")
    emit("s_add_u32 %[bk], %[bk], 128")
    for t in range(NT):
        for r in range(16):
            if t < 16: emit(f"v_accvgpr_write_b32 a{16 * t + r}, 0")
            else: emit(f"v_mov_b32_e32 v{112 + 16 * (t - 16) + r}, 0")
    emit("s_waitcnt vmcnt(0)")
    emit("s_barrier")
    for ins in reads(0, 0, 0): emit(ins)
    emit("1:")
    kstep(0)
    emit("s_add_i32 %[cnt], %[cnt], -1")
    emit("s_cmp_eq_u32 %[cnt], 0")
    emit("s_cbranch_scc1 2f")
    kstep(1)
    emit("s_add_i32 %[cnt], %[cnt], -1")
    emit("s_cmp_eq_u32 %[cnt], 0")
    emit("s_cbranch_scc0 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0)")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    return L, dict(MB=MB, NB=NB, NT=NT, AI=AI, BI=BI)


def gen16(BM, BN, WGM, WGN, opt=None):
    """The same K loop on v_mfma_f32_16x16x32_bf16 (round 3).  What the vendor library runs on these shapes (disassembly of its
    MT256x256x64_MI16x16x1 kernel): four waves of 128 x 128, LDS-DMA staging, 16x16x32 MFMAs - the shape that holds a higher
    clock than 32x32x16 at equal cycles per FLOP (MI355X guide, DVFS item 7).  Per 64-deep K-step a wave runs two 32-deep
    halves of MB x NB MFMAs (16 x 16 output blocks, 4 accumulator registers each); fragments in two register sets of
    (NB + MB) x 4 registers: while half h runs, the fragments of the next half are read (one ds_read_b128 per even gap) and,
    in the first half, the DMA pieces of the NEXT K-step are issued (odd gaps; M0 one gap ahead); vmcnt(0) + s_barrier sit
    in the middle of the second half, the first fragment reads of the next stage follow it.
    Register map: v[0 : 8 (NB + MB)) fragment sets; v[RB : RB+8) read bases (W [stage][half], A [stage][half]); v[RB+8 : RB+8+AI+BI)
    DMA source offsets; accumulators a[4 t : 4 t + 3], t = nb * MB + mb."""
    opt = opt or {}
    WM, WN = BM // WGM, BN // WGN
    MB, NB = WM // 16, WN // 16
    NT = MB * NB
    AI, BI = BM // 32, BN // 32
    STAGE = (BM + BN) * 128
    FS = 4 * (NB + MB)                     # registers per fragment set
    RB = 2 * FS                            # first register after the fragment sets
    assert 4 * NT <= 320 and AI + BI <= 18 and RB + 8 + AI + BI <= 192
    L = []
    emit = L.append
    abl = set(x for x in str(opt.get("abl", "")).split("+") if x)
    def acc(nb, mb):                                       # blocks 0..63 in the AGPR half, the rest (320 x 256 tile) in v[192:255]
        t = nb * MB + mb
        return f"a[{4 * t}:{4 * t + 3}]" if t < 64 else f"v[{192 + 4 * (t - 64)}:{192 + 4 * (t - 64) + 3}]"
    def wfrag(st, nb): return f"v[{FS * st + 4 * nb}:{FS * st + 4 * nb + 3}]"
    def afrag(st, mb): return f"v[{FS * st + 4 * NB + 4 * mb}:{FS * st + 4 * NB + 4 * mb + 3}]"
    def reads(stage, kh, st):
        out = [f"ds_read_b128 {wfrag(st, nb)}, v{RB + 2 * stage + kh} offset:{nb * 2048}" for nb in range(NB)]
        out += [f"ds_read_b128 {afrag(st, mb)}, v{RB + 4 + 2 * stage + kh} offset:{mb * 2048}" for mb in range(MB)]
        return [] if "nolds" in abl else out
    def dma(stage, p):
        if "nodma" in abl: return None
        if p < AI:
            return (f"s_add_u32 m0, %[ldsw], {stage * STAGE + p * 4096}", f"buffer_load_dwordx4 v{RB + 8 + p}, %[ra], %[asoff] offen lds")
        return (f"s_add_u32 m0, %[ldsw], {stage * STAGE + BM * 128 + (p - AI) * 4096}", f"buffer_load_dwordx4 v{RB + 8 + p}, %[rw], %[bsoff] offen lds")
    bar_gap = int(opt.get("bar_gap", NT // 2))
    PGR = int(opt.get("pgr", 2))
    TRACE = int(opt.get("trace", 0)) and PGR == 2          # s_memtime stamps summed per segment of the K-step (tools/gemm_asm_tune.py trace)
    npieces = AI + BI
    # ---- conv mode (conv=1; round 5): the A operand of a 3x3x3 conv3d as an implicit GEMM with the activation rows re-staged per tap
    # (gemm_big's conv mode), K-steps in the canonical order of every bf16 conv kernel here - frame tap `it`, 64-channel slice `kc`,
    # in-plane tap (ih, iw).  Per K-step the A pieces' lane offsets are rebuilt: offset = abase[it][piece] + kc * 128 + tap delta
    # (32-bit wrap: the delta may be negative) where the tap's voxel lies inside the plane (6 validity bits per piece), else out of
    # range (zeros: the H / W zero padding); the temporal replicate padding is inside abase (one set per frame tap).
    # Registers: v[RB+24 : RB+32) abase of the CURRENT frame tap, [RB+32 : RB+40) / [RB+40 : RB+48) of the next two; v[RB+48 : RB+56)
    # validity masks; v[RB+56], v[RB+57] temporaries.  The fetch state (iw, ih, kc, running offsets) lives in SGPR operands and is
    # advanced once per K-step without branches, except the copy of the abase sets when `it` changes (twice per tile at most).
    CONV = int(opt.get("conv", 0))
    if CONV: assert PGR == 2 and not TRACE and AI == 8 and RB + 59 <= 256
    CUR, NXT, NX2, VMK, VT0, VT1, VOOB = RB + 24, RB + 32, RB + 40, RB + 48, RB + 56, RB + 57, RB + 58      # VOOB: 0x80000000 (a literal and VCC cannot share an instruction's constant bus)

    def conv_scalars():
        """top of a fetch: the wave-uniform parts of the step being fetched (state = that step)"""
        return ["s_add_u32 %[t0], %[kcoff], %[tapd]",                      # lane-offset addend of the A rows: slice + tap delta
                "s_lshl_b32 %[t1], 1, %[ih]", "s_lshl_b32 %[t2], 8, %[iw]", "s_or_b32 %[vbit], %[t1], %[t2]",
                "s_cmp_le_u32 %[cnt], %[livemin]",
                "s_cselect_b32 %[asoff], 0x80000000, 0",
                "s_cselect_b32 %[bsoff], 0x80000000, %[bso]"]

    def conv_offsets():
        """the AI A-piece offsets of the step being fetched: 4 VALU per piece, returned as AI groups"""
        out = []
        for j in range(AI):
            out.append([f"v_add_u32_e32 v{VT0}, %[t0], v{CUR + j}",
                        f"v_and_b32_e32 v{VT1}, %[vbit], v{VMK + j}",
                        f"v_cmp_eq_u32_e32 vcc, %[vbit], v{VT1}",
                        f"v_cndmask_b32_e32 v{RB + 8 + j}, v{VOOB}, v{VT0}, vcc"])
        return out

    def conv_advance():
        """state of the NEXT step to fetch (three chunks of SALU, then the rare copy of the abase sets)"""
        c1 = ["s_add_u32 %[iw], %[iw], 1", "s_add_u32 %[tapd], %[tapd], %[cinb]", "s_add_u32 %[bso], %[bso], %[nk2]",
              "s_cmp_eq_u32 %[iw], 3", "s_cselect_b32 %[iw], 0, %[iw]", "s_cselect_b32 %[t1], %[fixw], 0", "s_cselect_b32 %[t2], 1, 0",
              "s_add_u32 %[tapd], %[tapd], %[t1]", "s_add_u32 %[ih], %[ih], %[t2]"]
        c2 = ["s_cmp_eq_u32 %[ih], 3", "s_cselect_b32 %[ih], 0, %[ih]", "s_cselect_b32 %[t1], %[fixh], 0", "s_cselect_b32 %[t2], %[fixkc], 0",
              "s_cselect_b32 %[t3], 128, 0", "s_cselect_b32 %[t4], 1, 0",
              "s_add_u32 %[tapd], %[tapd], %[t1]", "s_add_u32 %[bso], %[bso], %[t2]", "s_add_u32 %[kcoff], %[kcoff], %[t3]", "s_add_u32 %[kc], %[kc], %[t4]"]
        c3 = ["s_cmp_eq_u32 %[kc], %[nkc]", "s_cselect_b32 %[kc], 0, %[kc]", "s_cselect_b32 %[kcoff], 0, %[kcoff]", "s_cselect_b32 %[t1], %[fixit], 0",
              "s_cselect_b32 %[t2], 1, 0", "s_add_u32 %[bso], %[bso], %[t1]", "s_cmp_eq_u32 %[t2], 1", "s_cbranch_scc0 7f"]
        c3 += [f"v_mov_b32_e32 v{CUR + j}, v{NXT + j}" for j in range(AI)] + [f"v_mov_b32_e32 v{NXT + j}, v{NX2 + j}" for j in range(AI)] + ["7:"]
        return c1, c2, c3

    def kstep(stage):
        """pgr = 1: the DMA pieces of K-step t+1 go into the other stage during the first half, vmcnt(0) + barrier in the
        middle of the second half (>= half a K-step for a piece to land)."""
        nxt_stage = stage ^ 1
        emit("s_cmp_eq_u32 %[cnt], 1")
        emit("s_cselect_b32 %[asoff], 0x80000000, %[ak]")
        emit("s_cselect_b32 %[bsoff], 0x80000000, %[bk]")
        for kh in range(2):
            cur, nxt = kh, kh ^ 1
            emit("s_waitcnt lgkmcnt(0)")
            slots = [[] for _ in range(NT)]
            if kh == 0:
                rd = reads(stage, 1, nxt)
                for i, r in enumerate(rd): slots[min(2 * i, NT - 1)].append(r)
                pcs = [d for d in (dma(nxt_stage, p) for p in range(npieces)) if d]
                for i, (m0, ld) in enumerate(pcs):
                    g = min(2 * i + 1, NT - 1)
                    slots[g - 1].append(m0)
                    slots[g].append(ld)
            else:
                slots[bar_gap - 1] += ["s_waitcnt vmcnt(0)"] + ([] if "nobar" in abl else ["s_barrier"])
                rd = reads(nxt_stage, 0, nxt)
                span = NT - bar_gap
                for i, r in enumerate(rd): slots[min(bar_gap + (i * span) // max(len(rd), 1), NT - 1)].append(r)
            i = 0
            for nb in range(NB):
                for mb in range(MB):
                    if "nomfma" not in abl:
                        emit(f"v_mfma_f32_16x16x32_bf16 {acc(nb, mb)}, {wfrag(cur, nb)}, {afrag(cur, mb)}, {acc(nb, mb)}")
                    for ins in slots[i]: emit(ins)
                    i += 1
        emit("s_add_u32 %[ak], %[ak], 128")
        emit("s_add_u32 %[bk], %[bk], 128")
        wrap()

    RES = int(opt.get("res_rows", 0))                       # residual tile rows per lane prefetched inside the loop (see c_function16)
    RES_PER = 2                                             # rows per K-step -> 2 * RES_PER loads
    RES_V = RB + 28                                         # v[RES_V : RES_V + 2 RES) first halves, then second halves; lane offsets in v[RB+24 : RB+25]
    if RES: assert AI + BI <= 16 and RES % RES_PER == 0 and RES_V + 4 * RES <= 256 and not TRACE

    def kstep2(stage, res_row=None):
        """pgr = 2 (the vendor kernel's depth): ALL fragments of K-step t are in registers early in iteration t (half 0's were
        read at the end of t-1, half 1's in the first gaps), so after a barrier the stage is free and takes the DMA pieces of
        K-step t+2, spread over the REST of the iteration: a 1 KB piece occupies the CU's address unit for ~16 cycles and the
        four waves run in step, so the 64 pieces of a K-step need ~1024 of its ~2048 cycles - packed into 37 gaps they
        stalled the issue (s_memtime trace: 1056 cycles for a 592-cycle segment).  K-step t+1 (issued during t-1) is waited
        for with a COUNTED vmcnt late in the second half: a piece has more than a K-step to land."""
        g_rd = int(opt.get("rd_gaps", 1))                  # a fragment read every g_rd gaps
        if CONV:
            for ins in conv_scalars(): emit(ins)
        else:
            emit("s_cmp_le_u32 %[cnt], 2")
            emit("s_cselect_b32 %[asoff], 0x80000000, %[ak]")
            emit("s_cselect_b32 %[bsoff], 0x80000000, %[bk]")
        nrd = NB + MB
        G = 2 * NT                                          # gaps of the K-step, half 0 then half 1
        slots = [[] for _ in range(G)]
        rd = reads(stage, 1, 1)
        for i, r in enumerate(rd): slots[min(g_rd * i, NT - 1)].append(r)
        b1 = min(g_rd * nrd + int(opt.get("b1_lag", 6)), NT - 2)      # half-1 fragments returned: the stage is free for every wave after this barrier
        slots[b1] += ["s_waitcnt lgkmcnt(0)"] + ([] if "nobar" in abl else ["s_barrier"])
        rd2_step = int(opt.get("rd2_step", 2))
        gb2 = G - int(opt.get("gb2_back", rd2_step * nrd + 4))  # K-step t+1 landed for every wave after this barrier
        last = int(opt.get("dma_last", G - 3))
        pcs = [d for d in (dma(stage, p) for p in range(npieces)) if d]
        dstep = max(2, (last - b1 - 2) // max(len(pcs), 1))
        before_b2 = 0
        nres = 0
        if res_row is not None:
            # residual rows res_row .. res_row + RES_PER - 1 of this lane: plain buffer loads between the fragment reads and the first
            # barrier, i.e. OLDER than this K-step's pieces and YOUNGER than the pieces the second barrier waits for
            g0 = min(g_rd * nrd + 1, b1 - 1)
            for j in range(RES_PER):
                r = res_row + j
                slots[min(g0 + 2 * j, b1 - 1)] += [f"buffer_load_dwordx2 v[{RES_V + 2 * r}:{RES_V + 2 * r + 1}], v{RB + 24}, %[rres], %[roff] offen",
                                                   f"buffer_load_dwordx2 v[{RES_V + 2 * RES + 2 * r}:{RES_V + 2 * RES + 2 * r + 1}], v{RB + 25}, %[rres], %[roff] offen",
                                                   "s_add_u32 %[roff], %[roff], %[rstride]"]
                nres += 2
        for i, (m0, ld) in enumerate(pcs):
            g = b1 + 2 + dstep * i
            if g == gb2: g += 1
            before_b2 += g < gb2
            slots[g - 1].append(m0)
            slots[g].append(ld)
        slots[gb2] = [f"s_waitcnt vmcnt({before_b2 + nres})"] + ([] if "nobar" in abl else ["s_barrier"]) + slots[gb2]
        if CONV:
            # the rebuilt A offsets in front of the first piece (from gap 2 on: the previous K-step's pieces were all issued inside it),
            # the state advance behind the last one
            flat = [ins for grp in conv_offsets() for ins in grp]           # two VALU per gap: what an MFMA's issue shadow holds
            for i in range(0, len(flat), 2): slots[2 + i // 2] += flat[i:i + 2]
            assert 2 + (len(flat) - 1) // 2 < b1 and b1 + 2 + dstep * (len(pcs) - 1) < G - 8
            c1, c2, c3 = conv_advance()
            slots[G - 7] += c1; slots[G - 6] += c2; slots[G - 5] += c3
        rd = reads(stage ^ 1, 0, 0)
        for i, r in enumerate(rd): slots[min(gb2 + 1 + rd2_step * i, G - 1)].append(r)
        if TRACE:
            slots[b1 - 4].append("s_memtime s[62:63]"); slots[b1].insert(2 - ("nobar" in abl), "s_memtime s[64:65]"); slots[NT - 5].append("s_memtime s[66:67]")
            slots[gb2].insert(2 - ("nobar" in abl), "s_memtime s[70:71]"); slots[G - 5].append("s_memtime s[72:73]")
        for kh in range(2):
            emit("s_waitcnt lgkmcnt(0)")
            if TRACE and kh == 0:
                # segments of the K-step that just ended: A->B reads, B->C first barrier, C->D first half after it, D->E half
                # switch, E->G second half up to the second barrier, G->H reads of the next stage, H->top wait; the stamps sit
                # >= 4 MFMAs before the wait that makes them readable except the "after" ones (A, C, E, G), consumed at the next
                for i, (a, b) in enumerate([(60, 62), (62, 64), (64, 66), (66, 68), (68, 70), (70, 72)]):
                    emit(f"s_sub_u32 s86, s{b}, s{a}"); emit(f"s_add_u32 s{76 + i}, s{76 + i}, s86")
                emit("s_sub_u32 s86, s60, s84"); emit("s_add_u32 s82, s82, s86"); emit("s_mov_b32 s84, s72")
                emit("s_memtime s[60:61]")
            if TRACE and kh == 1: emit("s_memtime s[68:69]")
            i = kh * NT
            for nb in range(NB):
                for mb in range(MB):
                    if "nomfma" not in abl:
                        emit(f"v_mfma_f32_16x16x32_bf16 {acc(nb, mb)}, {wfrag(kh, nb)}, {afrag(kh, mb)}, {acc(nb, mb)}")
                    for ins in slots[i]: emit(ins)
                    i += 1
        if not CONV:
            emit("s_add_u32 %[ak], %[ak], 128")
            emit("s_add_u32 %[bk], %[bk], 128")
        wrap()

    def wrap():
        """experiment (stagger=1): the K loop of a block starts at a block-dependent K position and wraps at the end of K
        (what the vendor kernels call StaggerU); changes the f32 summation order, so it is not a shipped option"""
        if not int(opt.get("stagger", 0)): return
        emit("s_cmp_eq_u32 %[ak], %[kend]")
        emit("s_cselect_b32 %[ak], 0, %[ak]")
        emit("s_cselect_b32 %[bk], 0, %[bk]")

    emit("s_nop 15")
    if CONV:
        emit(f"v_mov_b32_e32 v{VOOB}, 0x80000000")
        # K-steps 0 and 1 of the range into stages 0 and 1 (livemin = 0 / 1: a range of one step fetches nothing for the second)
        for st in range(2):
            emit(f"s_mov_b32 %[livemin], {st}")
            for ins in conv_scalars(): emit(ins)
            for grp in conv_offsets():
                for ins in grp: emit(ins)
            emit("s_nop 1")
            for p in range(npieces):
                d = dma(st, p)
                if d:
                    emit(d[0]); emit("s_nop 0"); emit(d[1])
            for chunk in conv_advance():
                for ins in chunk: emit(ins)
        emit("s_mov_b32 %[livemin], 2")
    else:
        emit("s_mov_b32 %[asoff], %[ak]")
        emit("s_mov_b32 %[bsoff], %[bk]")
        for p in range(npieces):
            d = dma(0, p)
            if d:
                emit(d[0]); emit("s_nop 0"); emit(d[1])
        emit("s_add_u32 %[ak], %[ak], 128")
        emit("s_add_u32 %[bk], %[bk], 128")
        wrap()
    if PGR == 2 and not CONV:                               # K-step 1 into stage 1 (out of range when the problem has one K-step)
        emit("s_cmp_le_u32 %[cnt], 1")
        emit("s_cselect_b32 %[asoff], 0x80000000, %[ak]")
        emit("s_cselect_b32 %[bsoff], 0x80000000, %[bk]")
        for p in range(npieces):
            d = dma(1, p)
            if d:
                emit(d[0]); emit("s_nop 0"); emit(d[1])
        emit("s_add_u32 %[ak], %[ak], 128")
        emit("s_add_u32 %[bk], %[bk], 128")
        wrap()
    for r in range(4 * NT): emit(f"v_accvgpr_write_b32 a{r}, 0" if r < 256 else f"v_mov_b32 v{192 + r - 256}, 0")
    emit(f"s_waitcnt vmcnt({npieces if PGR == 2 else 0})")
    emit("s_barrier")
    if TRACE:
        emit("s_memrealtime s[88:89]")                     # 100 MHz clock at the end of the prologue (first two K-steps landed) -> t7
        emit("s_memtime s[72:73]"); emit("s_waitcnt lgkmcnt(0)")
        for r in (60, 62, 64, 66, 68, 70, 84): emit(f"s_mov_b32 s{r}, s72")
        for r in range(76, 84): emit(f"s_mov_b32 s{r}, 0")
    for ins in reads(0, 0, 0): emit(ins)
    if RES:
        # phase A: the first RES / RES_PER K-steps, unrolled, each requesting RES_PER rows of the residual tile (the caller
        # guarantees at least that many K-steps + 2); an even count, so the main loop starts on stage 0 as always
        assert PGR == 2 and (RES // RES_PER) % 2 == 0
        for t in range(RES // RES_PER):
            kstep2(t & 1, res_row=t * RES_PER)
            emit("s_add_i32 %[cnt], %[cnt], -1")
        emit("s_cmp_eq_u32 %[cnt], 0")
        emit("s_cbranch_scc1 2f")
    emit(".p2align 6")
    emit("1:")
    (kstep2 if PGR == 2 else kstep)(0)
    emit("s_add_i32 %[cnt], %[cnt], -1")
    emit("s_cmp_eq_u32 %[cnt], 0")
    emit("s_cbranch_scc1 2f")
    (kstep2 if PGR == 2 else kstep)(1)
    emit("s_add_i32 %[cnt], %[cnt], -1")
    emit("s_cmp_eq_u32 %[cnt], 0")
    emit("s_cbranch_scc0 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0)")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    if TRACE:
        for i in range(7): emit(f"s_mov_b32 %[t{i}], s{76 + i}")
        emit("s_mov_b32 %[t7], s88")
    return L, dict(MB=MB, NB=NB, NT=NT, AI=AI, BI=BI, FS=FS, RB=RB, TRACE=TRACE, RES=RES, RES_V=RES_V, CONV=CONV)


def c_function16(name, BM, BN, WGM, WGN, opt=None):
    lines, d = gen16(BM, BN, WGM, WGN, opt)
    NT, RB, FS = d["NT"], d["RB"], d["FS"]
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    n32 = (4 * NT + 31) // 32
    sig = ", ".join(f"f32x32& c{i}" for i in range(n32))
    def creg(i): return f"a[{32 * i}:{32 * i + 31}]" if i < 8 else f"v[{192 + 32 * (i - 8)}:{192 + 32 * (i - 8) + 31}]"
    outs = ", ".join(f'"={{{creg(i)}}}"(c{i})' for i in range(n32))
    clob = [f'"v{i}"' for i in range(0, 2 * FS)] + ['"scc"', '"memory"']
    npc = d["AI"] + d["BI"]
    d1sig = ", const u32x2& dma1" if npc > 16 else ""
    d1in = f', "{{v[{RB + 24}:{RB + 25}]}}"(dma1)' if npc > 16 else ""
    ressig = resouts = resins = ""
    fname = name
    if d["RES"]: text = '        "s_mov_b32 %[roff], 0\\n\\t"\n' + text
    if d["RES"]:
        # the residual-prefetching variant: rres = buffer descriptor of the residual tile, rvoff = this lane's byte offsets of its
        # first row's two 4-column groups (or out of range), rstride = bytes from one of the lane's rows to the next (8 tile rows);
        # res0.. = the lane's RES rows of the first column group, then of the second (two dwords per row)
        nres16 = (4 * d["RES"] + 15) // 16
        assert 4 * d["RES"] % 16 == 0
        fname = name + "_res"
        ressig = ", " + ", ".join(f"u32x16& res{i}" for i in range(nres16)) + ", const u32x2& rvoff, const u32x4& rres, uint32_t rstride"
        resouts = ", " + ", ".join(f'"={{v[{d["RES_V"] + 16 * i}:{d["RES_V"] + 16 * i + 15}]}}"(res{i})' for i in range(nres16)) + ', [roff] "=&s"(roff)'
        resins = f', "{{v[{RB + 24}:{RB + 25}]}}"(rvoff), [rres] "s"(rres), [rstride] "s"(rstride)'
    trsig = trouts = ""
    if d["TRACE"]:
        clob += [f'"s{i}"' for i in range(60, 90)]
        trsig = ", uint32_t (&tr)[8]"
        trouts = ", " + ", ".join(f'[t{i}] "=s"(tr[{i}])' for i in range(8))
    return f"""// GENERATED by tools/gen_gemm_asm.py - do not edit.  {len(lines)} instructions: tile {BM} x {BN}, waves {WGM} x {WGN}, v_mfma_f32_16x16x32_bf16.
__device__ __forceinline__ void gemm_asm16_loop_{fname}({sig}, const u32x8& rbase, const u32x16& dma0{d1sig},
        const u32x4& ra, const u32x4& rw, int cnt, uint32_t ak, uint32_t bk, uint32_t ldsw, uint32_t kend{trsig}{ressig}) {{
    uint32_t asoff, bsoff{", roff" if d["RES"] else ""};
    asm volatile(
{text}        : {outs}, [cnt] "+s"(cnt), [ak] "+s"(ak), [bk] "+s"(bk), [asoff] "=&s"(asoff), [bsoff] "=&s"(bsoff){trouts}{resouts}
        : "{{v[{RB}:{RB + 7}]}}"(rbase), "{{v[{RB + 8}:{RB + 23}]}}"(dma0){d1in}{resins}, [ra] "s"(ra), [rw] "s"(rw), [ldsw] "s"(ldsw), [kend] "s"(kend)
        : {", ".join(clob)});
}}
"""


def c_function16_conv(name, BM, BN, WGM, WGN, opt=None):
    """The conv-mode loop (gen16 conv=1) as a C function.  State operands (in / out): the fetch position inside the canonical
    K order - iw, ih (in-plane tap), kc (64-channel slice), tapd = ((ih - 1) * W + (iw - 1)) * Cin * 2, kcoff = kc * 128, bso = byte
    offset of the weight rows of (frame tap, tap, slice) - and cnt = K-steps of this range.  Constants: cinb = Cin * 2, nk2 = N * K * 2,
    fixw = W * Cin * 2 - 3 cinb, fixh = -3 W * Cin * 2, fixkc = 128 - 9 nk2, fixit = 9 nk2 - KC * 128, nkc = KC."""
    lines, d = gen16(BM, BN, WGM, WGN, dict(opt or {}, conv=1))
    NT, RB, FS = d["NT"], d["RB"], d["FS"]
    assert 4 * NT <= 256
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    n32 = (4 * NT + 31) // 32
    sig = ", ".join(f"f32x32& c{i}" for i in range(n32))
    outs = ", ".join(f'"={{a[{32 * i}:{32 * i + 31}]}}"(c{i})' for i in range(n32))
    clob = [f'"v{i}"' for i in range(0, 2 * FS)] + [f'"v{RB + 56}"', f'"v{RB + 57}"', f'"v{RB + 58}"', '"vcc"', '"scc"', '"memory"']
    st = ["iw", "ih", "kc", "tapd", "kcoff", "bso"]
    tmps = ["t0", "t1", "t2", "t3", "t4", "vbit", "livemin", "asoff", "bsoff"]
    consts = ["cinb", "nk2", "fixw", "fixh", "fixkc", "fixit", "nkc"]
    return f"""// GENERATED by tools/gen_gemm_asm.py - do not edit.  {len(lines)} instructions: tile {BM} x {BN}, waves {WGM} x {WGN}, v_mfma_f32_16x16x32_bf16, conv mode (3x3x3 taps, A rows re-staged per tap).
__device__ __forceinline__ void gemm_asm16_conv_loop_{name}({sig}, const u32x8& rbase, u32x16& dma0, u32x8& abase0, u32x8& abase1, u32x8& abase2, const u32x8& vmask,
        const u32x4& ra, const u32x4& rw, int cnt, uint32_t ldsw, {", ".join("uint32_t " + x for x in st)}, {", ".join("uint32_t " + x for x in consts)}) {{
    uint32_t {", ".join(tmps)};
    asm volatile(
{text}        : {outs}, [cnt] "+s"(cnt), {", ".join(f'[{x}] "+s"({x})' for x in st)}, {", ".join(f'[{x}] "=&s"({x})' for x in tmps)},
          "+{{v[{RB + 8}:{RB + 23}]}}"(dma0), "+{{v[{RB + 24}:{RB + 31}]}}"(abase0), "+{{v[{RB + 32}:{RB + 39}]}}"(abase1), "+{{v[{RB + 40}:{RB + 47}]}}"(abase2)
        : "{{v[{RB}:{RB + 7}]}}"(rbase), "{{v[{RB + 48}:{RB + 55}]}}"(vmask), [ra] "s"(ra), [rw] "s"(rw), [ldsw] "s"(ldsw), {", ".join(f'[{x}] "s"({x})' for x in consts)}
        : {", ".join(clob)});
}}
"""


def store_functions16(name, BM, BN, WGM, WGN):
    """After the K loop: f32 accumulators -> LDS, straight from the AGPRs (ds_write_b128 takes an AGPR source), one function
    per group of a wave's blocks that falls into one row pass of the wide epilogue in csrc/gemm_asm.hip (RP tile rows of
    1 KiB each).  The accumulator tuples are tied INPUTS, so the compiler never copies or extracts them (extracting elements
    of 1024-bit tuples under a wave-uniform branch is what it cannot do: 'Illegal instruction detected').
    ad[nb] = LDS byte address of (local row rr, chunk (column block nb, q) ^ rr); block (lm, nb) sits lm * 16 KiB further.
    256 x 256 (waves 2 x 2, RP 128): one group, pass = wm.  160 x 256 (1 x 4, RP 80): groups mb 0-4 / 5-9 = passes 0 / 1.
    320 x 256 (2 x 2 waves of 160 rows, RP 80): the same two groups, pass = 2 wm + group."""
    WM, WN = BM // WGM, BN // WGN
    MB, NB = WM // 16, WN // 16
    NT = MB * NB
    RP = 128 if BM == 256 else 80
    n32 = (4 * NT + 31) // 32
    sig = ", ".join(f"const f32x32& c{i}" for i in range(n32))
    def creg(i): return f"a[{32 * i}:{32 * i + 31}]" if i < 8 else f"v[{192 + 32 * (i - 8)}:{192 + 32 * (i - 8) + 31}]"
    ins = ", ".join(f'"{{{creg(i)}}}"(c{i})' for i in range(n32))
    out = ""
    per = min(MB, RP // 16)
    groups = [list(range(g0, g0 + per)) for g0 in range(0, MB, per)]
    for p, mbs in enumerate(groups):
        L = []
        far = any((mb - mbs[0]) * 16384 > 65535 for mb in mbs)
        if far:
            for nb in range(NB): L.append(f"v_add_u32 v{nb}, 0x10000, v{144 + nb}")
        for mb in mbs:
            for nb in range(NB):
                off = (mb - mbs[0]) * 16384
                t = nb * MB + mb
                src = f"a[{4 * t}:{4 * t + 3}]" if t < 64 else f"v[{192 + 4 * (t - 64)}:{192 + 4 * (t - 64) + 3}]"
                base, o = (f"v{nb}", off - 65536) if off > 65535 else (f"v{144 + nb}", off)
                L.append(f"ds_write_b128 {base}, {src} offset:{o}")
        L.append("s_waitcnt lgkmcnt(0)")
        text = "".join(f'        "{x}\\n\\t"\n' for x in L)
        clob = ", ".join([f'"v{i}"' for i in range(NB)] + ['"memory"'])
        out += f"""// GENERATED by tools/gen_gemm_asm.py - do not edit.  Block group {p} of the wide epilogue, tile {BM} x {BN}: {len(mbs) * NB} accumulator blocks -> LDS (f32).
__device__ __forceinline__ void gemm_asm16_store_{name}_p{p}({sig}, const u32x8& ad) {{
    asm volatile(
{text}        :
        : {ins}, "{{v[144:151]}}"(ad)
        : {clob});
}}
"""
    return out


TILES16 = {"256x256": (256, 256, 2, 2), "160x256": (160, 256, 1, 4), "320x256": (320, 256, 2, 2)}


def c_function(name, BM, BN, WGM, WGN, opt=None):
    lines, d = gen(BM, BN, WGM, WGN, opt)
    NT = d["NT"]
    text = "".join(f'        "{ins}\\n\\t"\n' for ins in lines)
    nacc_a = min(NT, 16)
    n32 = (nacc_a + 1) // 2                       # f32x32 AGPR outputs (two tiles each)
    sig = ", ".join(f"f32x32& c{i}" for i in range(n32))
    outs = ", ".join(f'"={{a[{32 * i}:{32 * i + 31}]}}"(c{i})' for i in range(n32))
    if NT > 16:
        sig += ", f32x32& cv0, f32x32& cv1"
        outs += ', "={v[112:143]}"(cv0), "={v[144:175]}"(cv1)'
    clob = [f'"v{i}"' for i in range(0, 72)] + ['"scc"', '"memory"']
    if d.get("REG"): clob = [f'"v{i}"' for i in range(0, 72)] + [f'"v{i}"' for i in range(112, 240)] + ['"scc"', '"memory"']
    return f"""// GENERATED by tools/gen_gemm_asm.py - do not edit.  {len(lines)} instructions: tile {BM} x {BN}, waves {WGM} x {WGN}.
__device__ __forceinline__ void gemm_asm_loop_{name}({sig}, const u32x16& rbase, const u32x16& dma0, const u32x2& dma1,
        const u32x4& ra, const u32x4& rw, int cnt, uint32_t ak, uint32_t bk, uint32_t ldsw) {{
    uint32_t asoff, bsoff;
    asm volatile(
{text}        : {outs}, [cnt] "+s"(cnt), [ak] "+s"(ak), [bk] "+s"(bk), [asoff] "=&s"(asoff), [bsoff] "=&s"(bsoff)
        : "{{v[72:87]}}"(rbase), "{{v[88:103]}}"(dma0), "{{v[104:105]}}"(dma1), [ra] "s"(ra), [rw] "s"(rw), [ldsw] "s"(ldsw)
        : {", ".join(clob)});
}}
"""


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "candle-video_amd", "csrc", "gemm_asm_loop.inc")
    opt = {}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--out": out = args.pop(0)
        else:
            k, v = a.split("="); opt[k] = v
    with open(out, "w") as f:
        for name, (BM, BN, WGM, WGN) in TILES.items():
            o = dict(opt)
            if str(o.get("stage", "dma")) == "reg" and (BM // 32 + BN // 32 != 16 or (BM // WGM // 32) * (BN // WGN // 32) > 16): o["stage"] = "dma"
            f.write(c_function(name.replace("x", "_"), BM, BN, WGM, WGN, o))
            f.write("\n")
        o16 = {k: v for k, v in opt.items() if k in ("abl", "bar_gap", "pgr", "rd_gaps", "b1_lag", "gb2_back", "rd2_step", "dma_last", "stagger", "trace")}
        for name, (BM, BN, WGM, WGN) in TILES16.items():
            f.write(c_function16(name.replace("x", "_"), BM, BN, WGM, WGN, o16))
            f.write("\n")
            f.write(store_functions16(name.replace("x", "_"), BM, BN, WGM, WGN))
            f.write("\n")
            plain = {k: v for k, v in o16.items() if k != "trace"}     # (trace builds: these two flavours are emitted without stamps)
            if name == "256x256":
                f.write(c_function16_conv(name.replace("x", "_"), BM, BN, WGM, WGN, plain))
                f.write("\n")
            if name == "160x256":
                f.write(c_function16(name.replace("x", "_"), BM, BN, WGM, WGN, dict(plain, res_rows=20)))
                f.write("\n")
    print("wrote", out, opt)


if __name__ == "__main__":
    main()
