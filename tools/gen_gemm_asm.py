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
    print("wrote", out, opt)


if __name__ == "__main__":
    main()
