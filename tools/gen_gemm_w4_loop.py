#!/usr/bin/env python3
"""Generator of med-ts-llm_amd/csrc/mtl_gemm_w4_loop.inc — the hand-placed k-loop of gemm_nt_w4_kernel (mtl_gemm.hip).

One 256 x 256 x 64 k-tile per loop body, 4 waves (2 x 2), one wave per SIMD, each wave a 128 x 128 sub-tile as 4 x 4
v_mfma_f32_32x32x16_bf16 accumulators (16 x f32x16 = 256 AGPRs, compiler-allocated "+a" operands). A k-tile is 64 MFMAs of 32 cycles each;
every other instruction of the k-tile (32 ds_read_b128, 16 LDS-DMA loads + their M0 updates, 3 barriers, the counted waits, the scalar
bookkeeping) sits in a FIXED slot between two MFMAs, at most two per gap, so the matrix pipe never waits for the issue of anything else.

Time structure of iteration t (LDS: two buffers per operand; buffer b = t & 1 holds k-tile t, k-tile t + 1 is landing in buffer b ^ 1):
  fragments live in two register sets: S0 = k 0..31 of the tile, S1 = k 32..63. At the top S0 holds tile t.
  MFMA  0..31 (S0) | B reads of S1, barrier 1 (B of buffer b is free) -> LDS-DMA of B(t + 2) into it, A reads of S1, barrier 2 -> LDS-DMA of A(t + 2)
  MFMA 32..63 (S1) | rest of the A DMA, vmcnt(those just issued) + barrier 3 (tile t + 1 has landed for everyone) -> S0 reads of tile t + 1
The loop is unrolled twice (buffer parity is an immediate in every ds_read offset and M0 base).

Tile chaining (a persistent workgroup's tiles i, i + 1, ...; %[flags] bit 1 = a next tile follows, bit 0 = the previous tile staged for this one): with a
next tile the two trailing iterations — which have nothing left to stage for tile i — stage k-tiles 0 and 1 of tile i + 1 (its row offsets: %[tabn]; the k
position wraps to 0 exactly like the per-XCD rotation's wrap), the statement's final vmcnt(0) covers them BEFORE the epilogue's stores are issued, and tile
i + 1 enters without any load or landing wait: its k-loop starts right after the epilogue's last store is ISSUED and the stores drain under its first
iterations (gfx950 retires loads and stores through one in-order counter, so a tile that stages after the stores waits for all of them). Needs an even
number of k-tiles (tile i + 1's k-tile 0 must land in buffer 0) and rot = 0.

Physical registers named by the asm (all listed as clobbers): v[112:239] fragments, v[240:247] per-k-step read bases,
s[72:75] / s[76:79] buffer descriptors of A / B, s[80:87] / s[88:95] the per-instruction row offsets, s96 loop counter, s97 / s70 k step (lo / sign word), s98 advances left, s99 advances until the k wrap, s71 the wrap's step,
s68 next-tile flag, s69 scratch.

usage: python tools/gen_gemm_w4_loop.py   (rewrites the .inc; the output is committed)"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "med-ts-llm_amd", "csrc", "mtl_gemm_w4_loop.inc")

FRAG0 = 112          # first fragment VGPR
KSB = 240            # v[240:243] A read bases per k-step, v[244:247] B read bases
A_BUF = 0x8000       # bytes between the two buffers of an operand
B_REGION = 0x10000   # B buffers start here
PIECE = 0x1000       # LDS bytes of one LDS-DMA instruction of the four waves (4 x 1 KiB)


def fa(s, ksl, mt):  # A-tile fragment (rows m): MFMA srcB
    r = FRAG0 + (((s * 2 + ksl) * 4 + mt) * 4)
    return f"v[{r}:{r + 3}]"


def fb(s, ksl, nt):  # B-tile fragment (rows n): MFMA srcA
    r = FRAG0 + 64 + (((s * 2 + ksl) * 4 + nt) * 4)
    return f"v[{r}:{r + 3}]"


def rd_a(s, ksl, mt, buf):
    return f"ds_read_b128 {fa(s, ksl, mt)}, v{KSB + s * 2 + ksl} offset:{buf * A_BUF + mt * 4096}"


def rd_b(s, ksl, nt, buf):
    return f"ds_read_b128 {fb(s, ksl, nt)}, v{KSB + 4 + s * 2 + ksl} offset:{buf * A_BUF + nt * 4096}"


def mfma(m, zero=False):
    s, ksl, nt, mt = m // 32, (m % 32) // 16, (m % 16) // 4, m % 4
    c = f"%[c{mt * 4 + nt}]"
    return f"v_mfma_f32_32x32x16_bf16 {c}, {fb(s, ksl, nt)}, {fa(s, ksl, mt)}, {'0' if zero else c}"


def dma(op, i, pol=""):
    if op == "a":
        return f"buffer_load_dwordx4 %[voa], s[72:75], s{80 + i} offen{pol} lds"
    return f"buffer_load_dwordx4 %[vob], s[76:79], s{88 + i} offen{pol} lds"


def policy(var):
    return " nt" if "n" in var else (" sc1" if "c" in var else (" sc0 sc1" if "C" in var else ""))


# A schedule = the MFMA slots (0..63; an instruction is placed AFTER that MFMA) of the k-tile's other work.
SCHED = {
    # round-6 first version (kept for the A/B record)
    "s0": dict(b1=0, bar1=12, dma_b=14, a1=14, bar2=29, dma_a=31, wait=43, s0=44, s0_per=1, final_lgkm=0),
    # every wait at least 8 MFMAs behind the last instruction it waits for; the S0 reads of k 16..31 may stay out across the loop end
    # (the next iteration's first wait, lgkmcnt(0) in front of barrier 1, precedes their first use at MFMA 16)
    "s1": dict(b1=0, bar1=15, dma_b=17, a1=16, bar2=31, dma_a=33, wait=46, s0=47, s0_per=1, final_lgkm=8),
    # the landing wait as late as the S0 reads allow (two reads per gap)
    "s2": dict(b1=0, bar1=15, dma_b=17, a1=16, bar2=31, dma_a=33, wait=53, s0=54, s0_per=2, final_lgkm=8),
    # the 16 LDS-DMA instructions spread over the iteration (one per four / three MFMAs) instead of two bursts of one per two
    "s3": dict(b1=0, bar1=14, dma_b=16, a1=16, bar2=31, dma_a=34, dma_step=4, wait=47, s0=48, s0_per=1, final_lgkm=8),
    "s4": dict(b1=0, bar1=14, dma_b=16, a1=16, bar2=31, dma_a=33, dma_step=3, wait=47, s0=48, s0_per=1, final_lgkm=8),
}
PRODUCT_SCHED = "s3"


_LBL = [0]


def body(b, sched, var="", first=False, chained_entry=False):
    """one k-tile, buffer parity b. Returns the instruction list. first: the tile's first k-tile — its first 16 MFMAs (one per accumulator) take the
    constant 0 as C, so the accumulators are write-only operands of the asm statement and nobody zero-fills 256 AGPRs per tile. `var`: ablation letters for the DIAGNOSTIC variants (timing only, wrong results):
    D = no in-loop LDS-DMA, B = no barriers, R = no fragment reads, W = no waits, V = no landing (vmcnt) wait, L = every DMA re-reads the same 8 rows"""
    sc = SCHED[sched]
    fill = {m: [] for m in range(64)}          # instructions placed AFTER MFMA m
    n_dma = 0
    loc = "L" in var
    # ---- S1 reads of B (tile t), then barrier 1
    order_b1 = [(1, ksl, nt) for ksl in range(2) for nt in range(4)]
    for j, (s, ksl, nt) in enumerate(order_b1):
        fill[sc["b1"] + j].append(rd_b(s, ksl, nt, b))
    # scalar bookkeeping of the iteration: k step of this iteration's DMA (k-tile t + 2). s99 reaches 0 where the k position wraps back to k-tile 0
    # (per-XCD rotation; with rot = 0 that is iteration nkt - 2, the first of the two trailing ones)
    fill[8] += ["s_sub_i32 s99, s99, 1", "s_cmp_eq_u32 s99, 0", "s_cselect_b32 s97, s71, 128"]
    # the two trailing iterations (s98 <= 0) have nothing left to stage for THIS tile. With a next tile (s68 != 0) they stage its k-tiles 0 / 1; without,
    # their 16 LDS-DMA instructions keep their slots (the counted waits stay static) but run with num_records = 0 — out of range, no memory access —
    # instead of re-fetching the last k-tile (2 x 64 KB per tile through the L2, and a landing wait for them at the loop end)
    fill[9] += ["s_cmp_gt_i32 s98, 0", "s_cselect_b32 s69, 1, s68", "s_cmp_lg_u32 s69, 0"]
    fill[10] += ["s_cselect_b32 s97, 0, 0" if loc else "s_cselect_b32 s97, s97, 0", "s_cselect_b32 s74, -1, 0", "s_cselect_b32 s78, -1, 0"]
    fill[11] += ["s_sub_i32 s98, s98, 1", "s_ashr_i32 s70, s97, 31"]
    # the iteration that wraps with a next tile: from here on the row offsets are the NEXT tile's (before this iteration's first DMA, after the previous one's last)
    _LBL[0] += 1
    lbl = f"L_w4_keep{_LBL[0]}_%="
    fill[12] += ["s_cmp_eq_u32 s99, 0", "s_cselect_b32 s69, s68, 0", "s_cmp_eq_u32 s69, 0", f"s_cbranch_scc1 {lbl}"]
    fill[12] += [f"v_readlane_b32 s{80 + j}, %[tabn], {j}" for j in range(16)] + [lbl + ":"]
    fill[sc["bar1"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier", "s_add_u32 s76, s76, s97", "s_addc_u32 s77, s77, s70"]
    # ---- DMA of B(t + 2) / A(t + 2) into buffer b: instruction i in slot start + step * i, its M0 (absolute) set one gap earlier
    # ---- S1 reads of A (tile t), then barrier 2
    order_a1 = [(1, ksl, mt) for ksl in range(2) for mt in range(4)]
    for j, (s, ksl, mt) in enumerate(order_a1):
        fill[sc["a1"] + j].append(rd_a(s, ksl, mt, b))
    assert sc["a1"] + 7 < sc["bar2"] and sc["bar1"] < sc["dma_b"] - 1 and sc["bar2"] < sc["dma_a"] - 1 and sc["bar1"] > 12
    fill[sc["bar2"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier", "s_add_u32 s72, s72, s97", "s_addc_u32 s73, s73, s70"]
    wait_at = sc["wait"]
    step = sc.get("dma_step", 2)
    slots = sorted([(sc["dma_b"] + step * i, "b", i) for i in range(8)] + [(sc["dma_a"] + step * i, "a", i) for i in range(8)])
    assert len({m for m, _, _ in slots}) == 16 and slots[-1][0] <= 63, slots
    for m, op, i in slots:
        base = (B_REGION if op == "b" else 0) + b * A_BUF + i * PIECE
        fill[m - 1].append(f"s_add_u32 m0, %[dma], {base}")
        fill[m].insert(0, dma(op, 0 if loc else i, policy(var)))          # (ahead of a later DMA's M0 update that shares the gap)
        if m < wait_at:
            n_dma += 1
    # ---- tile t + 1 has landed: everything older than this iteration's loads issued so far
    # (a chained entry's tile t + 1 = k-tile 1 landed before the previous tile's epilogue: no wait — it would wait for that epilogue's stores)
    fill[wait_at] = ([] if chained_entry else [f"s_waitcnt vmcnt({n_dma})"]) + ["s_barrier"] + fill[wait_at]
    # ---- S0 reads of tile t + 1 from buffer b ^ 1, in the order the next iteration consumes them
    order0 = []
    for ksl in range(2):
        order0.append(("b", 0, ksl, 0))
        order0 += [("a", 0, ksl, mt) for mt in range(4)]
        order0 += [("b", 0, ksl, nt) for nt in range(1, 4)]
    for j, (op, s, ksl, x) in enumerate(order0):
        m = sc["s0"] + j // sc["s0_per"]
        assert wait_at < m <= 63 and m >= 32      # (S0 registers are free once MFMA 31 has issued)
        fill[m].append(rd_a(s, ksl, x, b ^ 1) if op == "a" else rd_b(s, ksl, x, b ^ 1))
    out = []
    for m in range(64):
        if "S" in var:      # ablation: the same FLOPs as two v_mfma_f32_16x16x32_bf16 on physical AGPR quads (garbage results: what the MFMA shape does to time / clock)
            s_, ksl, nt, mt = m // 32, (m % 32) // 16, (m % 16) // 4, m % 4
            for half in range(2):
                q = ((m % 32) * 2 + half) * 4
                out.append(f"v_mfma_f32_16x16x32_bf16 a[{q}:{q + 3}], {fb(s_, ksl, nt)}, {fa(s_, ksl, mt)}, a[{q}:{q + 3}]")
        else:
            out.append(mfma(m, first and m < 16))
        for ins in fill[m]:
            if "D" in var and ins.startswith("buffer_load"):
                continue
            if "B" in var and ins == "s_barrier":
                continue
            if "R" in var and ins.startswith("ds_read"):
                continue
            if "W" in var and ins.startswith("s_waitcnt"):
                continue
            if "V" in var and ins.startswith("s_waitcnt vmcnt"):
                continue
            out.append(ins)
    out.append(f"s_waitcnt lgkmcnt({sc['final_lgkm']})")
    return out


def setup():
    """descriptors, row offsets, read bases, k bookkeeping — common to a cold and a chained entry"""
    o = ["s_barrier",
         "s_mov_b64 s[72:73], %[pa]", "s_mov_b32 s74, -1", "s_mov_b32 s75, 0x20000",
         "s_mov_b64 s[76:77], %[pb]", "s_mov_b32 s78, -1", "s_mov_b32 s79, 0x20000"]
    for j in range(16):
        o.append(f"v_readlane_b32 s{80 + j}, %[tab], {j}")
    for ks in range(4):
        o += [f"v_xor_b32 v{KSB + ks}, {2 * ks}, %[xa]", f"v_xor_b32 v{KSB + 4 + ks}, {2 * ks}, %[xb]"]
    for ks in range(4):
        o += [f"v_lshl_add_u32 v{KSB + ks}, v{KSB + ks}, 4, %[rba]", f"v_lshl_add_u32 v{KSB + 4 + ks}, v{KSB + 4 + ks}, 4, %[rbb]"]
    # k position of the LDS-DMA stream: k-tile (j + rot) mod nkt for the j-th tile staged (per-XCD rotation; rot = 0: plain order).
    # s98 = advances left, s99 = advances until the wrap back to k-tile 0, s71 = the wrap's byte step -(nkt - 1) * 128
    o += ["s_mov_b32 s96, %[nkt]", "s_sub_i32 s98, %[nkt], 2", "s_sub_i32 s99, %[nkt], %[rot]",
          "s_sub_i32 s71, 1, %[nkt]", "s_lshl_b32 s71, s71, 7", "v_readfirstlane_b32 s69, %[flags]", "s_and_b32 s68, s69, 2",
          "s_lshl_b32 s97, %[rot], 7", "s_add_u32 s72, s72, s97", "s_addc_u32 s73, s73, 0", "s_add_u32 s76, s76, s97", "s_addc_u32 s77, s77, 0", "s_nop 4"]
    return o


def s0_reads():
    o = []
    for ksl in range(2):
        for x in range(4):
            o += [rd_b(0, ksl, x, 0), rd_a(0, ksl, x, 0)]
    return o + ["s_waitcnt lgkmcnt(0)"]


def cold_stage():
    """k-tiles 0 and 1 of the tile, staged and waited for here (first tile of a workgroup, or odd k-tile counts)"""
    def tile(buf):
        t = [f"s_add_u32 m0, %[dma], {buf * A_BUF}", "s_nop 0"]
        for i in range(8):
            t.append(dma("a", i))
            if i < 7:
                t += [f"s_add_u32 m0, m0, {PIECE}", "s_nop 0"]
        t += [f"s_add_u32 m0, %[dma], {B_REGION + buf * A_BUF}", "s_nop 0"]
        for i in range(8):
            t.append(dma("b", i))
            if i < 7:
                t += [f"s_add_u32 m0, m0, {PIECE}", "s_nop 0"]
        return t
    o = tile(0)
    # k-tile 1 (a tile with one k-tile: out of range, nothing fetched)
    o += ["s_sub_i32 s99, s99, 1", "s_cmp_eq_u32 s99, 0", "s_cselect_b32 s97, s71, 128",
          "s_cmp_gt_i32 %[nkt], 1", "s_cselect_b32 s97, s97, 0", "s_cselect_b32 s74, -1, 0", "s_cselect_b32 s78, -1, 0", "s_ashr_i32 s70, s97, 31",
          "s_add_u32 s72, s72, s97", "s_addc_u32 s73, s73, s70", "s_add_u32 s76, s76, s97", "s_addc_u32 s77, s77, s70"]
    o += tile(1)
    o += ["s_waitcnt vmcnt(16)", "s_barrier"]
    return o


def chained_stage():
    """the previous tile's trailing iterations staged k-tiles 0 / 1 and its final vmcnt(0) + this entry's barrier published them: only the
    stream position moves to k-tile 1 (rot = 0, nkt even >= 2)"""
    return ["s_sub_i32 s99, s99, 1", "s_add_u32 s72, s72, 128", "s_addc_u32 s73, s73, 0", "s_add_u32 s76, s76, 128", "s_addc_u32 s77, s77, 0"]


def program(sched=PRODUCT_SCHED, var=""):
    _LBL[0] = 0
    dec_end = ["s_sub_u32 s96, s96, 1", "s_cmp_eq_u32 s96, 0", "s_cbranch_scc1 L_w4_end_%="]
    lines = setup()
    lines += ["s_and_b32 s69, s69, 1", "s_cmp_eq_u32 s69, 0", "s_cbranch_scc0 L_w4_chained_%="]       # (%[flags] is wave-uniform but arrives in a VGPR)
    lines += cold_stage() + s0_reads()
    lines += body(0, sched, var, first=True)
    lines += dec_end + ["s_branch L_w4_top_%="]
    lines.append("L_w4_chained_%=:")
    lines += chained_stage() + s0_reads()
    lines += body(0, sched, var, first=True, chained_entry=True)
    lines += dec_end
    lines.append("L_w4_top_%=:")
    lines += body(1, sched, var)
    lines += dec_end
    lines += body(0, sched, var)
    lines += ["s_sub_u32 s96, s96, 1", "s_cmp_lg_u32 s96, 0", "s_cbranch_scc1 L_w4_top_%="]
    lines.append("L_w4_end_%=:")
    # this tile's stray / the next tile's first loads land before the LDS is touched again (and before the epilogue's stores queue up behind them);
    # MFMA results settle before the compiler reads them
    lines += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 15"]
    return lines


# MTL_W4_LOOP_ASM_V1 .. (diagnostic builds, -DMTL_DIAG_W4VAR): (schedule, ablation letters)
VARIANTS = [("s3", "S"), ("s3", "D"), ("s3", "DBRW"), ("s3", "SDBRW"), ("s3", "L")]


def emit(f, name, lines):
    f.write(f"#define {name} \\\n")
    for l in lines:
        f.write(f'    "{l}\\n\\t" \\\n')
    f.write('    ""\n')


def main():
    lines = program()
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4_loop.py — do not edit by hand. The k-loop of gemm_nt_w4_kernel as ONE asm statement.\n")
        f.write(f"// {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs, {len(lines)} instructions.\n")
        emit(f, "MTL_W4_LOOP_ASM", lines)
        f.write("#if defined(MTL_DIAG_W4VAR) || defined(MTL_W4_ASM_SELECT)      // ablations for timing (WRONG results): what the in-loop DMA / barriers / reads / waits cost\n")
        for i, (sc, v) in enumerate(VARIANTS):
            f.write(f"// V{i + 1}: schedule {sc}, ablation '{v}'\n")
            emit(f, f"MTL_W4_LOOP_ASM_V{i + 1}", program(sc, v))
        f.write("#endif\n")
        clob = [f'"v{r}"' for r in range(FRAG0, KSB + 8)] + [f'"s{r}"' for r in range(68, 100)]
        f.write("#define MTL_W4_LOOP_CLOBBERS " + ", ".join(clob) + ', "scc", "memory"\n')
    print("wrote", OUT, len(lines), "instructions")


if __name__ == "__main__":
    main()
