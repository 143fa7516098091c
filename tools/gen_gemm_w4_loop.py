#!/usr/bin/env python3
"""Generator of med-ts-llm_amd/csrc/mtl_gemm_w4_loop.inc — the hand-placed k-loop of gemm_nt_w4_kernel (mtl_gemm.hip).

One 256 x 256 x 64 k-tile per loop body, 4 waves (2 x 2), one wave per SIMD, each wave a 128 x 128 sub-tile as 8 x 8 output tiles of
v_mfma_f32_16x16x32_bf16 (64 accumulator quads = the 256 AGPRs, PHYSICAL: tile (mi, ni) = a[4 (8 mi + ni) .. + 3]; the kernel's asm operands
"={a[16 k : 16 k + 15]}" tell the compiler). A k-tile is 128 MFMAs of 16 cycles each, two per SLOT (the slot structure of the first, 32 x 32 x 16
version — same cycles per FLOP, but the chip clocks 5 - 9 % lower under that shape: profiles/r06_gemm_w4_experiments.txt section 15); every other
instruction of the k-tile (32 ds_read_b128, 16 LDS-DMA loads + their M0 updates, 3 barriers, the counted waits, the scalar bookkeeping) sits in
a FIXED slot, so the matrix pipe never waits for the issue of anything else.

Time structure of iteration t (LDS: two buffers per operand; buffer b = t & 1 holds k-tile t, k-tile t + 1 is landing in buffer b ^ 1):
  fragments live in two register sets: S0 = k-step 0 (k 0..31 of the tile), S1 = k-step 1 (k 32..63); per set 8 A row blocks + 8 B column blocks of 4 VGPRs.
  At the top S0 holds tile t.
  slots  0..31 (S0: 64 MFMAs) | B reads of S1, barrier 1 (B of buffer b is free) -> LDS-DMA of B(t + 2) into it, A reads of S1, barrier 2 -> LDS-DMA of A(t + 2)
  slots 32..63 (S1: 64 MFMAs) | rest of the A DMA, vmcnt(those just issued) + barrier 3 (tile t + 1 has landed for everyone) -> S0 reads of tile t + 1
  MFMA order of a k-step: row blocks 0..3 against the eight column blocks (snaking: consecutive MFMAs share a fragment), then row blocks 4..7
  (every accumulator once per k-step, 64 MFMAs apart).
The loop is unrolled twice (buffer parity is an immediate in every ds_read offset and M0 base).

Tile chaining (a persistent workgroup's tiles i, i + 1, ...; %[flags] bit 1 = a next tile follows, bit 0 = the previous tile staged for this one): with a
next tile the two trailing iterations — which have nothing left to stage for tile i — stage k-tiles 0 and 1 of tile i + 1 (its row offsets: %[tabn]; the k
position wraps to 0 exactly like the per-XCD rotation's wrap), the statement's final vmcnt(0) covers them BEFORE the epilogue's stores are issued, and tile
i + 1 enters without any load or landing wait: its k-loop starts right after the epilogue's last store is ISSUED and the stores drain under its first
iterations (gfx950 retires loads and stores through one in-order counter, so a tile that stages after the stores waits for all of them). Needs an even
number of k-tiles (tile i + 1's k-tile 0 must land in buffer 0) and rot = 0.

Physical registers named by the asm (all listed as clobbers): v[112:239] fragments, v[240:241] / v[244:245] per-k-step read bases of A / B, a[0:255] the accumulators (operands),
s[72:75] / s[76:79] buffer descriptors of A / B, s[80:87] / s[88:95] the per-instruction row offsets, s96 loop counter, s97 / s70 k step (lo / sign word), s98 advances left, s99 advances until the k wrap, s71 the wrap's step,
s66 the item's k-tile count, s67 its k byte offset, s68 next-tile flag, s69 scratch.

usage: python tools/gen_gemm_w4_loop.py   (rewrites the .inc; the output is committed)"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "med-ts-llm_amd", "csrc", "mtl_gemm_w4_loop.inc")

FRAG0 = 112          # first fragment VGPR
KSB = 240            # v[240:243] A read bases per k-step, v[244:247] B read bases
A_BUF = 0x8000       # bytes between the two buffers of an operand
B_REGION = 0x10000   # B buffers start here
PIECE = 0x1000       # LDS bytes of one LDS-DMA instruction of the four waves (4 x 1 KiB)


def fa(s, blk):      # A-tile fragment of k-step s (k 32 s .. 32 s + 31), 16-row block blk of the wave's 128 rows: MFMA srcB
    r = FRAG0 + (s * 16 + blk) * 4
    return f"v[{r}:{r + 3}]"


def fb(s, blk):      # B-tile fragment (rows n): MFMA srcA
    r = FRAG0 + (s * 16 + 8 + blk) * 4
    return f"v[{r}:{r + 3}]"


def rd_a(s, blk, buf):
    return f"ds_read_b128 {fa(s, blk)}, v{KSB + s} offset:{buf * A_BUF + blk * 2048}"


def rd_b(s, blk, buf):
    # column blocks come in PAIRS: lane (l15, g) of blocks 2t / 2t + 1 reads B-tile row 32 t + 8 (l15 >> 2) + (l15 & 3) [+ 4], so that the lane's 4 + 4 output
    # columns of the pair are the 8 consecutive columns 32 t + 8 g .. + 7 (16-byte bf16 stores in the epilogue). The lane part sits in the base register.
    return f"ds_read_b128 {fb(s, blk)}, v{KSB + 4 + s} offset:{buf * A_BUF + (blk >> 1) * 4096 + (blk & 1) * 512}"


def acc(mi, ni):     # accumulator of output tile (row block mi, column block ni) of the wave's 8 x 8: physical AGPR quad
    q = 4 * (mi * 8 + ni)
    return f"a[{q}:{q + 3}]"


ORDER = ["snake"]     # MFMA order inside a k-step: "snake" (product) / "colmajor" / "amajor" (ablations; set by program())


def mfma_tile(j):    # j-th MFMA of a k-step -> (mi, ni): row blocks 0..3 against all eight column blocks first (snaking), then row blocks 4..7
    h, jj = (0, j) if j < 32 else (4, j - 32)
    if ORDER[0] == "amajor":          # row block by row block: consecutive MFMAs share the A fragment
        return h + jj // 8, jj % 8
    mi, ni = jj % 4, jj // 4            # "colmajor": column block by column block, row blocks always upwards
    if ORDER[0] == "snake" and ni % 2:  # ... the row blocks alternately up and down: consecutive MFMAs always share one fragment
        mi = 3 - mi
    return h + mi, ni


def mfma(n, zero=False):     # n-th of the k-tile's 128 MFMAs
    s, (mi, ni) = n // 64, mfma_tile(n % 64)
    return f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {fb(s, ni)}, {fa(s, mi)}, {'0' if zero else acc(mi, ni)}"


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
    "s1": dict(b1=0, bar1=15, dma_b=17, a1=16, bar2=31, dma_a=33, wait=46, s0=47, s0_per=1, final_lgkm=4),
    # the landing wait as late as the S0 reads allow (two reads per gap)
    "s2": dict(b1=0, bar1=15, dma_b=17, a1=16, bar2=31, dma_a=33, wait=53, s0=54, s0_per=2, final_lgkm=4),
    # the 16 LDS-DMA instructions spread over the iteration (one per four / three MFMAs) instead of two bursts of one per two
    "s3": dict(b1=0, bar1=14, dma_b=16, a1=16, bar2=31, dma_a=34, dma_step=4, wait=47, s0=48, s0_per=1, final_lgkm=4),
    "s4": dict(b1=0, bar1=14, dma_b=16, a1=16, bar2=31, dma_a=33, dma_step=3, wait=47, s0=48, s0_per=1, final_lgkm=4),
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
    for j in range(8):
        fill[sc["b1"] + j].append(rd_b(1, j, b))
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
    for j in range(8):
        fill[sc["a1"] + j].append(rd_a(1, j, b))
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
    for j, (op, x) in enumerate(S0_ORDER):
        m = sc["s0"] + j // sc["s0_per"]
        assert wait_at < m <= 63 and m >= 32      # (S0 registers are free once slot 31 has issued)
        fill[m].append(rd_a(0, x, b ^ 1) if op == "a" else rd_b(0, x, b ^ 1))
    out = []
    for m in range(64):
        out.append(mfma(2 * m, first and m < 32))
        if "P" not in var and len(fill[m]) >= 2:      # a slot's instructions go into BOTH of its MFMA gaps (a 16-cycle MFMA leaves 12 issue cycles per gap); ablation 'P': all behind the pair
            k = 1 if len(fill[m]) == 2 else (len(fill[m]) + 1) // 2
            early, fill[m] = fill[m][:k], fill[m][k:]
            if not any(i.endswith(":") or i.startswith("s_cbranch") for i in early + fill[m]):
                out += early
            else:
                fill[m] = early + fill[m]
        out.append(mfma(2 * m + 1, first and m < 32))
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
    for ks in range(2):        # k-step ks: the lane's 16-byte chunk 4 ks + (lane >> 4), XOR the row's swizzle
        o += [f"v_xor_b32 v{KSB + ks}, {4 * ks}, %[xa]", f"v_xor_b32 v{KSB + 4 + ks}, {4 * ks}, %[xb]"]
    for ks in range(2):
        o += [f"v_lshl_add_u32 v{KSB + ks}, v{KSB + ks}, 4, %[rba]", f"v_lshl_add_u32 v{KSB + 4 + ks}, v{KSB + 4 + ks}, 4, %[rbb]"]
    # k position of the LDS-DMA stream: k-tile (j + rot) mod nkt for the j-th tile staged (per-XCD rotation; rot = 0: plain order).
    # s98 = advances left, s99 = advances until the wrap back to k-tile 0, s71 = the wrap's byte step -(nkt - 1) * 128
    # %[flags] (wave-uniform, in a VGPR): bits 1:0 = chaining flags, bits 15:2 = the item's k-tile count, bits 31:16 = its first k-tile (split-K items
    # start inside the K range: k-slab of a (tile, slab) work item; 0 otherwise). s66 = nkt, s67 = byte offset of the first k-tile.
    o += ["v_readfirstlane_b32 s69, %[flags]", "s_bfe_u32 s66, s69, 0xe0002", "s_lshr_b32 s67, s69, 16", "s_lshl_b32 s67, s67, 7", "s_and_b32 s68, s69, 2",
          "s_add_u32 s72, s72, s67", "s_addc_u32 s73, s73, 0", "s_add_u32 s76, s76, s67", "s_addc_u32 s77, s77, 0",
          "s_mov_b32 s96, s66", "s_sub_i32 s98, s66, 2", "s_sub_i32 s99, s66, %[rot]",
          "s_sub_i32 s71, 1, s66", "s_lshl_b32 s71, s71, 7",
          "s_lshl_b32 s97, %[rot], 7", "s_add_u32 s72, s72, s97", "s_addc_u32 s73, s73, 0", "s_add_u32 s76, s76, s97", "s_addc_u32 s77, s77, 0", "s_nop 4"]
    return o


# S0 (k-step 0) fragments in the order the next iteration consumes them: row blocks 0..3 and column block 0 at once, column block ni from MFMA 4 ni on,
# row blocks 4..7 from MFMA 32 (= slot 16) on — those four may stay out across the loop end (final_lgkm = 4: the next wait, lgkmcnt(0) in front of
# barrier 1 at slot 14, precedes their first use)
S0_ORDER = [("a", 0), ("a", 1), ("a", 2), ("a", 3)] + [("b", x) for x in range(8)] + [("a", 4), ("a", 5), ("a", 6), ("a", 7)]


def s0_reads():
    return [rd_a(0, x, 0) if op == "a" else rd_b(0, x, 0) for op, x in S0_ORDER] + ["s_waitcnt lgkmcnt(0)"]


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
          "s_cmp_gt_i32 s66, 1", "s_cselect_b32 s97, s97, 0", "s_cselect_b32 s74, -1, 0", "s_cselect_b32 s78, -1, 0", "s_ashr_i32 s70, s97, 31",
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
    ORDER[0] = "colmajor" if "k" in var else ("amajor" if "j" in var else "snake")       # product: snake (cold probe: 4096^3 101.9 -> 99.1 us, qkv 294.7 -> 290.6)
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
VARIANTS = [("s3", "k"), ("s3", "j"), ("s3", "DBRW"), ("s3", "P"), ("s3", "L")]


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
        clob = [f'"v{r}"' for r in range(FRAG0, KSB + 8)] + [f'"s{r}"' for r in range(66, 100)]
        f.write("#define MTL_W4_LOOP_CLOBBERS " + ", ".join(clob) + ', "scc", "memory"\n')
    print("wrote", OUT, len(lines), "instructions")


if __name__ == "__main__":
    main()
