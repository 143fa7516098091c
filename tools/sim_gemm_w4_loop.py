#!/usr/bin/env python3
"""Scalar-side simulator of the generated k-loop of gemm_nt_w4_kernel (med-ts-llm_amd/csrc/mtl_gemm_w4_loop.inc): runs the asm text's SALU
instructions, branches and M0 updates for one wave and records, in program order, every LDS-DMA (which operand, which row piece, which k-tile,
which LDS address), every fragment read (operand, k-step, buffer) and every barrier / wait. `check()` then asserts the protocol the kernel relies on:
  * every k-tile of the tile is staged exactly once per operand piece, k-tile j of the stream into buffer j & 1, at the piece's LDS address;
  * iteration t reads buffer t & 1, after a landing wait + barrier that covers tile t's loads; a buffer region is re-staged only after the barrier
    that follows the last read of it; the counted vmcnt / lgkmcnt waits leave exactly the intended instructions outstanding;
  * the physical k order is the per-XCD rotation (j + rot) mod nkt; the loads of the two trailing iterations run out of range (num_records = 0);
  * every accumulator's first MFMA takes the constant 0 as C (the asm statement's accumulators are write-only operands);
  * tile chaining (%[flags]): with a next tile the trailing iterations stage ITS k-tiles 0 / 1 (row table %[tabn]) into buffers 0 / 1 and the final
    vmcnt(0) covers them; a chained entry issues no load and no landing wait before its iteration 1.
Used by tests/test_w4_loop_sim.py (CPU) — the asm is otherwise only exercised on the GPU box."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "med-ts-llm_amd", "csrc", "mtl_gemm_w4_loop.inc")


def load(name="MTL_W4_LOOP_ASM", path=INC):
    text = open(path).read()
    m = re.search(r"#define " + name + r" \\\n(.*?)\n    \"\"\n", text, re.S)
    assert m, name
    return [re.match(r'\s*"(.*?)\\n\\t" \\', l).group(1) for l in m.group(1).splitlines()]


M32 = 0xFFFFFFFF


def run(lines, nkt, rot=0, dma_base=0x400, max_steps=2_000_000, flags=0, k0=0):
    """returns the event list. Operand values: %[pa] = 0x1000_0000_0000, %[pb] = 0x2000_0000_0000 (so that base - start = k byte offset)."""
    PA, PB = 0x100000000000, 0x200000000000
    s = {}                     # SGPRs (+ 'm0', 'scc')
    sym = {"%[rot]": rot, "%[dma]": dma_base, "%[flags]": flags | (nkt << 2) | (k0 << 16)}      # the kernel's packed word: chain flags, k-tile count, first k-tile
    table = {}                 # row-offset SGPR -> "tab" (this tile's rows) / "tabn" (the next tile's)
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    ev = []

    def val(tok):
        tok = tok.strip()
        if tok in sym:
            return sym[tok] & M32
        if tok == "m0":
            return s["m0"]
        if re.fullmatch(r"s\d+", tok):
            return s[tok]
        return int(tok, 0) & M32

    pc, steps = 0, 0
    while pc < len(lines):
        steps += 1
        assert steps < max_steps, "runaway loop"
        ins = lines[pc]
        pc += 1
        if ins.endswith(":"):
            continue
        op, _, rest = ins.partition(" ")
        a = [x.strip() for x in rest.split(",")] if rest else []
        if op == "s_mov_b32":
            s[a[0]] = val(a[1])
        elif op == "s_mov_b64":
            lo = int(re.match(r"s\[(\d+):", a[0]).group(1))
            v = {"%[pa]": PA, "%[pb]": PB}[a[1]]
            s[f"s{lo}"], s[f"s{lo + 1}"] = v & M32, v >> 32
        elif op in ("s_add_u32", "s_sub_u32", "s_sub_i32", "s_add_i32"):
            x, y = val(a[1]), val(a[2])
            r = x + y if op.startswith("s_add") else x - y
            if op.endswith("u32"):
                s["scc"] = int(r > M32 or r < 0)
            s[a[0]] = r & M32
        elif op == "s_addc_u32":
            r = val(a[1]) + val(a[2]) + s["scc"]
            s["scc"] = int(r > M32)
            s[a[0]] = r & M32
        elif op == "s_bfe_u32":
            c = val(a[2])
            s[a[0]] = (val(a[1]) >> (c & 31)) & ((1 << ((c >> 16) & 0x7f)) - 1)
        elif op == "s_lshr_b32":
            s[a[0]] = val(a[1]) >> val(a[2])
        elif op == "s_and_b32":
            s[a[0]] = val(a[1]) & val(a[2])
        elif op == "v_readfirstlane_b32":
            s[a[0]] = val(a[1])
        elif op == "v_readlane_b32":
            table[a[0]] = a[1].strip("%[]")
            assert int(a[0][1:]) - 80 == int(a[2]), ins
        elif op == "s_lshl_b32":
            s[a[0]] = (val(a[1]) << val(a[2])) & M32
        elif op == "s_ashr_i32":
            x = val(a[1])
            x = x - (1 << 32) if x >> 31 else x
            s[a[0]] = (x >> val(a[2])) & M32
        elif op in ("s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_gt_i32"):
            x, y = val(a[0]), val(a[1])
            if op == "s_cmp_gt_i32":
                x, y = (x - (1 << 32) if x >> 31 else x), (y - (1 << 32) if y >> 31 else y)
                s["scc"] = int(x > y)
            else:
                s["scc"] = int((x == y) == (op == "s_cmp_eq_u32"))
        elif op == "s_cselect_b32":
            s[a[0]] = val(a[1]) if s["scc"] else val(a[2])
        elif op == "s_branch":
            pc = labels[a[0]]
        elif op in ("s_cbranch_scc1", "s_cbranch_scc0"):
            if s["scc"] == (op == "s_cbranch_scc1"):
                pc = labels[a[0]]
        elif op == "buffer_load_dwordx4":
            which = "a" if a[0] == "%[voa]" else "b"
            lo = 72 if which == "a" else 76
            base = s[f"s{lo}"] | (s[f"s{lo + 1}"] << 32)
            off = base - (PA if which == "a" else PB) - k0 * 128              # relative to the item's first k-tile
            piece = int(re.match(r"s(\d+)", a[2].split()[0]).group(1)) - (80 if which == "a" else 88)
            assert off % 128 == 0 and 0 <= off // 128 < nkt, (which, off, nkt)
            ev.append(("dma", which, piece, off // 128, s["m0"] - dma_base, s[f"s{lo + 2}"] != 0, table[a[2].split()[0]]))   # num_records != 0 (0 = out of range: no memory access), row table
        elif op == "ds_read_b128":
            reg = int(re.match(r"v\[(\d+):", a[0]).group(1))
            addr_reg = int(re.match(r"v(\d+)", a[1].split()[0]).group(1))
            offset = int(re.search(r"offset:(\d+)", ins).group(1))
            which = "a" if addr_reg < 244 else "b"
            off = offset % 0x8000
            blk = off // 2048 if which == "a" else (off // 4096) * 2 + (off % 4096) // 512        # (B column blocks are read in pairs: 32 t rows + 4 rows for the odd one)
            assert (off % 2048 == 0) if which == "a" else (off % 4096 in (0, 512)), ins
            ev.append(("read", which, (addr_reg - 240) % 4, offset // 0x8000, blk, reg))      # k-step, buffer, block, destination
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                ev.append(("vmcnt", int(m.group(1))))
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                ev.append(("lgkmcnt", int(m.group(1))))
        elif op == "s_barrier":
            ev.append(("barrier",))
        elif op == "v_mfma_f32_16x16x32_bf16":
            srcs = [int(re.match(r"v\[(\d+):", x).group(1)) for x in a[1:3]]
            ev.append(("mfma", a[0], srcs[0], srcs[1], a[3]))
        elif op in ("s_nop", "v_xor_b32", "v_lshl_add_u32"):
            pass
        else:
            raise AssertionError(f"unknown instruction: {ins}")
    return ev


def check(ev, nkt, rot=0, has_prev=False, has_next=False):
    """assert the staging / reading protocol (see the module docstring). has_prev: the previous tile of the workgroup staged this tile's k-tiles 0 / 1
    (they are landed and published at entry); has_next: this tile's two trailing iterations stage k-tiles 0 / 1 of the next tile from ITS row table"""
    assert not (has_prev or has_next) or (rot == 0 and nkt % 2 == 0 and nkt >= 2)
    FA0, FB0 = 112, 176
    landed = {}                 # (operand, buffer, piece) -> k-tile whose data is visible to every wave (after vmcnt + barrier)
    pending = []                # DMAs issued and not yet known complete: (operand, buffer, piece, ktile)
    waited = []                 # DMAs complete for this wave, not yet behind a barrier
    reads_out = []              # fragment reads issued, not yet known complete: (operand, buffer, reg)
    frag = {}                   # fragment register -> (operand, k-tile, k-step) once its read has been waited for
    last_read_unbarriered = set()   # (operand, buffer) regions read since the last (lgkmcnt(0), barrier) pair
    reads_done_since_barrier = set()
    staged = {}                 # (operand, piece) -> list of k-tiles in stream order
    if has_prev:
        for which in "ab":
            for piece in range(8):
                staged[(which, piece)] = [0, 1]
                landed[(which, 0, piece)], landed[(which, 1, piece)] = 0, 1
    mf = 0
    seen = {}                   # (tile, k-step) -> output tiles accumulated
    zeroed = set()              # accumulators whose first MFMA (C = 0) has issued
    cur = {}                    # buffer -> k-tile currently landed in it, per operand
    for e in ev:
        if e[0] == "dma":
            _, which, piece, kt, m0, live, tab = e
            region = 0 if which == "a" else 0x10000
            rel = m0 - region
            buf, idx = rel // 0x8000, (rel % 0x8000) // 0x1000
            assert 0 <= rel < 0x10000 and rel % 0x1000 == 0 and idx == piece, e
            j = len(staged.setdefault((which, piece), []))
            if j < nkt:
                assert kt == (j + rot) % nkt and live and tab == "tab", ("the tile's own k-tile j from its own rows", e, j)
            elif has_next:
                assert kt == j - nkt and live and tab == "tabn", ("trailing iterations stage the NEXT tile's k-tiles 0 / 1 from its rows", e, j)
            else:
                assert not live, ("without a next tile the trailing iterations' loads are out of range (no traffic)", e, j)
            assert buf == j & 1, ("stream tile j must land in buffer j & 1", e, j)
            staged[(which, piece)].append(kt)
            assert (which, buf) not in last_read_unbarriered, ("re-staging a region with reads not yet behind a barrier", e)
            pending.append((which, buf, piece, j))
        elif e[0] == "vmcnt":
            n = e[1]
            assert not has_prev or mf >= 128, "a chained entry waits for no load before its second iteration (it would wait for the previous epilogue's stores)"
            done, pending = pending[:max(0, len(pending) - n)], pending[max(0, len(pending) - n):]
            waited += done
        elif e[0] == "barrier":
            for (which, buf, piece, j) in waited:
                landed[(which, buf, piece)] = j
            waited = []
            # reads that were complete (lgkmcnt(0)) before this barrier no longer pin their regions
            last_read_unbarriered -= reads_done_since_barrier
            reads_done_since_barrier = set()
        elif e[0] == "read":
            _, which, ks, buf, tile, reg = e
            js = {landed.get((which, buf, p)) for p in range(8)}
            assert len(js) == 1 and None not in js, ("reading a buffer whose pieces have not all landed behind a barrier", e, js)
            reads_out.append((which, buf, reg, js.pop(), ks, tile))
            last_read_unbarriered.add((which, buf))
        elif e[0] == "lgkmcnt":
            n = e[1]
            done, reads_out = reads_out[:max(0, len(reads_out) - n)], reads_out[max(0, len(reads_out) - n):]
            for (which, buf, reg, j, ks, blk) in done:
                frag[reg] = (which, j, ks, blk)
            if n == 0:
                reads_done_since_barrier |= set(last_read_unbarriered)
        elif e[0] == "mfma":
            _, acc, rb, ra, csrc = e
            assert csrc == ("0" if mf < 64 else acc), ("the first MFMA of every accumulator takes C = 0, all later ones accumulate", mf, acc, csrc)
            if mf < 64:
                zeroed.add(acc)
            q = int(re.match(r"a\[(\d+):", acc).group(1))
            mi, ni = (q // 4) // 8, (q // 4) % 8          # accumulator quad -> output tile (row block, column block) of the wave's 8 x 8
            t = mf // 128                       # iteration = stream tile
            if t < nkt:
                ks = (mf % 128) // 64
                for reg, which, blk in ((rb, "b", ni), (ra, "a", mi)):
                    assert reg not in {r for (_, _, r, _, _, _) in reads_out}, ("MFMA reads a fragment whose ds_read may still be in flight", mf, reg)
                    assert frag.get(reg) == (which, t, ks, blk), ("MFMA operand is not (operand, tile, k-step, block)", mf, reg, frag.get(reg), (which, t, ks, blk))
                seen.setdefault((t, ks), set()).add((mi, ni))
            mf += 1
    assert mf == 128 * nkt and len(zeroed) == 64, (mf, nkt, len(zeroed))
    assert all(len(v) == 64 for v in seen.values()) and len(seen) == 2 * nkt, "every k-step accumulates into each of the 64 output tiles exactly once"
    assert not pending, "the statement's final vmcnt(0) covers every load (the next tile enters without a landing wait)"
    if has_next:
        for (which, piece), lst in staged.items():
            assert lst[nkt:] == [0, 1], (which, piece, lst[nkt:])
    for (which, piece), lst in staged.items():
        assert lst[:nkt] == [(j + rot) % nkt for j in range(nkt)], (which, piece, lst[:nkt + 2])
    assert len(staged) == 16
    return True


if __name__ == "__main__":
    lines = load()
    for nkt in (1, 2, 3, 4, 5, 8, 64):
        for rot in sorted({0, 1, (3 * nkt) // 8, nkt - 1} & set(range(nkt))):
            check(run(lines, nkt, rot), nkt, rot)
    for nkt, k0 in ((37, 5), (38, 712), (1, 785)):          # split-K items: a k-slab that starts inside the K range
        check(run(lines, nkt, k0=k0), nkt)
    for nkt in (2, 4, 8, 64):          # a chain of three tiles: cold entry -> chained -> chained, last
        check(run(lines, nkt, flags=2), nkt, has_next=True)
        check(run(lines, nkt, flags=3), nkt, has_prev=True, has_next=True)
        check(run(lines, nkt, flags=1), nkt, has_prev=True)
    print("w4 k-loop protocol ok")
