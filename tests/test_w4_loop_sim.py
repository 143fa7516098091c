"""CPU test of the generated k-loop of gemm_nt_w4_kernel (csrc/mtl_gemm_w4_loop.inc): the asm text is run through tools/sim_gemm_w4_loop.py, which
executes its scalar bookkeeping and checks the staging / reading protocol the kernel relies on (every k-tile staged once into the right buffer, reads
only behind a landing wait + barrier, no re-staging of a region with unfinished reads, counted waits, every MFMA operand = the fragment it must be).
The GPU tests (test_gpu_kernels.py::test_gemm_w4_*) check the arithmetic; this one catches a schedule edit that only races."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sim_gemm_w4_loop as sim      # noqa: E402


def test_committed_inc_is_what_the_generator_writes(tmp_path):
    before = open(sim.INC).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_w4_loop.py")], check=True, capture_output=True)
    assert open(sim.INC).read() == before, "med-ts-llm_amd/csrc/mtl_gemm_w4_loop.inc is stale: run tools/gen_gemm_w4_loop.py"


@pytest.mark.parametrize("nkt", [1, 2, 3, 4, 5, 8, 64, 344])
def test_k_loop_protocol(nkt):
    lines = sim.load()
    for rot in sorted({0, 1, (3 * nkt) // 8, nkt - 1} & set(range(nkt))):
        assert sim.check(sim.run(lines, nkt, rot), nkt, rot)


def test_checker_catches_a_short_landing_wait():
    lines = sim.load()
    waits = [l for l in lines if l.startswith("s_waitcnt vmcnt(") and l != "s_waitcnt vmcnt(16)" and "lgkmcnt" not in l]
    assert waits, "no in-loop landing wait found"
    n = int(waits[0].split("(")[1].split(")")[0])
    bad = [l.replace(f"vmcnt({n})", f"vmcnt({n + 2})") if l == waits[0] else l for l in lines]
    with pytest.raises(AssertionError):
        sim.check(sim.run(bad, 6, 0), 6, 0)


@pytest.mark.parametrize("nkt", [2, 4, 8, 64, 172])
def test_tile_chaining_protocol(nkt):
    """a persistent workgroup's chain of tiles: cold entry with a next tile -> chained entry with a next tile -> chained entry, last tile. The trailing
    iterations stage the next tile's k-tiles 0 / 1 from ITS row table into buffers 0 / 1, the final vmcnt(0) covers them, and a chained entry
    issues neither a load nor a landing wait before its second iteration"""
    lines = sim.load()
    assert sim.check(sim.run(lines, nkt, flags=2), nkt, has_next=True)
    assert sim.check(sim.run(lines, nkt, flags=3), nkt, has_prev=True, has_next=True)
    assert sim.check(sim.run(lines, nkt, flags=1), nkt, has_prev=True)


def test_checker_catches_a_chain_into_the_wrong_row_table():
    lines = sim.load()
    bad = [l.replace("%[tabn]", "%[tab]") for l in lines]
    with pytest.raises(AssertionError):
        sim.check(sim.run(bad, 8, flags=2), 8, has_next=True)


def test_checker_catches_a_wrong_accumulator_or_fragment():
    """the 16 x 16 x 32 kernel names its 64 accumulator quads and 32 fragments physically: an MFMA that accumulates tile (mi, ni) from another block's fragment,
    or two MFMAs of a k-step that hit the same accumulator, must not pass"""
    lines = sim.load()
    idx = [i for i, l in enumerate(lines) if l.startswith("v_mfma_f32_16x16x32_bf16")]
    a, b = lines[idx[3]].split(", "), lines[idx[4]].split(", ")      # MFMAs 3 / 4 of the k-step: column blocks 0 / 1 (snake order), same row block
    assert a[1] != b[1] and a[0] != b[0]
    swapped = list(lines)
    swapped[idx[3]] = ", ".join([a[0], b[1], a[2], a[3]])             # the neighbour's B fragment under this tile's accumulator
    with pytest.raises(AssertionError):
        sim.check(sim.run(swapped, 4), 4)
    doubled = list(lines)
    doubled[idx[4]] = ", ".join([a[0]] + b[1:3] + [a[3]])      # MFMA 4 writes MFMA 3's accumulator quad a second time
    with pytest.raises(AssertionError):
        sim.check(sim.run(doubled, 4), 4)


@pytest.mark.parametrize("nkt,k0", [(37, 5), (38, 712), (1, 785), (10, 65)])
def test_k_slab_items(nkt, k0):
    """split-K work items: the k-loop of a slab that starts k0 k-tiles into the K range (count and offset arrive in the packed flags word)"""
    assert sim.check(sim.run(sim.load(), nkt, k0=k0), nkt)
