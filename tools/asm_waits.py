"""Static check of the compiled kernels (no GPU): for every kernel that uses LDS-DMA (global_load_lds) list the
`s_waitcnt vmcnt(0)` the compiler placed directly in front of a ds_read group -- each one drains the DMA pipeline at
that point ("what de-pipelines an LDS-DMA loop").
usage: python tools/asm_waits.py med-ts-llm_amd/csrc/mtl_gemm.hip [...]"""
import re, subprocess, sys, tempfile, os

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.split("\n")

for src in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", s], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(s).read().split("\n")
    fn, rows, cur = None, [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = {"name": m.group(1), "glds": 0, "drain": 0, "reads": 0, "vm0": 0}
            rows.append(cur)
            continue
        if cur is None:
            continue
        if "global_load_lds" in l or "buffer_load" in l and " lds" in l:
            cur["glds"] += 1
        if "s_waitcnt vmcnt(0)" in l:
            cur["vm0"] += 1
        if re.search(r"\bds_read", l) and not any(re.search(r"\bds_read", x) for x in lines[max(0, i - 6):i]):
            cur["reads"] += 1
            ctx = [x for x in lines[max(0, i - 8):i] if "s_waitcnt" in x]
            if ctx and "vmcnt(0)" in ctx[-1]:
                cur["drain"] += 1
    names = demangle([r["name"] for r in rows])
    print(f"== {src}")
    for r, n in zip(rows, names):
        if r["glds"]:
            print(f"  drains-before-ds_read={r['drain']:2d}  ds_read groups={r['reads']:3d}  vmcnt(0)={r['vm0']:3d}  glds={r['glds']:3d}  {n[:150]}")
