#!/usr/bin/env python3
"""usage: tools/asm_loops.py <file.hip> <mangled-name substring>  -> per basic block of the kernel that contains MFMAs: instruction mix
(compiles the device code to assembly; no GPU needed)"""
import re
import subprocess
import sys
from collections import Counter

src, pat = sys.argv[1], sys.argv[2]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I",
                "/root/repo/med-ts-llm_amd/csrc", src, "-o", "/tmp/_k.s"], check=True, stderr=subprocess.DEVNULL, cwd="/tmp")
s = open("/tmp/_k.s").read()
starts = [m for m in re.finditer(r"^(\S+):\s*(;.*)?$", s, re.M) if pat in m.group(1) and not m.group(1).startswith(".L")]
for m in starts:
    end = s.index(".end_amdhsa_kernel", m.end()) if ".end_amdhsa_kernel" in s[m.end():] else len(s)
    nxt = s.find("s_endpgm", m.end())
    body = s[m.end():s.find("\n.Lfunc_end", m.end())]
    blocks, cur = {"entry": []}, "entry"
    for line in body.split("\n"):
        t = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            cur = t.split(":")[0]
            blocks[cur] = []
        elif t and not t.startswith((".", ";")):
            blocks[cur].append(t)
    print(m.group(1), sum(len(v) for v in blocks.values()), "instructions")
    for k, v in blocks.items():
        n_mfma = sum("mfma" in x for x in v)
        if n_mfma:
            c = Counter(x.split()[0] for x in v)
            valu = sum(n for op, n in c.items() if op.startswith("v_") and "mfma" not in op)
            print(f"  {k:10s} instr {len(v):4d} mfma {n_mfma:3d} valu {valu:4d} salu {sum(n for op, n in c.items() if op.startswith('s_')):4d} "
                  f"ds {sum(n for op, n in c.items() if op.startswith('ds_')):3d} vmem {sum(n for op, n in c.items() if op.startswith(('global_', 'buffer_', 'flat_'))):3d} "
                  f"exp {c.get('v_exp_f32', 0):3d} waitcnt {c.get('s_waitcnt', 0):3d} nop {c.get('s_nop', 0):3d} | top valu: "
                  + ", ".join(f"{op}:{n}" for op, n in c.most_common(40) if op.startswith("v_") and "mfma" not in op)[:230])
