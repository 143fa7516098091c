#!/usr/bin/env python3
"""usage: tools/kernel_resources.py <file.hip> [regex]  -> registers / scratch / occupancy / LDS per kernel, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles gfx950 without a GPU)"""
import re
import subprocess
import sys

src = sys.argv[1]
filt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/_res.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/tmp").stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"^void ", "", d).split("(")[0]
    if filt.search(d):
        r = rows[n]
        print(f"{d:62s} VGPR {r.get('VGPRs', '?'):>4s} AGPR {r.get('AGPRs', '?'):>4s} SGPR {r.get('TotalSGPRs', '?'):>4s} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} "
              f"occupancy {r.get('Occupancy [waves/SIMD]', '?'):>2s} LDS {r.get('LDS Size [bytes/block]', '?')}")
