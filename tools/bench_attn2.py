"""Per-kernel times of the self-attention kernels at one shape, from the library's launch profiler (kernel begin -> end per dispatch).
usage: python tools/bench_attn2.py [B H Hkv T Tq D [reps]]   (Tq < T: the last Tq rows are the queries — the prompt-row cache / pruned backward)
       MTL_LIB_PATH=<variant .so> selects a diagnostic build. Reports us per launch, TFLOP/s in the full-rectangle convention and EXECUTED
       (causal) TFLOP/s. Random operands (GUIDE: zero-filled operands clock higher)."""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from med_ts_llm_amd.hip import ops, _native as N

a = [int(x) for x in sys.argv[1:]]
B, H, Hkv, T, Tq, D = (a + [16, 32, 32, 1664, 1536, 128][len(a):])[:6]
reps = a[6] if len(a) > 6 else 10
g = torch.Generator().manual_seed(0)
W = (H + 2 * Hkv) * D
qkv = torch.randn(B, T, W, generator=g).to(torch.bfloat16).cuda()
q, k, v = qkv[:, T - Tq:, :H * D], qkv[..., H * D:(H + Hkv) * D], qkv[..., (H + Hkv) * D:]
do = torch.randn(B, Tq, H * D, generator=g).to(torch.bfloat16).cuda()
scale = 1 / math.sqrt(D)
lib = N.lib()
if os.environ.get("ATTN_CHUNKED") == "1":      # the non-resident kernels (32-rows-per-wave / chunked) also where K / V would fit the LDS
    ops._TUNE["attn"] = 1
o, lse = ops.attention_fwd(q, k, v, H, Hkv, D, scale, True, causal_off=T - Tq)
ops.attention_bwd(q, k, v, o, lse, do, H, Hkv, D, scale, True, causal_off=T - Tq, kv_row0=T - Tq)
torch.cuda.synchronize()
lib.mtl_prof_enable(1)
for _ in range(reps):
    o, lse = ops.attention_fwd(q, k, v, H, Hkv, D, scale, True, causal_off=T - Tq)
    ops.attention_bwd(q, k, v, o, lse, do, H, Hkv, D, scale, True, causal_off=T - Tq, kv_row0=T - Tq)
torch.cuda.synchronize()
rows = N.prof_rows()
lib.mtl_prof_enable(0)
frac = (Tq * (T - Tq) + Tq * (Tq + 1) / 2.0) / (Tq * T)
print(f"B={B} H={H} Hkv={Hkv} T={T} Tq={Tq} D={D}  lib={os.environ.get('MTL_LIB_PATH', 'default')}  causal fraction {frac:.3f}")
for r in sorted(rows, key=lambda r: r["kernel"]):
    us = r["total_ms"] / r["launches"] * 1e3
    tf = r["work"] / (r["total_ms"] * 1e-3) / 1e12
    print(f"  {r['kernel']:48s} {us:9.1f} us  min {r['min_ms'] * 1e3:9.1f}  {tf:7.1f} TF/s rect  {tf * frac:7.1f} TF/s executed")
