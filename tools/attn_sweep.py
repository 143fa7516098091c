"""backbone self-attention kernels over shapes (per-kernel durations from the launch profiler): which term of the cost model
(fixed latency / linear in T / quadratic in T / grid size) dominates at the metric shape (B=32, H=12, T=256, D=64)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from med_ts_llm_amd.hip import ops, _native as N

lib = N.lib()


def run(B, T, H, D, drop, n=30):
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, T, 3 * H * D, generator=g).to(torch.bfloat16).cuda()
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    do = torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).cuda()
    scale = 1 / math.sqrt(D)
    o, lse = ops.attention_fwd(q, k, v, H, H, D, scale, True, dropout=drop)
    ops.attention_bwd(q, k, v, o, lse, do, H, H, D, scale, True, dropout=drop)
    torch.cuda.synchronize()
    lib.mtl_prof_enable(1)
    for _ in range(n):
        o, lse = ops.attention_fwd(q, k, v, H, H, D, scale, True, dropout=drop)
        ops.attention_bwd(q, k, v, o, lse, do, H, H, D, scale, True, dropout=drop)
    torch.cuda.synchronize()
    rows = N.prof_rows()
    lib.mtl_prof_enable(0)
    return {r["kernel"].split("<")[0].replace("attn_", ""): r["total_ms"] / r["launches"] * 1e3 for r in rows}


SHAPES = [(32, 256, 12, 64), (32, 256, 32, 128)] if len(sys.argv) > 1 else [(32, 256, 12, 64), (32, 128, 12, 64), (32, 64, 12, 64), (16, 256, 12, 64), (8, 256, 12, 64), (64, 256, 12, 64), (32, 256, 32, 128), (32, 256, 6, 64)]
for B, T, H, D in SHAPES:
    for drop in ((0.0, 0), (0.1, 7)):
        r = run(B, T, H, D, drop)
        print(f"B={B:3d} T={T:4d} H={H:3d} D={D:4d} drop={drop[0]}: " + "  ".join(f"{k} {v:7.2f} us" for k, v in r.items()), flush=True)
