"""microbenchmark of the backbone self-attention kernels at the metric shape (B=32, H=12, T=256, D=64)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import math, torch
from med_ts_llm_amd.hip import ops
B, T, H, D = 32, 256, 12, 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * H * D, generator=g).to(torch.bfloat16).cuda()
q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
do = torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).cuda()
scale = 1 / math.sqrt(D)
for drop in ((0.0, 0), (0.1, 7)):
    o, lse = ops.attention_fwd(q, k, v, H, H, D, scale, True, dropout=drop)
    def t(fn, n=50):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    print(f"dropout={drop[0]}: fwd {t(lambda: ops.attention_fwd(q, k, v, H, H, D, scale, True, dropout=drop)):.1f} us   "
          f"bwd(dq+dkv) {t(lambda: ops.attention_bwd(q, k, v, o, lse, do, H, H, D, scale, True, dropout=drop)):.1f} us")
