"""GPU probe (tools/, not product): the mapping layer's forward GEMM source[1024, 768] = Wmap[1024, 51200] @ WembT[768, 51200]^T (split-K): the persistent 8-wave
split path (S = 16, a divisor of the 800 k-steps) against the 4-wave kernel's (tile, k-slab) items with S = 21 uneven slabs; cold operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops                   # noqa: E402

BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
S_, d, Vp = 1024, 768, 51200
wm = (torch.randn(S_, Vp, generator=g) * 0.02).to(BF16).cuda()
wT = (torch.randn(d, Vp, generator=g) * 0.05).to(BF16).cuda()
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ref = wm.float() @ wT.float().t()
for label, S, tune in (("8-wave split, S = 16", 16, dict(bm=128, bn=192, stages=2, waves=8)), ("4-wave items, S = 21", 21, dict(bm=256, bn=256, stages=2, waves=4)), ("8-wave split, S = 16", 16, dict(bm=128, bn=192, stages=2, waves=8)),
                       ("4-wave items, S = 21", 21, dict(bm=256, bn=256, stages=2, waves=4)), ("4-wave items, S = 16", 16, dict(bm=256, bn=256, stages=2, waves=4)), ("4-wave items, S = 42", 42, dict(bm=256, bn=256, stages=2, waves=4))):
    def go():
        if tune:
            with ops.gemm_tune(**tune):
                return ops.gemm_nt(wm, wT, split_k=S)
        return ops.gemm_nt(wm, wT, split_k=S)
    out = go()
    err = float((out.float() - ref).norm() / ref.norm())
    ts = []
    for _ in range(10):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{label:24s}: {ts[len(ts) // 2]:7.1f} us (GEMM + reduce)   rel err {err:.2e}", flush=True)
