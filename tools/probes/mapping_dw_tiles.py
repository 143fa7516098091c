"""GPU probe (tools/, not product): the mapping layer's weight-gradient GEMM dW[1024, 50257] = dsrc[1024, 768] @ Wemb[50257, 768]^T (fp32 output, odd row
length: dword stores) — main columns (49152 = 3 whole rounds of 256 x 256 tiles) on the 4-wave kernel against the 8-wave one, warm L2-sized operands, cold output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops                   # noqa: E402

BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
S, V, d, NM = 1024, 50257, 768, 49152
dsrc = torch.randn(S, d, generator=g).to(BF16).cuda()
w = (torch.randn(V, d, generator=g) * 0.05).to(BF16).cuda()
dW = torch.empty(S, V, dtype=torch.float32, device="cuda")
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ref = dsrc.float() @ w[:NM].float().t()
for waves in (8, 4, 8, 4):
    with ops.gemm_tune(bm=256, bn=256, stages=2, waves=waves):
        ops.gemm_nt(dsrc, w[:NM], out=dW[:, :NM])
        err = float((dW[:, :NM] - ref).norm() / ref.norm())
        ts = []
        for _ in range(10):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_nt(dsrc, w[:NM], out=dW[:, :NM]); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"waves {waves}: {ts[len(ts) // 2]:7.1f} us   rel err vs fp32 math {err:.2e}", flush=True)
