"""GPU probe (tools/, not product): the workload of tools/probes/gemm_memside_pmc.sh — gemm_nt_w4_kernel and the vendor's assembly GEMM (torch.matmul) on two
Llama shapes, cold operands, 6 launches each, so that rocprofv3 --pmc passes can set the two kernels' memory-side counters next to each other."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops                   # noqa: E402

BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for M, Nn, K in ((4096, 4096, 4096), (4096, 4096, 22016), (4096, 12288, 4096)):
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
    out = torch.empty(M, Nn, dtype=BF16, device="cuda")
    for it in range(6):
        flush.fill_(it)
        with ops.gemm_tune(bm=256, bn=256, stages=2, waves=4):
            ops.gemm_nt(A, B, out=out)
        flush.fill_(it + 8)
        torch.matmul(A, B.t(), out=out)
    torch.cuda.synchronize()
