"""Run one GEMM shape/variant a few times (target for PMC passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from med_ts_llm_amd.hip import ops, _native as N
M, Nn, K, mode, bm, bn, st, nw = [int(a) for a in sys.argv[1:9]]
lib = N.lib(); ops._TUNE["gemm"] = (1 if mode == 0 else 2, bm, bn, st, nw)      # per-call fields of mtl_gemm_args
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(); B = (torch.randn(Nn, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
out = ops.gemm_nt(A, B)
for _ in range(10):
    ops.gemm_nt(A, B, out=out)
torch.cuda.synchronize()
