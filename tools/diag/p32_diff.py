import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from med_ts_llm_amd.hip import ops, _native as N
lib = N.lib()
g = torch.Generator().manual_seed(0)
for (M, Nn, K) in ((4096, 4096, 12288), (8192, 4096, 4096), (4096, 4096, 4096), (2048, 4096, 512), (8192, 12288, 4096), (512, 512, 4096), (256, 4096, 256)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    outs = []
    for v in ((1, 256, 256, 2, 8), (1, 256, 256, 5, 8), (1, 256, 256, 5, 8)):
        ops._TUNE["gemm"] = (1 if v[0] == 0 else 2,) + tuple(v[1:])
        outs.append(ops.gemm_nt(A, B, out_dtype=torch.float32).clone())
    ops._TUNE["gemm"] = (0, 0, 0, 0, 0)
    ref = (A.float() @ B.float().t())
    d = (outs[1] - outs[0]).abs()
    bad = (d > 0).nonzero()
    print(M, Nn, K, "max diff", float(d.max()), "n diff", int((d > 0).sum()), "p32 vs fp32 ref", float((outs[1] - ref).abs().max()), "2st vs ref", float((outs[0] - ref).abs().max()),
          "p32 self-consistent", bool(torch.equal(outs[1], outs[2])))
    if len(bad):
        r, c = bad[:, 0], bad[:, 1]
        print("   rows", int(r.min()), int(r.max()), "cols", int(c.min()), int(c.max()), "tiles(m)", sorted(set((r // 256).tolist()))[:10], "tiles(n)", sorted(set((c // 256).tolist()))[:20])
