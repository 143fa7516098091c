"""Randomised cross-check of every persistent-GEMM tile configuration x epilogue on ragged shapes (GPU box).
Forces each configuration through mtl_gemm_args.tune_* (ops.gemm_tune) and compares with a float64 reference of the same bf16-rounded operands."""
import sys, os, itertools, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from med_ts_llm_amd.hip import ops, _native as N
lib = N.lib()
BF16, F32 = torch.bfloat16, torch.float32
cfgs = [(128, 64, 2, 4), (128, 64, 3, 4), (128, 96, 2, 4), (128, 96, 3, 4), (128, 128, 2, 8), (128, 128, 3, 8), (128, 128, 2, 4),
        (128, 192, 2, 8), (256, 128, 3, 16), (256, 128, 2, 16), (256, 192, 2, 8)]
random.seed(0)
g = torch.Generator().manual_seed(0)
gelu = lambda v: 0.5 * v * (1 + torch.tanh(0.7978845608028654 * (v + 0.044715 * v ** 3)))
bad = 0
for cfg in cfgs:
    for epi in (N.EPI_STORE, N.EPI_GELU, N.EPI_RESID, N.EPI_DGELU, N.EPI_ACCUM):
        for _ in range(3):
            M, Nn, K = random.choice([1, 37, 128, 300, 515]), random.choice([4, 64, 100, 192, 260, 388]), 64 * random.choice([1, 2, 5])
            A = torch.randn(M, K, generator=g).to(BF16).cuda()
            B = (torch.randn(Nn, K, generator=g) * 0.2).to(BF16).cuda()
            bias = torch.randn(Nn, generator=g).cuda() if epi != N.EPI_ACCUM else None
            lin = A.double().cpu() @ B.double().cpu().t() + (bias.double().cpu() if bias is not None else 0)
            kw, tol = {}, 4e-3
            if epi == N.EPI_STORE:
                ref = lin
            elif epi == N.EPI_GELU:
                kw = dict(aux_out=torch.empty(M, Nn, dtype=BF16, device="cuda"))
                ref = gelu(lin.float().to(BF16).double())
            elif epi == N.EPI_RESID:
                res = torch.randn(M, Nn, generator=g).cuda()
                kw = dict(aux_in=res, out_dtype=F32)
                ref = res.double().cpu() + lin.float().to(BF16).double()
            elif epi == N.EPI_DGELU:
                pre = torch.randn(M, Nn, generator=g).to(BF16).cuda()
                kw = dict(aux_in=pre)
                x = pre.double().cpu().requires_grad_(True)
                gelu(x).sum().backward()
                ref = lin * x.grad
            else:
                c0 = torch.randn(M, Nn, generator=g).cuda()
                kw = dict(out=c0.clone())
                ref = c0.double().cpu() + lin
                tol = 1e-5
            with ops.gemm_tune(*cfg):
                out = ops.gemm_nt(A, B, bias=bias, epilogue=epi, **kw)
            err = float((out.double().cpu() - ref).norm() / (ref.norm() + 1e-30))
            if not err < tol or (epi == N.EPI_GELU and float((kw["aux_out"].double().cpu() - lin).norm() / lin.norm()) > 4e-3):
                bad += 1
                print("MISMATCH", cfg, epi, (M, Nn, K), err)
print("gemm stress:", "FAILED %d" % bad if bad else "all configurations x epilogues agree")
