"""diagnostic: is any kernel's rounding error BIASED (signed mean error vs mean |value|)? A coherent bias of 1e-4 survives sums over
10^4 elements; random rounding does not."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from med_ts_llm_amd.hip import ops, _native as N
from helpers import hf_cfg
from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
from oracle import medtsllm_oracle as O
BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)

def rep(name, got, ref):
    got, ref = got.double().cpu().flatten(), ref.double().cpu().flatten()
    e = got - ref
    print(f"{name:46s} norm-rel {float(e.norm() / ref.norm()):.2e}   signed sum(e)/sum|ref| {float(e.sum() / ref.abs().sum()):+.2e}   sum(e*sign(ref))/sum|ref| {float((e * ref.sign()).sum() / ref.abs().sum()):+.2e}")

A = torch.randn(512, 256, generator=g).to(BF16); B = (torch.randn(384, 256, generator=g) * 0.1).to(BF16)
ref = A.double() @ B.double().t()
rep("gemm bf16 out", ops.gemm_nt(A.cuda(), B.cuda()), ref)
rep("  (reference rounding of the exact result)", ref.float().to(BF16), ref)
for kind in ("llama", "gpt2"):
    cfg = hf_cfg(kind); sd = random_state_dict(cfg, seed=7, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda"); d = bb.cfg["d"]
    for T, n_last in ((9, 9), (64, 64)):
        Bt = 2
        h0 = torch.randn(Bt, T, d, generator=g) * 1.5 + 0.3
        dout = torch.randn(Bt, n_last, d, generator=g).to(BF16)
        def run(autocast):
            hr = h0.clone().requires_grad_(True)
            with torch.autocast("cpu", dtype=BF16, enabled=autocast):
                r = O.backbone_forward(hr, sd, cfg)[:, -n_last:, :]
            (r.float() * dout.float()).sum().backward()
            return r.detach().float(), hr.grad
        r32, g32 = run(False); r16, g16 = run(True)
        hin = (h0 + sd["wpe.weight"][:T] if kind == "gpt2" else h0).cuda()
        out, saved = bb.run_forward(hin, n_last)
        dh = bb.run_backward(hin, dout.cuda(), saved, n_last)
        rep(f"{kind} T={T} stack fwd  hip", out.float(), r32); rep(f"{kind} T={T} stack fwd  mixed", r16, r32)
        rep(f"{kind} T={T} stack dh0  hip", dh, g32); rep(f"{kind} T={T} stack dh0  mixed", g16, g32)
        print(f"    row sums of dh0 error / row L1: hip {float(((dh.cpu() - g32).sum(-1).abs()).mean() / g32.abs().sum(-1).mean()):.2e}  mixed {float(((g16 - g32).sum(-1).abs()).mean() / g32.abs().sum(-1).mean()):.2e}")
