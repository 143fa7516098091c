import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import *
from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
from oracle import medtsllm_oracle as O
BF16 = torch.bfloat16
for kind, B, T, n_last in [("gpt2", 2, 100, 37), ("gpt2", 2, 248, 12), ("llama", 2, 100, 37), ("llama", 2, 248, 12), ("llama", 2, 192, 12), ("llama", 2, 200, 200),
                           ("llama_gqa", 2, 130, 20), ("llama_gqa", 2, 248, 12)]:
    cfg = hf_cfg(kind)
    sd = random_state_dict(cfg, seed=3, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda")
    d = bb.cfg["d"]
    g = torch.Generator().manual_seed(5)
    h0 = torch.randn(B, T, d, generator=g)
    dout = torch.randn(B, n_last, d, generator=g).to(BF16)
    res = {}
    for ac in (False, True):
        h0r = h0.clone().requires_grad_(True)
        with torch.autocast("cpu", dtype=BF16, enabled=ac):
            ref = O.backbone_forward(h0r, sd, cfg)[:, -n_last:, :]
            (ref.float() * dout.float()).sum().backward()
        res[ac] = (ref.detach().float(), h0r.grad.clone())
    h_in = (h0 + sd["wpe.weight"][:T] if kind == "gpt2" else h0).cuda()
    out, saved = bb.run_forward(h_in, n_last)
    dh0 = bb.run_backward(h_in, dout.cuda(), saved, n_last)
    print(f"{kind:10s} T={T:4d} n_last={n_last:4d} fwd hip {rel_err(out.float(), res[False][0]):.3e} mixed {rel_err(res[True][0], res[False][0]):.3e} | "
          f"bwd hip {rel_err(dh0, res[False][1]):.3e} mixed {rel_err(res[True][1], res[False][1]):.3e}")
    # per-token error profile of the forward
    e = ((out.float().cpu() - res[False][0]).norm(dim=-1) / res[False][0].norm(dim=-1))
    print("    per-row fwd err (sample 0):", " ".join(f"{v:.1e}" for v in e[0][:: max(1, n_last // 12)].tolist()))
    eb = ((dh0.cpu() - res[False][1]).norm(dim=-1) / (res[False][1].norm(dim=-1) + 1e-9))
    print("    per-row bwd err (sample 0):", " ".join(f"{v:.1e}" for v in eb[0][:: max(1, T // 16)].tolist()))
