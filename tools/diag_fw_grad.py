"""diagnostic: where does the feature-weighting gradient error of the weighted-average covariate mode come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import hf_cfg, model_config, FakeDataset, oracle_mcfg, rel_err
from med_ts_llm_amd.models import model_lookup
from med_ts_llm_amd.models.backbone import random_state_dict
from med_ts_llm_amd.utils import dict_to_object
from oracle import medtsllm_oracle as O

kind, task, B, L, C, pred, cov, down = "llama", "forecasting", 2, 64, 3, 16, "weighted-average", "linear"
cfg = hf_cfg(kind); sd = random_state_dict(cfg, seed=7, std=0.06)
off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
torch.manual_seed(11)
model = model_lookup["medtsllm"](dict_to_object(model_config(task, L, pred, cov, down, off)), FakeDataset(C, 0), backbone_state=(cfg, sd))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.requires_grad and p.ndim == 1:
            p.copy_(0.1 * torch.randn(p.shape))
    model.mapping_layer.weight.mul_(3.0)
model = model.to("cuda"); model.train()
g = torch.Generator().manual_seed(13)
x = torch.randn(B, L, C, generator=g) * torch.tensor([1.0, 2.5, 0.3][:C]) + torch.tensor([0.5, -1.0, 3.0][:C])
tap = model.debug_tap = {}
pred_hip = model({"x_enc": x.cuda()})
p = {n: t.detach().cpu().float().clone().requires_grad_(t.requires_grad) for n, t in model.named_parameters() if n != "word_embeddings"}
meta = {"task": task, "pred_len": pred, "patch_len": 16, "stride": 8, "n_heads": 2, "d_ff": 64, "covariate_mode": cov, "embedding_downsample_mode": down, "n_classes": 0, "C": C}
m = oracle_mcfg(meta)
tgt = torch.randn(pred_hip.shape, generator=g)

def oracle(autocast):
    pp = {n: t.detach().clone().requires_grad_(t.requires_grad) for n, t in p.items()}
    keep = {}
    orig = O.encode_ts
    def enc_ts(*a, **k):
        r = orig(*a, **k)
        r[0].retain_grad(); keep["x_tok"] = r[0]
        return r
    O.encode_ts = enc_ts
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = O.medtsllm_forward(x, pp, sd, cfg, m, token_ids=None, training=True)
            loss = torch.nn.functional.mse_loss(out, tgt)
        loss.backward()
    finally:
        O.encode_ts = orig
    return out, pp, keep["x_tok"]

o32, p32, xt32 = oracle(False)
o16, p16, xt16 = oracle(True)
torch.nn.functional.mse_loss(pred_hip, tgt.cuda()).backward()
gh = {n: t.grad.detach().cpu().float() for n, t in model.named_parameters() if t.requires_grad}
print("pred: hip", rel_err(pred_hip, o32), "mixed", rel_err(o16.float(), o32))
up_h = tap["grad:fw_out@0"].float().cpu().squeeze(-1)          # [B, P, d]
up32, up16 = xt32.grad.float(), xt16.grad.float()
print("upstream grad of the weighted tokens: hip vs fp32", rel_err(up_h, up32), " mixed vs fp32", rel_err(up16, up32), "dtype", xt16.grad.dtype)
print("  sum over everything: fp32 %.6e  mixed %.6e  hip %.6e   L1 mass %.6e" % (up32.sum(), up16.sum(), up_h.sum(), up32.abs().sum()))
for n in ("feature_weighting.bias", "feature_weighting.weight", "reprogramming_layer.out_projection.weight"):
    print(n, "fp32", p32[n].grad.flatten()[:3].tolist(), "mixed", p16[n].grad.flatten()[:3].tolist(), "hip", gh[n].flatten()[:3].tolist())
fw_in = tap["fw_in@0"].double().cpu().reshape(-1, C)
dy = tap["grad:fw_out@0"].double().cpu().reshape(-1, 1)
print("exact contraction of hip's own upstream:", (dy.t() @ fw_in).flatten().tolist(), " bias", dy.sum().item())
# which part of the upstream error matters: error projected on ones vs norm
e_h, e_m = (up_h - up32), (up16 - up32)
print("upstream error: hip norm %.4e sum %.4e | mixed norm %.4e sum %.4e" % (e_h.norm(), e_h.sum(), e_m.norm(), e_m.sum()))
print("per-token-row sum of upstream (first 6): fp32", up32.sum(-1).flatten()[:6].tolist(), "\n   hip", up_h.sum(-1).flatten()[:6].tolist(), "\n   mixed", up16.sum(-1).flatten()[:6].tolist())
