"""Evidence for the loose parity bar on cancellation-prone gradients (key/query bias, 1xC feature weighting, mapping bias): on the GPU box,
compares plain torch CUDA fp32/bf16 linears against CPU fp32 on the same shapes and shows that the deviation is the random projection of a
bf16-rounded incoming gradient, not an error of the HIP kernels (referenced from DESIGN.md section 3 and tests/test_gpu_model.py)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from helpers import *
from med_ts_llm_amd.models import model_lookup
from med_ts_llm_amd.models.backbone import random_state_dict
from med_ts_llm_amd.utils import dict_to_object
from med_ts_llm_amd.hip.ops import *
from oracle import medtsllm_oracle as O
kind, task, B, L, C, pred, cov, down = "llama", "forecasting", 2, 64, 3, 16, "weighted-average", "linear"
cfg = hf_cfg(kind); sd = random_state_dict(cfg, seed=7, std=0.06)
off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
torch.manual_seed(11)
model = model_lookup["medtsllm"](dict_to_object(model_config(task, L, pred, cov, down, off)), FakeDataset(C, 0), backbone_state=(cfg, sd))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.requires_grad and p.ndim == 1: p.copy_(0.1 * torch.randn(p.shape))
    model.mapping_layer.weight.mul_(3.0)
model = model.to("cuda"); model.train()
g = torch.Generator().manual_seed(13)
x = torch.randn(B, L, C, generator=g) * torch.tensor([1.0, 2.5, 0.3]) + torch.tensor([0.5, -1.0, 3.0])
tgt = torch.randn(B, pred, C, generator=g)
p = {n: t.detach().cpu().float().clone().requires_grad_(True) for n, t in model.named_parameters() if n != "word_embeddings"}
m = oracle_mcfg({"task": task, "pred_len": pred, "patch_len": 16, "stride": 8, "n_heads": 2, "d_ff": 64, "covariate_mode": cov,
                 "embedding_downsample_mode": down, "n_classes": 0, "C": C})
ref, inter = O.medtsllm_forward(x, p, sd, cfg, m, token_ids=None, training=True, return_intermediates=True)
inter["llm_inputs_embeds"].retain_grad()
F.mse_loss(ref, tgt).backward()
dh0_ref = inter["llm_inputs_embeds"].grad

# HIP: capture dh0 via hook on h0
import med_ts_llm_amd.hip.ops as ops
cap = {}
orig = ops.AssembleFn.backward
def patched(ctx, dh0):
    cap["dh0"] = dh0.detach().cpu().clone()
    return orig(ctx, dh0)
ops.AssembleFn.backward = staticmethod(patched)
out = model({"x_enc": x.cuda()})
F.mse_loss(out, tgt.cuda()).backward()
dh0 = cap["dh0"]
print("pred err", rel_err(out, ref), " dh0 err", rel_err(dh0, dh0_ref))
print("sum(dh0) hip", float(dh0.sum()), "ref", float(dh0_ref.sum()), " |dh0|1", float(dh0_ref.abs().sum()))
print("per-row sums hip", dh0.sum(-1)[0, :4].tolist(), "ref", dh0_ref.sum(-1)[0, :4].tolist())
print("fw.weight grad hip", model.feature_weighting.weight.grad.cpu().tolist(), "ref", p["feature_weighting.weight"].grad.tolist())
print("fw.bias grad hip", model.feature_weighting.bias.grad.cpu().tolist(), "ref", p["feature_weighting.bias"].grad.tolist())
print("T =", dh0.shape)
