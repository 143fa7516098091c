"""Stage-by-stage error of the HIP path vs the fp32 oracle and vs the oracle under CPU bf16 autocast (GPU box only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from helpers import *
from med_ts_llm_amd.models import model_lookup
from med_ts_llm_amd.models.backbone import random_state_dict
from med_ts_llm_amd.utils import dict_to_object
from med_ts_llm_amd.hip import ops
from med_ts_llm_amd.hip.ops import *
from oracle import medtsllm_oracle as O

kind, task, B, L, C, pred, cov, down = "gpt2", "segmentation", 2, 64, 1, 64, "univariate", "linear"
cfg = hf_cfg(kind); sd = random_state_dict(cfg, seed=7, std=0.06)
off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
torch.manual_seed(11)
model = model_lookup["medtsllm"](dict_to_object(model_config(task, L, pred, cov, down, off)), FakeDataset(C, 0), backbone_state=(cfg, sd))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.requires_grad and p.ndim == 1: p.copy_(0.1 * torch.randn(p.shape))
    model.mapping_layer.weight.mul_(3.0)
model = model.to("cuda"); model.train()
ids = torch.randint(0, 384, (1, 40), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
model.fixed_prompt_ids = ids
g = torch.Generator().manual_seed(13)
x = torch.randn(B, L, C, generator=g) + 0.5
p = {n: t.detach().cpu().float().clone() for n, t in model.named_parameters() if n != "word_embeddings"}
m = oracle_mcfg({"task": task, "pred_len": pred, "patch_len": 16, "stride": 8, "n_heads": 2, "d_ff": 64, "covariate_mode": cov,
                 "embedding_downsample_mode": down, "n_classes": 0, "C": C})
tok = [[ids[0].tolist()]] * B

def stages(autocast):
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else torch.autocast("cpu", enabled=False)
    out = {}
    with ctx, torch.no_grad():
        mean, stdev = O.revin_stats(x)
        xn = O.revin_norm(x, mean, stdev)
        out["tokens"] = O.patch_embed(xn, p["patch_embedding.value_embedding.tokenConv.weight"], 16, 8)
        we = O.word_embeddings_of(sd, cfg)
        out["source"] = O.source_embeddings(we, p["mapping_layer.weight"], p["mapping_layer.bias"])
        out["x_tok"] = O.reprogramming(out["tokens"], out["source"], p, 2)
        prompt = O.prompt_embeddings(tok, sd["wte.weight"], 0)
        enc = torch.cat([prompt, out["x_tok"].float()], dim=1)
        out["h0"] = enc + sd["wpe.weight"][: enc.shape[1]]
        dec = O.backbone_forward(enc, sd, cfg)
        out["dec"] = dec[:, -out["x_tok"].shape[1]:, :]
        out["down"] = F.linear(out["dec"], p["embedding_downsample_layer.weight"], p["embedding_downsample_layer.bias"])
        hi = out["down"].permute(0, 2, 1).reshape(B, -1)
        out["head"] = F.linear(hi, p["output_projection.linear.weight"], p["output_projection.linear.bias"])
    return {k: v.float() for k, v in out.items()}

r32, r16 = stages(False), stages(True)
hip = {}
with torch.no_grad():
    bb = model._ensure_backbone(torch.device("cuda"))
    tokens, mean, stdev = PatchTokenizeFn.apply(x.cuda(), model.patch_embedding.value_embedding.tokenConv.weight, 16, 8, False)
    hip["tokens"] = tokens[..., :8].float().cpu()
    source = MappingFn.apply(model.mapping_layer.weight, model.mapping_layer.bias, model._wT, model._w, model._map_split_k)
    hip["source"] = source.float().cpu()
    rl = model.reprogramming_layer
    q = LinearFn.apply(tokens, rl.query_projection.weight, rl.query_projection.bias)
    k = LinearFn.apply(source, rl.key_projection.weight, rl.key_projection.bias)
    v = LinearFn.apply(source, rl.value_projection.weight, rl.value_projection.bias)
    a = CrossAttnFn.apply(q, k, v, 2, 64)
    enc = LinearFn.apply(a, rl.out_projection.weight, rl.out_projection.bias)
    hip["x_tok"] = enc.float().cpu()
    h0 = AssembleFn.apply(enc, ids.cuda(), bb.embed_f32, bb.wpe)
    hip["h0"] = h0.cpu()
    dec = BackboneFn.apply(h0, bb, model.n_patches)
    hip["dec"] = dec.float().cpu()
    dn = LinearFn.apply(dec, model.embedding_downsample_layer.weight, model.embedding_downsample_layer.bias)
    hip["down"] = dn.float().cpu()
    hi = dn.permute(0, 2, 1).reshape(B, -1).contiguous()
    hip["head"] = LinearFn.apply(hi, model.output_projection.linear.weight, model.output_projection.linear.bias).float().cpu()
    # isolated backbone: feed the fp32-oracle h0 to the HIP stack
    dec_iso, _ = bb.run_forward(r32["h0"].cuda().contiguous(), model.n_patches, keep=False)
    hip["dec(iso: oracle h0)"] = dec_iso.float().cpu()
r32["dec(iso: oracle h0)"] = r32["dec"]; r16["dec(iso: oracle h0)"] = r16["dec"]
for kk in hip:
    print(f"{kk:22s} hip-vs-fp32 {rel_err(hip[kk], r32[kk]):.3e}   mixed-vs-fp32 {rel_err(r16[kk], r32[kk]):.3e}   |ref| {float(r32[kk].norm()):.3e}")
