import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from helpers import *
g = torch.Generator().manual_seed(0)
for R, K, N in [(4096, 3, 1), (32, 9, 3), (4096, 256, 64)]:
    x = torch.randn(R, K, generator=g); w = torch.randn(N, K, generator=g); b = torch.randn(N, generator=g); dy = torch.randn(R, N, generator=g)
    xc = x.clone().requires_grad_(True); wc = w.clone().requires_grad_(True)
    F.linear(xc, wc, b).backward(dy)
    xg = x.cuda().requires_grad_(True); wg = w.cuda().requires_grad_(True)
    yg = F.linear(xg, wg, b.cuda()); yg.backward(dy.cuda())
    print(f"ATen fp32 linear R={R} K={K} N={N}: fwd {rel_err(yg, F.linear(x, w, b)):.2e} dx {rel_err(xg.grad, xc.grad):.2e} dW {rel_err(wg.grad, wc.grad):.2e}")
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "fp32 precision", torch.get_float32_matmul_precision())

# ---- llama/concat semseg forward, stage by stage
from med_ts_llm_amd.models import model_lookup
from med_ts_llm_amd.models.backbone import random_state_dict
from med_ts_llm_amd.utils import dict_to_object
from med_ts_llm_amd.hip.ops import *
from oracle import medtsllm_oracle as O
kind, task, B, L, C, pred, cov, down = "llama", "semantic_segmentation", 2, 100, 3, 100, "concat", "linear"
cfg = hf_cfg(kind); sd = random_state_dict(cfg, seed=7, std=0.06)
off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
torch.manual_seed(11)
model = model_lookup["medtsllm"](dict_to_object(model_config(task, L, pred, cov, down, off)), FakeDataset(C, 4), backbone_state=(cfg, sd))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.requires_grad and p.ndim == 1: p.copy_(0.1 * torch.randn(p.shape))
    model.mapping_layer.weight.mul_(3.0)
model = model.to("cuda"); model.train()
for ntok in (0, 40, 150, 236):
    ids = torch.randint(0, 384, (1, ntok), generator=torch.Generator().manual_seed(2), dtype=torch.int32) if ntok else None
    model.fixed_prompt_ids = ids
    if ids is None:
        model.model_config.prompting.dataset = False
    x = torch.randn(B, L, C, generator=torch.Generator().manual_seed(13)) * torch.tensor([1.0, 2.5, 0.3]) + torch.tensor([0.5, -1.0, 3.0])
    p = {n: t.detach().cpu().float().clone() for n, t in model.named_parameters() if n != "word_embeddings"}
    m = oracle_mcfg({"task": task, "pred_len": pred, "patch_len": 16, "stride": 8, "n_heads": 2, "d_ff": 64, "covariate_mode": cov,
                     "embedding_downsample_mode": down, "n_classes": 4, "C": C})
    tok = [[ids[0].tolist()]] * B if ids is not None else None
    with torch.no_grad():
        ref, inter = O.medtsllm_forward(x, p, sd, cfg, m, token_ids=tok, pad_token_id=0, training=True, return_intermediates=True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            r16 = O.medtsllm_forward(x, p, sd, cfg, m, token_ids=tok, pad_token_id=0, training=True)
        out = model({"x_enc": x.cuda()})
        bb = model.backbone
        x_tok, mean, stdev = model.encode_ts(x.cuda())
        print(f"ntok={ntok}: pred hip {rel_err(out, ref):.3e} mixed {rel_err(r16.float(), ref):.3e}; x_tok err {rel_err(x_tok.float(), inter['reprog_tokens']):.3e}", end="")
        h0 = AssembleFn.apply(x_tok, None if ids is None else ids.cuda(), bb.embed_f32, bb.wpe)
        print(f"  h0 err {rel_err(h0, inter['llm_inputs_embeds']):.3e}", end="")
        dec = BackboneFn.apply(h0, bb, model.n_patches)
        dref = O.backbone_forward(inter['llm_inputs_embeds'], sd, cfg)[:, -model.n_patches:]
        print(f"  dec err {rel_err(dec.float(), dref):.3e}")
