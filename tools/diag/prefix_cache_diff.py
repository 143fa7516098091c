"""where do the cached and the full forward differ? (tools/diag: run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import FakeDataset, hf_cfg, model_config, rel_err
from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict

BF16 = torch.bfloat16
for kind, T, n_tok in (("gpt2", 192, 64), ("llama", 192, 64), ("gpt2", 200, 37)):
    cfg = hf_cfg(kind)
    sd = random_state_dict(cfg, seed=3, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda")
    d = cfg.get("hidden_size", cfg.get("n_embd"))
    B, n_last = 3, T - n_tok
    g = torch.Generator().manual_seed(4)
    h0 = torch.randn(B, T, d, generator=g)
    h0[:, :n_tok] = h0[:1, :n_tok]
    if kind == "gpt2":
        h0 = h0 + sd["wpe.weight"][:T]
    h0 = h0.cuda()
    dout = torch.randn(B, n_last, d, generator=g).to(BF16).cuda()
    out_f, saved_f = bb.run_forward(h0, n_last, n_save=n_last)
    dh_f = bb.run_backward(h0, dout, saved_f, n_last, n_last)
    prefix = bb.prefix_cache(h0[:1, :n_tok], ("t", T, n_tok), T)
    out_c, saved_c = bb.run_forward(h0, n_last, n_save=n_last, prefix=prefix)
    dh_c = bb.run_backward(h0, dout, saved_c, n_last, n_last)
    print(kind, T, n_tok, "out", rel_err(out_c.float(), out_f.float()), "dh", rel_err(dh_c[:, n_tok:], dh_f[:, n_tok:]), "prefix", bb.last_n_prefix)
    # per-sample
    for b in range(B):
        print("   sample", b, rel_err(out_c[b].float(), out_f[b].float()))
    # compare saved buffers region by region (same layout)
    sf, sc = saved_f.view(torch.uint8), saved_c.view(torch.uint8)
    n = sf.numel()
    print("   saved bytes", n, "equal frac", float((sf == sc).float().mean()))
