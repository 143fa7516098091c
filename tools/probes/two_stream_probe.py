"""Probe (GPU box): frozen GPT-2-small stack forward + backward on B = 32 sequences of T = 256 — one stream vs the batch split in
two halves on two streams (does one half's epilogue / prologue burst hide under the other half's main loops?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict

CFG = {"model_type": "gpt2", "vocab_size": 50257, "n_positions": 1024, "n_embd": 768, "n_layer": 12, "n_head": 12,
       "layer_norm_epsilon": 1e-5, "embd_pdrop": 0.1, "attn_pdrop": 0.1, "resid_pdrop": 0.1}
sd = random_state_dict(CFG, seed=0, std=0.02, device="cuda", dtype=torch.bfloat16)
bb = FrozenBackbone(CFG, sd, "cuda")
B, T, d, n_last, n_grad = 32, 256, 768, 128, 128
h0 = torch.randn(B, T, d, device="cuda")
dout = torch.randn(B, n_last, d, device="cuda").to(torch.bfloat16)
drop = (0.1, 0.1, 1234)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    out, saved = bb.run_forward(h0, n_last, drop=drop, n_save=n_grad)
    return bb.run_backward(h0, dout, saved, n_last, n_grad, drop=drop)


def two():
    cur = torch.cuda.current_stream()
    res = []
    for s, sl in ((s1, slice(0, B // 2)), (s2, slice(B // 2, B))):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            out, saved = bb.run_forward(h0[sl], n_last, drop=drop, n_save=n_grad)
            res.append((s, sl, saved))
    outs = []
    for s, sl, saved in res:
        with torch.cuda.stream(s):
            outs.append(bb.run_backward(h0[sl], dout[sl], saved, n_last, n_grad, drop=drop))
    for s in (s1, s2):
        cur.wait_stream(s)
    return outs


for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per fwd+bwd of the stack")
