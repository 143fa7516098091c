"""Prompt-row forward cache (SURVEY.md 7 "legal shortcut i"; mtl_backbone_fwd's prefix_kv): with ONE constant prompt shared by every sample and a
deterministic stack, the per-layer keys / values of the prompt rows are step- and sample-invariant; the forward then runs on the patch rows only.
Cached == uncached up to the summation order of differently tiled GEMMs (the computed rows go through the same kernels on fewer rows), the cache
follows the prompt ids, and anything that makes the prompt rows step-dependent (GPT-2's train-mode dropouts, per-sample prompts, a backward that
wants prompt-row gradients) falls back to the full forward."""
import pytest
import torch

from helpers import FakeDataset, hf_cfg, model_config, rel_err

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
OFF = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}


@pytest.mark.parametrize("kind", ["gpt2", "llama", "llama_gqa", "llama_hd128", "llama_gqa_hd128"])
@pytest.mark.parametrize("T,n_tok", [(192, 64), (200, 37), (1664, 128)])
def test_stack_cached_equals_uncached(kind, T, n_tok):
    from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
    cfg = hf_cfg(kind)
    if kind == "gpt2":
        if T > 256:
            pytest.skip("GPT-2 fixture has 256 positions")
    sd = random_state_dict(cfg, seed=3, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda")
    d = cfg.get("hidden_size", cfg.get("n_embd"))
    B, n_last = 3, T - n_tok
    g = torch.Generator().manual_seed(4)
    h0 = torch.randn(B, T, d, generator=g)
    h0[:, :n_tok] = h0[:1, :n_tok]                        # the shared constant prompt
    if kind == "gpt2":
        h0 = h0 + sd["wpe.weight"][:T]
    h0 = h0.cuda()
    dout = torch.randn(B, n_last, d, generator=g).to(BF16).cuda()
    out_f, saved_f = bb.run_forward(h0, n_last, n_save=n_last)
    dh_f = bb.run_backward(h0, dout, saved_f, n_last, n_last)
    prefix = bb.prefix_cache(h0[:1, :n_tok], ("t", T, n_tok), T)
    assert prefix[1] == n_tok and bb.prefix_cache(h0[:1, :n_tok], ("t", T, n_tok), T)[0] is prefix[0]     # kept while the key stands
    h0c = h0.clone()
    h0c[:, :n_tok] = float("nan")                         # the cached forward must not read the prompt rows of h0
    out_c, saved_c = bb.run_forward(h0c, n_last, n_save=n_last, prefix=prefix)
    assert bb.last_n_prefix == n_tok
    # (hd 128: the cache build runs the n_tok prompt rows through the RESIDENT attention kernels — fewer than 128 queries — while the full forward sends all rows
    #  through the 32-rows-per-wave kernels; each is within 3e-3 of fp32 math (tests/test_gpu_kernels.py), against each other measured 3.1e-3)
    assert rel_err(out_c.float(), out_f.float()) < (4e-3 if "hd128" in kind else 3e-3)
    dh_c = bb.run_backward(h0c, dout, saved_c, n_last, n_last)
    assert torch.all(dh_c[:, :n_tok] == 0)
    assert rel_err(dh_c[:, n_tok:], dh_f[:, n_tok:]) < 5e-3
    # inference: nothing saved, same output
    out_i, _ = bb.run_forward(h0c, n_last, keep=False, prefix=prefix)
    assert torch.equal(out_i, out_c)
    # a changed prompt rebuilds
    builds = bb.prefix_builds
    bb.prefix_cache(h0[:1, :n_tok] * 0.5, ("t", T, n_tok, "other"), T)
    assert bb.prefix_builds == builds + 1


@pytest.mark.parametrize("kind,task,cov", [("llama", "forecasting", "concat"), ("llama_gqa", "semantic_segmentation", "add"),
                                            ("llama", "reconstruction", "independent"), ("gpt2", "forecasting", "concat")])
def test_model_cached_equals_uncached(kind, task, cov):
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    cfg = hf_cfg(kind)
    sd = random_state_dict(cfg, seed=7, std=0.06)
    pred = 16 if task == "forecasting" else 64
    torch.manual_seed(11)
    model = model_lookup["medtsllm"](dict_to_object(model_config(task, 64, pred, cov, "linear", OFF)), FakeDataset(3, 4 if task == "semantic_segmentation" else 0),
                                     backbone_state=(cfg, sd)).to("cuda")
    ids = torch.randint(0, 512, (1, 24), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    model.fixed_prompt_ids = ids
    x = {"x_enc": torch.randn(3, 64, 3, generator=torch.Generator().manual_seed(5)).cuda()}

    def run(cache, train):
        model.prompt_row_cache = cache
        model.train(train)
        model.zero_grad()
        with torch.set_grad_enabled(train):
            out = model(x)
            if train:
                out.float().square().mean().backward()
        return out.detach().float(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, model.backbone.last_n_prefix

    gpt2 = kind == "gpt2"
    o_ref, g_ref, npre = run(False, True)
    assert npre == 0
    o_c, g_c, npre = run(True, True)
    # GPT-2's own dropouts are off in this fixture config (pdrop = 0): its stack is deterministic in train mode too
    assert npre == 24
    assert rel_err(o_c, o_ref) < 3e-3
    params = dict(model.named_parameters())
    for n in g_ref:
        # analytically-zero gradients (the key bias: softmax shift invariance) are pure round-off: compared on an absolute scale
        scale = max(float(g_ref[n].norm()), 1e-3 * float(params[n].detach().norm()) + 1e-6)
        assert float((g_c[n] - g_ref[n]).norm()) / scale < 1.5e-2, n
    e_ref, _, _ = run(False, False)
    e_c, _, npre = run(True, False)
    assert npre == 24 and rel_err(e_c, e_ref) < 3e-3
    builds = model.backbone.prefix_builds
    run(True, True)
    assert model.backbone.prefix_builds == builds          # same ids: no rebuild
    model.fixed_prompt_ids = torch.randint(0, 512, (1, 24), generator=torch.Generator().manual_seed(9), dtype=torch.int32)
    o_new, _, npre = run(True, False)
    assert npre == 24 and model.backbone.prefix_builds == builds + 1
    model.prompt_row_cache = False
    assert rel_err(o_new, model(x).float()) < 3e-3            # the new prompt's cache, not the old one
    # full (unpruned) backward needs prompt-row state: the cache steps aside
    model.prompt_row_cache, model.prune_dead_prompt_grads = True, False
    model.train()
    model(x).float().square().mean().backward()
    assert model.backbone.last_n_prefix == 0
    if gpt2:
        # live train-mode dropouts make the prompt rows step-dependent: no cache in train mode, cache in eval
        cfg2 = dict(cfg, embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1)
        m2 = model_lookup["medtsllm"](dict_to_object(model_config(task, 64, pred, cov, "linear", OFF)), FakeDataset(3), backbone_state=(cfg2, sd)).to("cuda")
        m2.fixed_prompt_ids = ids
        m2.train()
        m2(x)
        assert m2.backbone.last_n_prefix == 0
        m2.eval()
        with torch.no_grad():
            m2(x)
        assert m2.backbone.last_n_prefix == 24


def test_per_sample_prompts_do_not_cache():
    """input statistics make every sample's prompt its own: ids [B, n_tok] -> full forward"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    from helpers import fixture_tokenizer
    cfg = hf_cfg("llama")
    sd = random_state_dict(cfg, seed=7, std=0.06)
    on = dict(OFF, dataset=True, task=True, input_stats=True)
    model = model_lookup["medtsllm"](dict_to_object(model_config("forecasting", 64, 16, "concat", "linear", on)), FakeDataset(3), backbone_state=(cfg, sd)).to("cuda")
    model.tokenizer = fixture_tokenizer()
    model.eval()
    with torch.no_grad():
        model({"x_enc": torch.randn(3, 64, 3, generator=torch.Generator().manual_seed(5)).cuda()})
    assert model.backbone.last_n_prefix == 0
    # dataset + task text only: one shared prompt -> cached, and the key follows the token ids
    shared = dict(OFF, dataset=True, task=True)
    model2 = model_lookup["medtsllm"](dict_to_object(model_config("forecasting", 64, 16, "concat", "linear", shared)), FakeDataset(3), backbone_state=(cfg, sd)).to("cuda")
    model2.tokenizer = fixture_tokenizer()
    model2.eval()
    xb = {"x_enc": torch.randn(3, 64, 3, generator=torch.Generator().manual_seed(5)).cuda()}
    with torch.no_grad():
        a = model2(xb)
        assert model2.backbone.last_n_prefix > 0 and model2.backbone.prefix_builds == 1
        b = model2(xb)
        assert model2.backbone.prefix_builds == 1 and torch.equal(a, b)
        model2.prompt_row_cache = False
        assert rel_err(a, model2(xb)) < 3e-3
