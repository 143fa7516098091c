"""GPU parity of the assembled path against the oracle: frozen backbone stack (fwd + activation-gradient bwd)
and the full MedTsLLM forward/backward with every trainable gradient.

Ladder: the HIP path computes GEMM/attention operands in bf16 with fp32 accumulation, the oracle is
fp32, so the bar is L3: norm-wise error <= 1.5 x the reference's own bf16-vs-fp32 deviation (7.8e-3) = 1.2e-2.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import (rel_err, hf_cfg, model_config, FakeDataset, fixture_tokenizer, oracle_mcfg, golden_loss, cancellation_checks,
                     MIXED_FACTOR, GRAD_FLOOR, EXACT_SUM, fwd_bar, grad_factor)

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
L3 = 1.2e-2


def _oracle_bcfg(cfg):
    c = dict(cfg)
    return c


@pytest.mark.parametrize("kind,B,T,n_last", [("gpt2", 2, 100, 37), ("gpt2", 3, 64, 64), ("llama", 2, 100, 37), ("llama_gqa", 2, 130, 20)])
def test_backbone_stack(kind, B, T, n_last):
    from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
    from oracle import medtsllm_oracle as O
    cfg = hf_cfg(kind)
    sd = random_state_dict(cfg, seed=3, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda")
    d = bb.cfg["d"]
    g = torch.Generator().manual_seed(5)
    h0 = torch.randn(B, T, d, generator=g)
    dout = torch.randn(B, n_last, d, generator=g).to(BF16)
    h0r = h0.clone().requires_grad_(True)
    ref = O.backbone_forward(h0r, sd, cfg)[:, -n_last:, :]
    (ref * dout.float()).sum().backward()
    h_in = h0 + sd["wpe.weight"][:T] if kind == "gpt2" else h0     # the GPT-2 position add is part of the assembly kernel
    h_in = h_in.cuda()
    out, saved = bb.run_forward(h_in, n_last)
    assert rel_err(out.float(), ref) < L3
    dh0 = bb.run_backward(h_in, dout.cuda(), saved, n_last)
    assert rel_err(dh0, h0r.grad) < 2 * L3
    # pruned backward: only the last n_grad tokens get a gradient (causality makes it exact for those rows), rest stays 0
    for n_grad in sorted({n_last, min(T, n_last + 23), T}):
        dh0p = bb.run_backward(h_in, dout.cuda(), saved, n_last, n_grad).cpu()
        assert rel_err(dh0p[:, T - n_grad:], h0r.grad[:, T - n_grad:]) < 2 * L3, n_grad
        assert rel_err(dh0p[:, T - n_grad:], dh0.cpu()[:, T - n_grad:]) < 2e-3, n_grad     # same arithmetic as the full pass
        assert torch.all(dh0p[:, : T - n_grad] == 0)


CASES = [
    # kind, task, B, L, C, pred, cov, down, prompt on
    ("gpt2", "forecasting", 2, 64, 3, 16, "concat", "linear", True),
    ("gpt2", "reconstruction", 2, 64, 3, 64, "independent", "truncate", True),
    ("gpt2", "anomaly_detection", 2, 64, 2, 64, "interleave", "average", False),
    ("llama", "semantic_segmentation", 2, 100, 3, 100, "concat", "linear", True),
    ("llama_gqa", "forecasting", 3, 72, 2, 24, "add", "linear", True),
    ("llama", "forecasting", 2, 64, 3, 16, "weighted-average", "linear", False),
    ("llama", "forecasting", 2, 64, 3, 16, "merge-end", "linear", True),
    ("gpt2", "segmentation", 2, 64, 1, 64, "univariate", "linear", True),
    # head width 7 * 3 = 21, not a multiple of 8: the Linear backward takes the transposed-copy route instead of the K-major GEMMs
    ("gpt2", "forecasting", 2, 64, 3, 7, "concat", "linear", False),
    # vocabulary > 100 000: the sub-sampled word-embedding table is a trainable parameter (Llama-3 quirk, R:models/medtsllm.py:220-222)
    ("llama_gqa_bigvocab", "reconstruction", 2, 64, 2, 64, "concat", "linear", False),
    # "examples" prompting: a tensor part inside the prompt goes through encode_ts (R:models/medtsllm.py:313-319)
    ("gpt2", "forecasting", 2, 64, 3, 16, "concat", "linear", "examples"),
    ("llama", "semantic_segmentation", 2, 64, 3, 64, "add", "linear", "examples"),
]


@pytest.mark.parametrize("kind,task,B,L,C,pred,cov,down,prompt_on", CASES)
def test_full_model_fwd_bwd(kind, task, B, L, C, pred, cov, down, prompt_on):
    _check_full_model(kind, task, B, L, C, pred, cov, down, prompt_on)


def test_psm_shaped_anomaly_detection_vs_oracle():
    """BASELINE.json configs[3] at a checkable size: anomaly detection (reconstruction) on a Llama MHA backbone, concat covariates with
    C = 25 channels -> a concat width of 25 * 32 = 800 that is NOT a multiple of 64 (padded to 832 for the query GEMM), the front-end
    dimensions of the benchmark (d_model 32, d_ff 128, 8 heads, 1024 prototypes), a text prompt, and a wide flatten head
    (128 * 64 -> 512 * 25 = 12 800 outputs). Forward, every gradient and the eval output against the fp32 oracle."""
    _check_full_model("llama", "anomaly_detection", 2, 512, 25, 512, "concat", "linear", True, d_model=32, d_ff=128, H=8, num_tokens=1024)


def _check_full_model(kind, task, B, L, C, pred, cov, down, prompt_on, d_model=8, d_ff=64, H=2, num_tokens=64, grad_bar=MIXED_FACTOR,
                      hf=None, sd=None, prompting=None, descriptions=None, llm_layers=-1, dataset=None, pure_bf16=False):
    """hf / sd: a full backbone config + CPU fp32 state dict instead of the small helpers.hf_cfg(kind) one (tests/test_gpu_realwidth.py);
    prompting: the config's prompting table as shipped (overrides prompt_on); descriptions: per-sample clip descriptions (`clip` prompts).
    pure_bf16: setup.dtype = "bf16" (R:tasks/base.py:261-262,205-208) — model.to(bf16), bf16 inputs, bf16 residual stream; the oracle gets the SAME
    bf16-rounded parameters and inputs in fp32, and the yardstick is the oracle run with everything cast to bf16 (tests/golden/rw_*_bf16 pin the
    same mode against the reference model itself)."""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    from oracle import medtsllm_oracle as O

    if hf is not None:
        cfg = hf
    else:
        cfg = hf_cfg("llama_gqa", vocab=100_100) if kind == "llama_gqa_bigvocab" else hf_cfg(kind)
    if sd is None:
        sd = random_state_dict(cfg, seed=7, std=0.06)
    ex_on = prompt_on == "examples"
    prompt_on = bool(prompt_on)
    if prompting is None:
        prompting = {"dataset": prompt_on, "task": prompt_on, "clip": False, "input_stats": prompt_on, "examples": ex_on,
                     "input_stats_dim": 0, "input_stats_select": "all"}
    else:
        prompt_on = any(prompting.get(k, False) for k in ("dataset", "task", "clip", "input_stats"))
    n_classes = 4 if task == "semantic_segmentation" else 0
    config = dict_to_object(model_config(task, L, pred, cov, down, prompting, d_model=d_model, d_ff=d_ff, H=H, num_tokens=num_tokens, llm_layers=llm_layers))
    torch.manual_seed(11)
    model = model_lookup["medtsllm"](config, dataset or FakeDataset(C, n_classes), backbone_state=(cfg, sd))
    model.tokenizer = fixture_tokenizer()
    with torch.no_grad():   # make every trainable weight O(0.1) so all branches matter
        for n, p in model.named_parameters():
            if p.requires_grad and p.ndim == 1:
                p.copy_(0.1 * torch.randn(p.shape))
        model.mapping_layer.weight.mul_(3.0)
    model = model.to("cuda")
    if pure_bf16:
        model = model.to(BF16)
        assert all(p_.dtype == BF16 for p_ in model.parameters())
    model.train()
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, L, C, generator=g) * torch.tensor(([1.0, 2.5, 0.3] * 9)[:C]) + torch.tensor(([0.5, -1.0, 3.0] * 9)[:C])
    if pure_bf16:
        x = x.to(BF16).float()            # what prepare_batch hands over (the oracle sees the same rounded window)
    inputs = {"x_enc": x.cuda().to(BF16) if pure_bf16 else x.cuda()}
    if descriptions is not None:
        inputs["descriptions"] = list(descriptions)
    if ex_on:
        ex = torch.randn(B, 40, C, generator=g) * 0.7 + 0.2
        inputs["examples"] = [("Example segment:", ex[i:i + 1].cuda()) for i in range(B)]
    tap = model.debug_tap = {}
    pred_hip = model(inputs)

    # ---- oracle on the same weights / prompt ids
    trainable_emb = model.word_embeddings.requires_grad
    assert trainable_emb == (cfg["vocab_size"] > 100_000) and model.vocab_size == min(cfg["vocab_size"], 100_000)
    p = {n: t.detach().cpu().float().clone().requires_grad_(t.requires_grad) for n, t in model.named_parameters()
         if n != "word_embeddings" or trainable_emb}
    we_kw = {"word_emb": p["word_embeddings"]} if trainable_emb else {}
    tok_ids = None
    if prompt_on:
        # input statistics (median / rFFT lags) are data dependent: take the strings the GPU model itself builds so that
        # both sides see byte-identical prompts (CPU vs GPU FFT round-off can reorder near-tied top-k lags)
        parts = model.build_prompt(inputs)
        tok_ids = [[model.tokenizer(s, padding=False, truncation=False).input_ids if isinstance(s, str) else s.cpu() for s in ps] for ps in parts]
    meta = {"task": task, "pred_len": pred, "patch_len": 16, "stride": 8, "n_heads": H, "d_ff": d_ff, "covariate_mode": cov,
            "embedding_downsample_mode": down, "n_classes": n_classes, "C": C}
    m = oracle_mcfg(meta)
    ref = O.medtsllm_forward(x, p, sd, cfg, m, token_ids=tok_ids, pad_token_id=model.tokenizer.pad_token_id, training=True, **we_kw)
    assert pred_hip.shape == ref.shape
    assert pred_hip.dtype == (BF16 if pure_bf16 else ref.dtype) or not pure_bf16
    # L3 bar = 1.5 x the reference's OWN bf16-vs-fp32 deviation on THIS model: the oracle run under CPU bf16 autocast is
    # the reference's dtype="mixed" arithmetic (same ATen autocast policy: bf16 linear/matmul, fp32 norm/softmax).
    # pure_bf16 (round 6: the HIP path runs its NATIVE bf16 mode there — bf16 residual stream): the yardstick is the reference's dtype = "bf16"
    # arithmetic instead — the same oracle with every parameter, frozen weight and input cast to bf16, no autocast (R:tasks/base.py:261-262)
    p16 = {n: (t.detach().to(BF16) if pure_bf16 else t.detach().clone()).requires_grad_(t.requires_grad) for n, t in p.items()}
    we_kw16 = {"word_emb": p16["word_embeddings"]} if trainable_emb else {}
    if pure_bf16:
        sd16 = {k: v.to(BF16) if v.is_floating_point() else v for k, v in sd.items()}
        ref16 = O.medtsllm_forward(x.to(BF16), p16, sd16, cfg, m, token_ids=tok_ids, pad_token_id=model.tokenizer.pad_token_id, training=True, **we_kw16)
        del sd16
    else:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ref16 = O.medtsllm_forward(x, p16, sd, cfg, m, token_ids=tok_ids, pad_token_id=model.tokenizer.pad_token_id, training=True, **we_kw16)
    self_err = rel_err(ref16.float(), ref)
    bar = fwd_bar(self_err)
    e = rel_err(pred_hip, ref)
    print(f"\n[{kind}/{cov}] pred: hip-vs-fp32 {e:.3e}  reference-mixed-vs-fp32 {self_err:.3e}")
    assert e < bar, (e, self_err)

    if task == "semantic_segmentation":
        tgt = torch.randint(0, n_classes, (B, pred), generator=g)
    elif task == "segmentation":
        tgt = (torch.rand(B, pred, generator=g) > 0.8).float()
    else:
        tgt = torch.randn(ref.shape, generator=g)
    golden_loss(ref, tgt, task).backward()
    if pure_bf16:
        l16 = golden_loss(ref16.float(), tgt, task)
    else:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            l16 = golden_loss(ref16, tgt, task)
    l16.backward()
    loss = golden_loss(pred_hip, tgt.cuda().to(pred_hip.dtype) if (pure_bf16 and tgt.is_floating_point()) else tgt.cuda(), task)
    loss.backward()
    grads = {n: t.grad for n, t in model.named_parameters() if t.requires_grad}
    exact, cond = cancellation_checks(tap, grads, GRAD_FLOOR)
    for n, (e_abs, mass) in exact.items():      # sums with cancellation: exactly the fp64 reduction of the path's own upstream gradient
        # (bf16 parameters carry bf16 gradients: one more rounding of 2^-9 of each element, on the scale of the sum's L1 mass / sqrt(rows))
        assert e_abs <= (4e-3 if pure_bf16 else EXACT_SUM) * mass + 1e-9, ("exact", n, e_abs, mass)
    model.debug_tap = None
    bad = {}
    for n, t in model.named_parameters():
        if not t.requires_grad:
            continue
        assert t.grad is not None, n
        gref = p[n].grad.detach()
        # analytically-zero gradients (key bias: softmax shift invariance) are compared on an absolute scale
        scale = max(float(gref.norm()), 1e-3 * float(p[n].detach().norm()) + 1e-6)
        e_hip = float((t.grad.cpu().float() - gref).norm()) / scale
        e_ref = float((p16[n].grad.detach().float() - gref).norm()) / scale
        print(f"   grad {n:55s} hip {e_hip:.3e}  reference-mixed {e_ref:.3e}")
        # bar: 1.5 x the reference-mixed arithmetic's own error on this tensor (floor 1e-2). Gradients with few elements — bias
        # vectors, the 1 x C feature weighting — are sums with cancellation over the rows of an upstream gradient: a 1e-2-level
        # perturbation of the upstream elements moves them by cond[n] however small the sum itself is (helpers.cancellation_checks),
        # and the ratio of two error samples scatters: 3 x, on the larger of the two scales. tests/test_gpu_golden.py bounds the
        # scatter the other way with an aggregate criterion over all gradients of a case.
        small = t.numel() < 4096
        if e_hip > grad_factor(t.numel(), grad_bar) * max(e_ref, GRAD_FLOOR, cond.get(n, 0.0) / scale if small else 0.0):
            bad[n] = (e_hip, e_ref)
    assert not bad, bad

    model.eval()
    with torch.no_grad():
        pe = model(inputs)
        pr = O.medtsllm_forward(x, p, sd, cfg, m, token_ids=tok_ids, pad_token_id=model.tokenizer.pad_token_id, training=False, **we_kw)
    assert rel_err(pe, pr) < bar


@pytest.mark.parametrize("kind,task,cov", [("gpt2", "forecasting", "concat"), ("llama_gqa", "semantic_segmentation", "add"),
                                           ("llama_gqa_bigvocab", "reconstruction", "concat")])
def test_pure_bf16_model_vs_oracle_on_rounded_parameters(kind, task, cov):
    """setup.dtype = "bf16": bf16 parameters + bf16 inputs through the same kernels, bf16 prediction and bf16 gradients"""
    _check_full_model(kind, task, 2, 64, 3, 16 if task == "forecasting" else 64, cov, "linear", True, pure_bf16=True)


def test_training_dropout_runs_and_is_unbiased():
    """training.dropout > 0 (the reference's shipped configs use 0.1): patch-embedding dropout + attention dropout of the
    reprogramming layer are active in train mode only; eval is deterministic and equals the p = 0 model."""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    cfg = hf_cfg("llama")
    sd = random_state_dict(cfg, seed=7, std=0.06)
    off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
    torch.manual_seed(3)
    model = model_lookup["medtsllm"](dict_to_object(model_config("forecasting", 64, 16, "concat", "linear", off, dropout=0.1)), FakeDataset(3),
                                     backbone_state=(cfg, sd)).to("cuda")
    x = torch.randn(4, 64, 3, generator=torch.Generator().manual_seed(5)).cuda()
    model.eval()
    with torch.no_grad():
        e1, e2 = model({"x_enc": x}), model({"x_enc": x})
    assert torch.equal(e1, e2)
    model.train()
    outs = []
    for _ in range(24):
        o = model({"x_enc": x})
        outs.append(o.detach())
    o.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)
    st = torch.stack(outs)
    assert float(st.std(dim=0).mean()) > 1e-4                       # stochastic in train mode
    assert rel_err(st.mean(dim=0), e1) < 0.15                       # and centred on the deterministic output


def _trainer_config(task, tmp_llm_dir, epochs=2, dropout=0.0):
    return {
        "DEBUG": True, "task": task, "model": "medtsllm", "history_len": 64, "pred_len": 16 if task == "forecasting" else 64,
        "data": {"dataset": "synthetic", "mode": "multivariate", "cols": "all", "normalize": True, "step": 8},
        "datasets": {"synthetic": {"n_features": 3, "n_windows": 32}},
        "training": {"epochs": epochs, "batch_size": 8, "optimizer": "adam", "learning_rate": 2e-3, "dropout": dropout,
                     "loss": "mse" if task != "semantic_segmentation" else "ce", "eval_metric": "loss", "eval_metric_direction": "min", "shuffle": False},
        "tasks": {"segmentation": {"mode": "boundary-prediction"}},
        "models": {"timellm": {"d_model": 8, "d_ff": 64, "n_heads": 2, "num_tokens": 64, "covariate_mode": "concat",
                               "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8},
                               "prompting": {"dataset": True, "task": True, "clip": False, "input_stats": True, "examples": False,
                                             "input_stats_dim": 0, "input_stats_select": "all"},
                               "llm": {"enabled": True, "llm": tmp_llm_dir, "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}},
        "setup": {"seed": 0, "device": "auto", "dtype": "mixed", "num_workers": 0, "logger": "print", "quiet": True},
    }


def _write_hf_dir(d, kind):
    """HuggingFace-format backbone directory (config.json + model.safetensors + tokenizer.json), as the reference expects."""
    import json
    import shutil
    from safetensors.torch import save_file
    from helpers import GOLDEN
    from med_ts_llm_amd.models.backbone import random_state_dict
    cfg = hf_cfg(kind)
    sd = random_state_dict(cfg, seed=5, std=0.05)
    (d / "config.json").write_text(json.dumps(cfg))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    shutil.copy(GOLDEN / "tokenizer.json", d / "tokenizer.json")
    (d / "tokenizer_config.json").write_text(json.dumps({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<|endoftext|>",
                                                         "eos_token": "<|endoftext|>"}))


def test_trainer_pure_bf16_dtype_trains(tmp_path):
    """setup.dtype = "bf16" (R:tasks/base.py:261-262): the product trainer casts the model and the batches to bf16, HipAdam updates the bf16
    parameters (fp32 moments), the loss goes down and the checkpoint holds bf16 tensors"""
    from med_ts_llm_amd.tasks import get_trainer
    from med_ts_llm_amd.utils import dict_to_object
    _write_hf_dir(tmp_path, "gpt2")
    cfg = _trainer_config("forecasting", str(tmp_path), epochs=3)
    cfg["setup"]["dtype"] = "bf16"
    trainer = get_trainer("DEBUG-bf16", dict_to_object(cfg))
    assert trainer.dtype == torch.bfloat16 and not trainer.mixed
    assert all(p.dtype == torch.bfloat16 for p in trainer.model.parameters())
    before = {n: p.detach().clone() for n, p in trainer.model.named_parameters() if p.requires_grad}
    trainer.train()
    losses = [h["train/loss"] for h in trainer.logger.history if "train/loss" in h]
    assert len(losses) >= 6 and all(l == l for l in losses)
    assert sum(losses[-3:]) / 3 < sum(losses[:3]) / 3
    moved = [n for n, p in trainer.model.named_parameters() if p.requires_grad and not torch.equal(p.detach(), before[n])]
    assert len(moved) >= len(before) - 2, moved
    st = trainer.optimizer.state[next(p for p in trainer.model.parameters() if p.requires_grad)]
    assert st["exp_avg"].dtype == torch.float32
    assert all(v.dtype == torch.bfloat16 for v in trainer.model.state_dict().values())
    scores = trainer.test()
    assert scores["test/loss"] == scores["test/loss"]


@pytest.mark.parametrize("kind,task", [("gpt2", "forecasting"), ("llama", "semantic_segmentation"), ("gpt2", "reconstruction")])
def test_trainer_end_to_end_on_gpu(tmp_path, kind, task):
    """a10 on the device: get_trainer(...).train() — the reference's loop body — runs on the HIP path from an on-disk
    HuggingFace-format backbone + tokenizer, the loss goes down, the checkpoint surface round-trips."""
    from med_ts_llm_amd.tasks import get_trainer
    from med_ts_llm_amd.utils import dict_to_object
    _write_hf_dir(tmp_path, kind)
    trainer = get_trainer("DEBUG-test", dict_to_object(_trainer_config(task, str(tmp_path), epochs=3)))
    assert trainer.device.type == "cuda" and trainer.mixed
    trainer.train()
    losses = [h["train/loss"] for h in trainer.logger.history if "train/loss" in h]
    assert len(losses) == 3 * 4 and trainer.step == 3 * 4 * 8          # step counter advances by batch_size per step
    assert all(l == l for l in losses)
    assert sum(losses[-4:]) < sum(losses[:4]), losses                      # the last epoch is better than the first
    scores = trainer.test()
    assert "test/loss" in scores
    sd = trainer.model.state_dict()
    assert not any(k.startswith("llm.") or k == "word_embeddings" for k in sd)
    trainer.model.load_state_dict(sd, strict=False)
    preds = trainer.predict(trainer.test_dataloader)
    assert preds.shape[0] == len(trainer.test_dataset) and torch.isfinite(preds).all()


def test_deferred_tail_optimizer_step_is_the_same_training_run(tmp_path):
    """HipAdam.defer: the tail's parameters (down-sample layer, flatten head) are updated on a side stream under the next step's front end and
    backbone; the model waits for it in front of the down-sample GEMM. Same kernels, same inputs: the run is bit-identical to one that updates
    everything on the main stream (the default: the overlap measured flat and is opt-in, setup.overlap_optimizer = true), losses and final weights."""
    from med_ts_llm_amd.tasks import get_trainer
    from med_ts_llm_amd.utils import dict_to_object
    _write_hf_dir(tmp_path, "gpt2")
    runs = []
    for overlap in (True, False):
        cfg = _trainer_config("forecasting", str(tmp_path), epochs=2)
        cfg["setup"]["overlap_optimizer"] = overlap
        tr = get_trainer("DEBUG-test", dict_to_object(cfg))
        assert (tr.model.optimizer_wait is not None) == overlap and bool(tr.optimizer._late) == overlap
        tr.train()
        runs.append(([h["train/loss"] for h in tr.logger.history if "train/loss" in h],
                     {n: p.detach().clone() for n, p in tr.model.named_parameters() if p.requires_grad}))
    assert runs[0][0] == runs[1][0]
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n


def test_linear_weight_shadows_track_the_masters(tmp_path):
    """the bf16 operand copies of the trainable Linear weights (written by HipAdam next to the fp32 masters, re-cast by the forward
    when a master changed behind the optimiser's back) always equal bf16(master) with zero K padding — after training steps, after
    load_state_dict, after an in-place edit — and predictions do not depend on who refreshed them"""
    from med_ts_llm_amd.tasks import get_trainer
    from med_ts_llm_amd.utils import dict_to_object
    from med_ts_llm_amd.hip import ops as _ops
    if not _ops._LINEAR_XT:
        pytest.skip("MTL_LINEAR_XT=0: the fallback Linear path keeps no weight shadows")
    _write_hf_dir(tmp_path, "gpt2")
    trainer = get_trainer("DEBUG-test", dict_to_object(_trainer_config("forecasting", str(tmp_path), epochs=1)))
    trainer.train()
    m = trainer.model

    def check():
        shadows = m.bf16_shadows()
        assert len(shadows) >= 6
        for sh in shadows:
            W = sh.param.detach()
            assert sh.fresh(), tuple(W.shape)
            assert torch.equal(sh.tensor[:, :W.shape[1]], W.to(torch.bfloat16))
            if sh.tensor.shape[1] > W.shape[1] + 1:              # (the mapping shadow keeps its bias in column V)
                assert not sh.tensor[:, W.shape[1] + 1:].any()
    check()                                                        # written by the optimiser
    batch = next(iter(trainer.test_dataloader))
    batch = {k: v.cuda() for k, v in batch.items() if torch.is_tensor(v)}
    m.eval()
    with torch.no_grad():
        p0 = m(batch).float()
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            m.output_projection.linear.weight.mul_(0.5)            # a master edited in place: stale shadow
        assert not m._linear_shadow(m.output_projection.linear).fresh()
        p1 = m(batch).float()
        assert float((p1 - p0).abs().max()) > 0
        m.load_state_dict(sd, strict=False)                        # and restored: stale again, re-cast by the next forward
        p2 = m(batch).float()
    assert torch.equal(p2, p0)
    check()


@pytest.mark.parametrize("task", ["forecasting", "anomaly_detection"])
def test_stitched_eval_on_gpu(tmp_path, task):
    """§8f-1 on the device: train one epoch on a sliding-window series dataset, then val()/test() through the HIP model
    with device-side window stitching (tasks/evalpath.py); scores are finite and the stitched arrays cover every point."""
    import numpy as np
    from med_ts_llm_amd.tasks import get_trainer
    from med_ts_llm_amd.tasks.windows import register_series
    from med_ts_llm_amd.utils import dict_to_object

    def source(config, split):
        g = np.random.default_rng({"train": 1, "val": 2, "test": 3}[split])
        n = 400
        t = np.arange(n, dtype=np.float32)
        data = np.stack([np.sin(t / 5), np.cos(t / 9), np.sin(t / 13) * 0.5], axis=-1).astype(np.float32) + 0.05 * g.standard_normal((n, 3)).astype(np.float32)
        lab = np.zeros(n, dtype=np.int64)
        lab[50:60] = 1
        lab[200:230] = 1
        data[lab == 1] += 2.0
        return {"data": data, "labels": lab if config.task == "anomaly_detection" else None}

    register_series("series_gpu", source)
    _write_hf_dir(tmp_path, "gpt2")
    cfg = _trainer_config(task, str(tmp_path), epochs=1)
    cfg["data"]["dataset"] = "series_gpu"
    cfg["training"]["eval_metric"] = "mse" if task == "forecasting" else "recon_mse"
    cfg["tasks"]["anomaly_detection"] = {"score_metric": "mse", "threshold": "auto", "normalize_by_feature": True, "normalize_moving_window": 0}
    trainer = get_trainer("DEBUG-eval", dict_to_object(cfg))
    assert trainer.device.type == "cuda"
    trainer.train()
    scores = trainer.test()
    if task == "forecasting":
        preds, targets = trainer.predict(trainer.test_dataloader)
        n = len(trainer.test_dataset)
        assert preds.shape == targets.shape == (16 + (n - 1) * 16, 3) and torch.isfinite(preds).all()
        assert np.isfinite(scores["test/mse"]) and np.isfinite(scores["test/mae"])
        x = trainer.test_dataset.data
        assert torch.equal(targets, x[64:64 + targets.shape[0]])        # stitched targets are the series itself
    else:
        assert {"test/f1", "test/auroc", "test/recon_mse", "test/anomaly_threshold"} <= set(scores)
        assert all(np.isfinite(v) for v in scores.values())


def _gpt2_masks(seed, attn_p, resid_p, B, T, H, d, L):
    from helpers import drop_mult_matrix, drop_mult_attention, drop_site_seed
    return {"attn": [drop_mult_attention(drop_site_seed(seed, i, 0), attn_p, B, H, T, T) for i in range(L)],
            "resid1": [drop_mult_matrix(drop_site_seed(seed, i, 1), resid_p, B * T, d).view(B, T, d) for i in range(L)],
            "resid2": [drop_mult_matrix(drop_site_seed(seed, i, 2), resid_p, B * T, d).view(B, T, d) for i in range(L)]}


@pytest.mark.parametrize("n_grad_extra", [0, 23, None])
def test_gpt2_stack_train_mode_dropouts(n_grad_extra):
    """a7, GPT-2 in train mode: attn_pdrop + resid_pdrop inside the stack (HF gpt2 :65,243,397). The kernels' counter-hash
    masks are rebuilt on the host and handed to the oracle as explicit multipliers, so forward and the input gradient
    (full and pruned) are compared exactly like the deterministic case."""
    from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
    from oracle import medtsllm_oracle as O
    cfg = hf_cfg("gpt2")
    sd = random_state_dict(cfg, seed=3, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda")
    B, T, n_last, d, H, L = 2, 100, 37, cfg["n_embd"], cfg["n_head"], cfg["n_layer"]
    attn_p, resid_p, seed = 0.1, 0.2, 987654321
    g = torch.Generator().manual_seed(5)
    h0 = torch.randn(B, T, d, generator=g)
    dout = torch.randn(B, n_last, d, generator=g).to(BF16)
    h0r = h0.clone().requires_grad_(True)
    ref = O.backbone_forward(h0r, sd, cfg, _gpt2_masks(seed, attn_p, resid_p, B, T, H, d, L))[:, -n_last:, :]
    (ref * dout.float()).sum().backward()
    h_in = (h0 + sd["wpe.weight"][:T]).cuda()
    drop = (attn_p, resid_p, seed)
    out, saved = bb.run_forward(h_in, n_last, drop=drop)
    assert rel_err(out.float(), ref) < L3
    plain, _ = bb.run_forward(h_in, n_last)
    assert rel_err(plain.float(), ref) > 10 * L3                      # the masks really are applied
    n_grad = None if n_grad_extra is None else n_last + n_grad_extra
    dh0 = bb.run_backward(h_in, dout.cuda(), saved, n_last, n_grad, drop=drop).cpu()
    lo = 0 if n_grad is None else T - n_grad
    assert rel_err(dh0[:, lo:], h0r.grad[:, lo:]) < 2 * L3


def test_full_model_gpt2_llm_dropout_train_vs_eval():
    """the model applies GPT-2's own dropouts (embd / attn / resid, from the backbone config) in train mode only; they are
    seeded per call, scale-preserving, switched off by eval() and by llm_dropout = False; gradients stay finite and alive"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    cfg = dict(hf_cfg("gpt2"), embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1)
    sd = random_state_dict(cfg, seed=7, std=0.06)
    off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
    torch.manual_seed(11)
    model = model_lookup["medtsllm"](dict_to_object(model_config("forecasting", 64, 16, "concat", "linear", off)), FakeDataset(3),
                                     backbone_state=(cfg, sd)).to("cuda")
    x = {"x_enc": torch.randn(4, 64, 3, generator=torch.Generator().manual_seed(2)).cuda()}
    model.eval()
    e1, e2 = model(x), model(x)
    assert torch.equal(e1, e2)
    model.train()
    torch.manual_seed(1); t1 = model(x)
    torch.manual_seed(2); t2 = model(x)
    torch.manual_seed(1); t3 = model(x)
    assert torch.equal(t1, t3) and not torch.equal(t1, t2)              # a function of the host RNG state only
    assert 1e-3 < rel_err(t1, e1) < 1.0                                   # perturbed (random tiny model: strongly), same scale
    t1.float().pow(2).mean().backward()
    for n, p_ in model.named_parameters():
        if p_.requires_grad:
            assert p_.grad is not None and torch.isfinite(p_.grad).all(), n
    model.llm_dropout = False
    assert torch.equal(model(x), e1)


def test_backbone_stack_at_gpt2_max_positions():
    """maximum size: T = n_positions = 1024 (GPT-2's learned-position limit; interleave-mode sequences get there) — the
    K/V-resident attention no longer fits the LDS, so the chunked causal kernels carry the stack; one more token is rejected"""
    from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
    from oracle import medtsllm_oracle as O
    cfg = dict(hf_cfg("gpt2"), n_positions=1024)
    sd = random_state_dict(cfg, seed=3, std=0.06)
    bb = FrozenBackbone(cfg, sd, "cuda")
    B, T, n_last, d = 1, 1024, 200, cfg["n_embd"]
    g = torch.Generator().manual_seed(5)
    h0 = torch.randn(B, T, d, generator=g)
    dout = torch.randn(B, n_last, d, generator=g).to(BF16)
    h0r = h0.clone().requires_grad_(True)
    ref = O.backbone_forward(h0r, sd, cfg)[:, -n_last:, :]
    (ref * dout.float()).sum().backward()
    h_in = (h0 + sd["wpe.weight"][:T]).cuda()
    out, saved = bb.run_forward(h_in, n_last)
    assert rel_err(out.float(), ref) < L3
    dh0 = bb.run_backward(h_in, dout.cuda(), saved, n_last, n_last + 56).cpu()
    lo = T - (n_last + 56)
    assert rel_err(dh0[:, lo:], h0r.grad[:, lo:]) < 2 * L3 and torch.all(dh0[:, :lo] == 0)
    with pytest.raises(ValueError):
        bb.run_forward(torch.zeros(1, 1025, d, device="cuda"), 1)


def test_trainable_vocabulary_shadows_equal_the_cast_path():
    """Vocabulary > 100 000 (Llama-3: the sub-sampled word-embedding table and the mapping weight both train, R:models/medtsllm.py:220-222):
    with the bf16 copies of both tables written by HipAdam (MedTsLLM.bf16_shadows -> MappingTrainableFn reads transposes of the shadows) four
    training steps give BIT-identical losses and parameters to the same steps with no shadow registered (every forward re-casts the fp32
    masters, the pre-shadow path); the shadows equal bf16(master) after every step, and an in-place edit of a master is picked up."""
    from med_ts_llm_amd.hip.optim import HipAdam
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    cfg = hf_cfg("llama_gqa", vocab=100_100)
    sd = random_state_dict(cfg, seed=7, std=0.06)
    off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
    g = torch.Generator().manual_seed(3)
    batches = [{"x_enc": torch.randn(2, 64, 2, generator=g).cuda()} for _ in range(2)]
    runs = []
    for with_shadows in (True, False):
        torch.manual_seed(11)
        model = model_lookup["medtsllm"](dict_to_object(model_config("reconstruction", 64, 64, "concat", "linear", off)), FakeDataset(2),
                                         backbone_state=(cfg, sd)).to("cuda")
        model.fixed_prompt_ids = torch.randint(0, cfg["vocab_size"], (1, 9), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
        model.train()
        assert model.word_embeddings.requires_grad and tuple(model.word_embeddings.shape) == (100_000, cfg["hidden_size"])
        opt = HipAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        shadows = model.bf16_shadows()
        vocab = [sh for sh in shadows if sh.param is model.word_embeddings or sh.param is model.mapping_layer.weight]
        assert len(vocab) == 2
        if with_shadows:
            for sh in shadows:
                opt.register_shadow(sh)
        losses = []
        for i in range(4):
            x = batches[i % 2]["x_enc"]
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = torch.nn.functional.mse_loss(model({"x_enc": x}), x)
            loss.backward()
            if with_shadows and i > 0:
                assert all(sh.fresh() for sh in vocab)            # this forward read the optimiser-written copies, no cast
            opt.step()
            opt.zero_grad()
            losses.append(float(loss.detach()))
            if with_shadows:
                for sh in vocab:
                    W = sh.param.detach()
                    assert sh.fresh() and torch.equal(sh.tensor[:, :W.shape[1]], W.to(torch.bfloat16))
        runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}, model, vocab))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][0][-1] < runs[0][0][0]
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n
    # a master edited behind the optimiser's back: the stale shadow is re-cast by the next forward
    _, _, model, vocab = runs[0]
    model.eval()
    with torch.no_grad():
        p0 = model(batches[0]).float()
        model.word_embeddings.mul_(0.5)
        assert not vocab[1].fresh() or not vocab[0].fresh()
        p1 = model(batches[0]).float()
        model.word_embeddings.mul_(2.0)
        p2 = model(batches[0]).float()
    assert float((p1 - p0).abs().max()) > 0 and torch.equal(p2, p0)
