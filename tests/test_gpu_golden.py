"""The HIP path against the REFERENCE's golden tensors, directly — no oracle in between.

tests/golden/case_*.npz were captured by importing the real reference (tests/golden/make_golden.py) at shapes the HIP path
accepts. Each case is loaded into the product's `MedTsLLM` (golden weights, golden inputs) and run on the device; every stage
the reference exposes, the loss and EVERY trainable gradient are compared with what the reference computed in fp32.

Bars (SURVEY.md 8c ladder, L3): the HIP path computes what the reference's default `dtype = "mixed"` computes (bf16 GEMM /
attention operands, fp32 accumulation, statistics and residual stream), so its distance to the reference's fp32 result is
held to 1.5 x the REFERENCE'S OWN mixed-vs-fp32 distance for the same tensor — `selferr.*` in the fixtures, measured by
running the reference itself under bf16 autocast — with a floor of one bf16 rounding per stage: every forward stage, the
prediction, the loss and every weight gradient. The yardstick is ONE sample of a random error, and so is the HIP error; for
gradients with few elements (bias vectors, the 8 x 16 x 3 patch convolution: < 4096 elements) the ratio of two such samples
scatters widely (measured over the 12 cases: up to 2.5), so those get 3 x per tensor, and the scatter is bounded the other way by an
aggregate criterion: over ALL gradients of a case the geometric mean of (HIP error / reference-mixed error) must be <= 1 —
on average the HIP path is at least as close to fp32 as the reference's own mixed mode (measured: 0.48 - 0.90). Sums with heavy cancellation (bias gradients, the 1 x C feature
weighting; the key bias is analytically zero) are additionally pinned exactly: the gradient the HIP path returns must be the
fp64 reduction of the HIP path's own upstream gradient.

The last test replays the reference TRAINER's golden run (8 Adam steps) through the product trainer on the device.
"""
import json

import numpy as np
import pytest
import torch

from helpers import (CASES, load_case, rel_err, golden_loss, fixture_tokenizer, big_grad_summary, cancellation_checks,
                     MIXED_FACTOR, FWD_FLOOR, GRAD_FLOOR, EXACT_SUM, SMALL_FACTOR, grad_factor)

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16

# (bars: tests/helpers.py "THE end-to-end parity bars")


def _cfg_from_meta(meta):
    from med_ts_llm_amd.utils import dict_to_object
    return dict_to_object({
        "DEBUG": True, "task": meta["task"], "model": "medtsllm", "history_len": meta["L"], "pred_len": meta["pred_len"],
        "training": {"dropout": 0.0}, "setup": {"dtype": "mixed"}, "tasks": {"segmentation": {"mode": "boundary-prediction"}},
        "models": {"timellm": {
            "d_model": meta["d_model"], "d_ff": meta["d_ff"], "n_heads": meta["n_heads"], "num_tokens": meta["num_tokens"],
            "covariate_mode": meta["covariate_mode"], "embedding_downsample_mode": meta["embedding_downsample_mode"],
            "patching": {"patch_len": meta["patch_len"], "stride": meta["stride"]}, "prompting": meta["prompting"],
            "llm": {"enabled": True, "llm": "fixture", "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}}})


class _DS:
    def __init__(self, meta):
        self.description, self.n_features, self.n_classes, self.task_description = meta["dataset_description"], meta["C"], meta["n_classes"], None


def _golden_model(name):
    from med_ts_llm_amd.models import model_lookup
    meta, data, bcfg, backbone = load_case(name)
    model = model_lookup["medtsllm"](_cfg_from_meta(meta), _DS(meta), backbone_state=(bcfg, backbone))
    model.tokenizer = fixture_tokenizer()
    sd = {k[len("param."):]: torch.from_numpy(v) for k, v in data.items() if k.startswith("param.") and k != "param.word_embeddings"}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"word_embeddings"}, (missing, unexpected)
    return model.to("cuda"), meta, data, bcfg


def _yard(data, key, ref, floor):
    """the reference's own mixed-vs-fp32 distance for this tensor (absolute norm), floored at `floor` x |ref|"""
    self_abs = float(data["selferr." + key]) if ("selferr." + key) in data else 0.0
    return max(self_abs, floor * float(np.linalg.norm(np.asarray(ref, dtype=np.float64))))


def canonical_prompts(prompts, L):
    """prompt part lists with every 'top 5 lags are [..]' list rewritten to twin-pair ids min(k, L - k)"""
    import re

    def canon(part):
        def sub(m):
            n = 2 * (L // 2)        # length of irfft's default output
            return "lags are " + str([min(int(k), n - int(k)) for k in m.group(1).split(",")])
        return re.sub(r"lags are \[([0-9, ]+)\]", sub, part)
    return [[canon(p) for p in ps] for ps in prompts]


def _abs(a, b):
    return float((torch.as_tensor(a).detach().cpu().double().flatten() - torch.as_tensor(b).detach().cpu().double().flatten()).norm())


@pytest.mark.parametrize("name", CASES)
def test_hip_model_vs_reference_golden(name):
    model, meta, data, bcfg = _golden_model(name)
    check_hip_vs_golden(model, meta, data, bcfg, name)


def check_hip_vs_golden(model, meta, data, bcfg, name, grad_bar=MIXED_FACTOR):
    """one train-mode forward + backward and one eval forward of the HIP model against a fixture captured from the reference (case_*.npz of
    make_golden.py, rw_*.npz of make_realwidth_golden.py: there meta["sampled"][key] = s says the fixture holds every s-th element of that tensor)"""
    B, C, P_, cm = meta["B"], meta["C"], None, meta["covariate_mode"]
    S = meta.get("sampled") or {}
    inputs = {"x_enc": torch.from_numpy(data["x_enc"]).cuda()}
    if next(p for p in model.parameters() if p.requires_grad).dtype == torch.bfloat16:       # setup.dtype = "bf16": the inputs are cast too
        inputs["x_enc"] = inputs["x_enc"].to(torch.bfloat16)
    if meta["descriptions"]:
        inputs["descriptions"] = meta["descriptions"]
    if "examples" in data:
        inputs["examples"] = [("Example segment:", torch.from_numpy(data["examples"][b:b + 1]).cuda()) for b in range(B)]
    # a6 / f2 on the device: the prompt built from DEVICE-computed statistics against the reference's strings. Every part must be
    # byte-identical, except that the order inside a twin pair of lags is not defined by the reference: the circular
    # autocorrelation is symmetric (corr[k] == corr[L-k] exactly), so which twin torch.topk lists first — and which twin of the
    # last pair is cut — is decided by 1-ulp FFT round-off and topk's tie handling (tools/lag_twin_noise.py shows the reference's
    # own CPU run going either way). Lags are therefore compared as twin-pair ids min(k, L-k).
    parts = [[p if isinstance(p, str) else "<TENSOR>" for p in ps] for ps in model.build_prompt(inputs)]
    assert canonical_prompts(parts, meta["L"]) == canonical_prompts(meta["prompts"], meta["L"])
    # downstream numerics are compared on the reference's exact token ids: hand the model the golden strings
    golden_parts = [[p if p != "<TENSOR>" else inputs["examples"][b][1] for p in ps] for b, ps in enumerate(meta["prompts"])]
    model.build_prompt = lambda _inputs: golden_parts

    model.train()
    tap = model.debug_tap = {}
    depth_keys = sorted((int(k.split(".")[1]), k) for k in (data.files if hasattr(data, "files") else list(data.keys())) if k.startswith("llm_hidden_after."))
    if depth_keys:
        tap["hidden_after"] = {layer: None for layer, _ in depth_keys}
    pred = model(inputs)
    report, failures, ratios = {}, [], []

    def check(key, got, ref, floor=FWD_FLOOR, extra_abs=0.0, factor=1.5):
        if key in S:
            got = torch.as_tensor(got).detach().float().flatten()[::S[key]]
        yard = max(_yard(data, key, ref, floor), extra_abs)
        e, bar = _abs(got, ref), factor * yard
        n = float(np.linalg.norm(np.asarray(ref, dtype=np.float64))) + 1e-30
        report[key] = (e / n, bar / n)
        if key.startswith("grad."):
            ratios.append(max(e, 1e-30) / yard)
        if not e <= bar:
            failures.append((key, e / n, bar / n))
        if ("mixed." + key) in data:
            # fixtures of round 6 also hold the reference-MIXED run's values: the HIP path's distance to the arithmetic it implements. Two
            # independent bf16 noises of the fp32 truth lie ~sqrt(2) x one noise apart; the bar is 2 x the reference's own mixed-vs-fp32 distance
            e_m = _abs(got, data["mixed." + key])
            report[key + " |to reference-mixed"] = (e_m / n, 2.0 * yard / n)
            if not e_m <= 2.0 * yard:
                failures.append((key + " |to reference-mixed", e_m / n, 2.0 * yard / n))

    # ---- stages (R:models/layers/RevIN.py, embed.py:186-197, medtsllm.py:281-282,349-350)
    d_patch = meta["d_model"]
    tok = tap["tokens"].float()
    n_p = tok.shape[1]
    if cm == "concat":          # the tokeniser writes the concat layout [B, P, C*d_patch (+pad)] directly
        tok = tok[:, :, :C * d_patch].reshape(B, n_p, C, d_patch).permute(0, 2, 1, 3).reshape(B * C, n_p, d_patch)
    else:
        tok = tok[:, :, :d_patch]
    check("patch_embed_out", tok, data["patch_embed_out"])
    check("source_embeddings", tap["source"].float(), data["source_embeddings"])
    check("reprog_out", tap["reprog"].float(), data["reprog_out"])
    h0 = tap["h0"].float()
    T = h0.shape[1]
    if bcfg["model_type"] == "gpt2":   # the assembly kernel adds GPT-2's learned positions; the reference's inputs_embeds is before them
        h0 = h0 - model.backbone.wpe[:T]
    check("llm_inputs_embeds", h0, data["llm_inputs_embeds"])
    n_last = tap["dec"].shape[1]
    # only the consumed rows get the final norm: compare them with the same rows of the reference's last_hidden_state
    self_rel = float(data["selferr.llm_last_hidden"]) / float(np.linalg.norm(data["llm_last_hidden"]))
    if "llm_last_hidden" in S:      # real-width fixtures hold (a sample of) the consumed rows only
        assert n_last == meta["n_patches"]
        check("llm_last_hidden", tap["dec"].float(), data["llm_last_hidden"], max(self_rel, FWD_FLOOR))
    else:
        check("llm_last_hidden[consumed rows]", tap["dec"].float(), data["llm_last_hidden"][:, -n_last:, :], max(self_rel, FWD_FLOOR))
    # intermediate depths of the stack (full-depth fixtures, round 6): the pre-norm residual stream after layers 1 / 8 / 16 / 24 on the consumed rows,
    # each against the reference's own mixed-arithmetic deviation AT THAT DEPTH — a failure of the last hidden state localises
    for layer, key in depth_keys:
        got = tap["hidden_after"][layer]
        assert got is not None and got.shape[1] == meta["n_patches"]
        rel_self = float(data["selferr." + key]) / float(np.linalg.norm(data[key]))
        check(key, got.float(), data[key], max(rel_self, FWD_FLOOR))
    assert "pred_train" in S or pred.shape == data["pred_train"].shape
    check("pred_train", pred, data["pred_train"])

    # ---- loss + every gradient
    loss = golden_loss(pred, torch.from_numpy(data["target"]).cuda(), meta["task"])
    loss.backward()
    ref_loss = float(data["loss"])
    report["loss"] = (abs(loss.item() - ref_loss) / abs(ref_loss), MIXED_FACTOR * max(float(data["selferr.loss"]), 5e-3 * abs(ref_loss)) / abs(ref_loss))
    if not report["loss"][0] <= report["loss"][1]:
        failures.append(("loss",) + report["loss"])
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    assert all(g is not None for g in grads.values())

    # Gradients that are SUMS WITH CANCELLATION over the rows of an upstream gradient A (bias gradients: sum_r A[r, :]; the tiny
    # feature-weighting layer: A^T X): a relative perturbation eps of the elements of A moves the sum by ~ eps * |A|_F * |X|_F /
    # sqrt(rows) (X = ones for a bias) however small the sum itself is — the key bias sums to exactly zero analytically. Such a
    # gradient is therefore (1) pinned EXACTLY against the fp64 reduction of the HIP path's own upstream gradient, and (2) compared
    # with the reference on the scale of that upstream mass instead of on the scale of the cancelled result.
    exact, cond = cancellation_checks(tap, grads, GRAD_FLOOR)
    # (bf16 parameters — setup.dtype = "bf16" — carry bf16 gradients: one more rounding of 2^-9 of each element, on the scale of the sum's L1 mass /
    #  sqrt(rows); same bar as tests/test_gpu_model.py)
    exact_bar = 4e-3 if next(iter(grads.values())).dtype == torch.bfloat16 else EXACT_SUM
    for n, (e, mass) in exact.items():
        report["exact:" + n] = (e / (mass + 1e-30), exact_bar)
        if not e <= exact_bar * mass + 1e-9:
            failures.append(("exact:" + n, e / (mass + 1e-30), exact_bar))

    n_checked = 0
    for k in data:
        if k.startswith("grad."):
            n = k[len("grad."):]
            check(k, grads[n], data[k], GRAD_FLOOR, cond.get(n, 0.0), grad_factor(data[k].size, grad_bar))
            n_checked += 1
        elif k.startswith("gradnorm."):           # 100 000-wide gradients: projections along both axes + strided slices
            n = k[len("gradnorm."):]
            norm, prow, pcol, sample = big_grad_summary(grads[n].reshape(grads[n].shape[0], -1), meta["synth"]["stride"])      # (a 51 200-long bias is summarised as [n, 1])
            self_rel = float(data["selferr.grad." + n]) / float(data[k])
            tol = grad_bar * max(self_rel, GRAD_FLOOR)
            report[f"grad.{n}[norm]"] = (abs(norm - float(data[k])) / float(data[k]), tol)
            for what, got, key in (("rows", prow, "gradproj_rows." + n), ("cols", pcol, "gradproj_cols." + n), ("sample", sample, "gradsample." + n)):
                want = data[key]
                if "selferr." + key in data:
                    # real-width fixtures carry the reference-mixed step's deviation of THIS view. A weight gradient of a layer that sees one row per
                    # sample (the flatten head) is a sum of B outer products: its projection is a B-sample statistic, and a strided sample of two rows
                    # of a [1024, 768] gradient a two-row statistic of an error spread over all rows — the small-tensor factor, on the larger of the
                    # view's own and the whole tensor's relative deviation
                    wn = float(np.linalg.norm(want)) + 1e-30
                    bar = SMALL_FACTOR * max(float(data["selferr." + key]) / wn, self_rel, GRAD_FLOOR)
                else:
                    # a projection onto a fixed vector keeps the relative error of the full tensor up to a random factor: 2 x
                    bar = 2 * tol
                report[f"grad.{n}[{what}]"] = (rel_err(got, want), bar)
            for key in (f"grad.{n}[norm]", f"grad.{n}[rows]", f"grad.{n}[cols]", f"grad.{n}[sample]"):
                if not report[key][0] <= report[key][1]:
                    failures.append((key,) + report[key])
            n_checked += 1
    assert n_checked == len(grads), (n_checked, sorted(grads))
    gmean = float(np.exp(np.mean(np.log(ratios))))
    report["grad geometric-mean(error / reference-mixed error)"] = (gmean, 1.0)
    if not gmean <= 1.0:
        failures.append(("gradient aggregate", gmean, 1.0))

    # ---- eval mode (a9 activations included)
    model.eval()
    model.debug_tap = None
    with torch.no_grad():
        pe = model(inputs)
    self_rel = float(data["selferr.pred_train"]) / float(np.linalg.norm(data["pred_train"]))
    check("pred_eval", pe, data["pred_eval"], max(self_rel, FWD_FLOOR))
    print(f"\n[{name}] (error, bar) relative to |reference|: " + json.dumps({k: (float(f"{v[0]:.3g}"), float(f"{v[1]:.3g}")) for k, v in report.items()}))
    assert not failures, (name, failures)


def test_hip_patch_index_map_vs_reference_golden():
    from med_ts_llm_amd.hip import ops
    for name in CASES:
        meta, data, _, _ = load_case(name)
        idx = ops.patch_index_map(meta["L"], meta["patch_len"], meta["stride"], "cuda").cpu().numpy()
        assert idx.dtype == np.int32 and np.array_equal(idx, data["patch_index_map"])


@pytest.mark.parametrize("prompt_stats", ["host", "device"])
def test_product_trainer_replays_reference_trajectory_on_gpu(tmp_path, prompt_stats):
    """a10 on the device: tasks.get_trainer(...).train() — BaseTask.train_step with the HIP model, HipAdam, bf16 autocast —
    from the reference trainer's initial weights over its batches: 8 per-step losses and the final weights. The reference ran
    in fp32; the bar is the mixed-precision ladder (loss values within 1 %, every weight within 10 % of the distance it moved).
    prompt_stats = "device": the run trains with the prompts built from the DEVICE statistics kernels (mtl_input_stats, f2) — what the product
    does; every prompt of every batch must equal the host-built one up to the order inside twin pairs of lags (see above). A swapped twin changes
    prompt tokens (lag k printed as L - k), which this 2-layer toy model feels: the run is then a different — equally valid — trajectory, so only
    its level is compared (every loss within 20 %, their mean within 3 % of the reference's; measured: mean 0.7401 vs 0.7415)."""
    from test_host_logic import golden_trainer_setup, load_golden_init
    from med_ts_llm_amd.tasks import get_trainer
    cfg, z, n_batches = golden_trainer_setup(tmp_path, "cuda", "mixed")
    trainer = get_trainer("DEBUG-golden-gpu", cfg)
    assert trainer.device.type == "cuda" and trainer.mixed and type(trainer.optimizer).__name__ == "HipAdam"
    load_golden_init(trainer, z)
    # the reference trainer ran on the CPU: its prompts carry the CPU's choice inside every twin pair of lags (see above). The
    # product's prompt builder on a host copy of the batch reproduces those strings byte for byte (tests/test_host_logic.py)
    build = trainer.model.build_prompt
    seen = []
    if prompt_stats == "host":
        trainer.model.build_prompt = lambda inputs: build({**inputs, "x_enc": inputs["x_enc"].cpu()})
    else:
        def build_checked(inputs):
            dev, host = build(inputs), build({**inputs, "x_enc": inputs["x_enc"].cpu()})
            Lx = inputs["x_enc"].shape[1]
            strs = lambda pr: [[q if isinstance(q, str) else "<T>" for q in ps] for ps in pr]
            assert canonical_prompts(strs(dev), Lx) == canonical_prompts(strs(host), Lx)
            seen.append(strs(dev) == strs(host))
            return dev
        trainer.model.build_prompt = build_checked
    trainer.train()
    losses = np.array([h["train/loss"] for h in trainer.logger.history if "train/loss" in h])
    assert len(losses) == 2 * n_batches == len(z["losses"])
    print("\nloss trajectory  hip:", np.round(losses, 5).tolist(), "\n            reference:", np.round(z["losses"], 5).tolist())
    if seen:
        print("device-built prompts byte-identical to the host-built ones in", sum(seen), "of", len(seen), "batches")
    if prompt_stats == "host":
        assert np.allclose(losses, z["losses"], rtol=1e-2, atol=1e-5), (losses, z["losses"])
    else:
        assert len(seen) >= 2 * n_batches
        assert np.allclose(losses, z["losses"], rtol=0.2) and abs(losses.mean() / z["losses"].mean() - 1.0) < 0.03, (losses, z["losses"])
        return
    p = dict(trainer.model.named_parameters())
    worst = {}
    for k in z.files:
        if k.startswith("final."):
            n = k[len("final."):]
            if n.endswith("key_projection.bias"):
                continue       # analytically-zero gradient: Adam turns pure round-off into +-lr steps (not reproducible, also not in the reference)
            moved = float(np.linalg.norm(z[k] - z["init." + n]))
            worst[n] = float((p[n].detach().cpu() - torch.from_numpy(z[k])).norm()) / (moved + 1e-12)
    print("final weights, distance to the reference's / distance moved:", {k: round(v, 4) for k, v in worst.items()})
    assert max(worst.values()) < 0.10, worst
    assert trainer.step == int(z["step_counter"])


def _gpu_losses_close(tr, z, rtol=1.5e-2):
    losses = np.array([h["train/loss"] for h in tr.logger.history if "train/loss" in h])
    print("\nloss trajectory  hip:", np.round(losses, 5).tolist(), "\n            reference:", np.round(z["losses"], 5).tolist())
    assert len(losses) == len(z["losses"]) and np.allclose(losses, z["losses"], rtol=rtol, atol=1e-5), (losses, z["losses"])


def test_pretraining_trainer_replays_reference_on_gpu(tmp_path):
    """R:tasks/pretraining.py on the device: the mixed-dataset windows the reference drew, through the HIP model + HipAdam in bf16 mixed
    mode: the reference's (fp32) per-step losses within 1.5 %, final weights within the mixed-precision ladder"""
    import test_datasets_trainer as T
    from med_ts_llm_amd.tasks import get_trainer
    G = (dict(np.load(T.GOLDEN / "datasets.npz")), json.loads((T.GOLDEN / "datasets.json").read_text()))
    z = np.load(T.GOLDEN / "trainer_pretraining.npz")
    T.register_part_sources(G)
    tr = get_trainer("DEBUG-pretrain-gpu", T.pretraining_config(tmp_path, device="cuda", dtype="mixed"))
    assert tr.device.type == "cuda" and tr.task == "pretraining" and type(tr.optimizer).__name__ == "HipAdam"
    for j, inds in enumerate(tr.train_dataset.dataset_inds):
        assert np.array_equal(inds.numpy(), z[f"mix.inds{j}"])
    T._load_init(tr, z)
    tr.train()
    _gpu_losses_close(tr, z)
    T._check_final(tr, z, tol_each=0.25, tol_all=0.10)


def test_finetuning_trainer_replays_reference_on_gpu(tmp_path):
    """R:tasks/base.py:88-91,118-155 on the device: pre-trained front end + fresh head, two HipAdam parameter groups, warm-up schedule"""
    import test_datasets_trainer as T
    from med_ts_llm_amd.tasks import get_trainer
    G = (dict(np.load(T.GOLDEN / "datasets.npz")), json.loads((T.GOLDEN / "datasets.json").read_text()))
    z = np.load(T.GOLDEN / "trainer_finetune.npz")
    meta = json.loads((T.GOLDEN / "trainer_finetune.json").read_text())
    T.register_part_sources(G)
    T.write_pretrained_checkpoint(tmp_path, z)
    tr = get_trainer("DEBUG-finetune-gpu", T.finetune_config(tmp_path, device="cuda", dtype="mixed"))
    assert tr.finetuning and tr.loaded_params == meta["loaded_params"] and type(tr.optimizer).__name__ == "HipAdam"
    named = {id(p): n for n, p in tr.model.named_parameters()}
    assert [[named[id(p)] for p in g["params"]] for g in tr.optimizer.param_groups] == meta["groups"]
    T._load_init(tr, z)
    lrs = []
    orig = tr.log_epoch
    tr.log_epoch = lambda scores={}, **kw: (lrs.append(list(tr.scheduler.get_last_lr())), orig(scores, **kw))[1]
    tr.train()
    assert np.allclose(lrs, z["lrs"], rtol=1e-6)
    _gpu_losses_close(tr, z)
    T._check_final(tr, z, tol_each=0.25, tol_all=0.10)


@pytest.mark.parametrize("dtype", ["mixed", "bf16"])
def test_resume_restores_hip_adam_state_on_gpu(tmp_path, dtype):
    """checkpoint -> from_run_id -> one more epoch == an uninterrupted run, with HipAdam's moments travelling through the checkpoint.
    setup.dtype = "bf16": bf16 parameters, but the moments stay fp32 through save / load (mtl_adam_step reads them as float*)"""
    import test_datasets_trainer as T
    from med_ts_llm_amd.tasks import get_trainer, task_lookup
    G = (dict(np.load(T.GOLDEN / "datasets.npz")), json.loads((T.GOLDEN / "datasets.json").read_text()))
    T.register_part_sources(G)
    llm = T._golden_backbone_dir(tmp_path)

    def cfg(epochs):
        return T.base_cfg("reconstruction", "bidmc", llm_dir=llm, epochs=epochs,
                          extra={"DEBUG": False, "paths": {"logdir": str(tmp_path / "logs")},
                                 "setup": {"seed": 0, "device": "cuda", "dtype": dtype, "num_workers": 0, "logger": "print", "quiet": True}})
    full = get_trainer("run-full", cfg(2))
    init = {k: v.detach().clone() for k, v in full.model.state_dict().items()}
    full.train()
    part = get_trainer("run-part", cfg(1))
    part.model.load_state_dict(init, strict=False)
    part.train()
    resumed = task_lookup["reconstruction"].from_run_id("run-part", cfg={"training": cfg(2).training.to_dict()}, basepath=str(tmp_path / "logs"))
    assert type(resumed.optimizer).__name__ == "HipAdam" and all(int(st["step"]) == 2 for st in resumed.optimizer.state.values())
    assert resumed.epochs_done == 1
    for q, st in resumed.optimizer.state.items():
        assert q.dtype == (torch.bfloat16 if dtype == "bf16" else torch.float32)
        assert st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].dtype == torch.float32 and st["exp_avg"].numel() == q.numel()
    # torch's own Optimizer.load_state_dict narrows floating state to the parameter dtype: HipAdam's override widens it again
    resumed.optimizer.load_state_dict(resumed.optimizer.state_dict())
    assert all(st["exp_avg"].dtype == torch.float32 for st in resumed.optimizer.state.values())
    resumed.train()                              # continues with epoch 2 of 2
    for (n, a), (_, b) in zip(full.model.named_parameters(), resumed.model.named_parameters()):
        if a.requires_grad:
            assert torch.equal(a, b), n          # same kernels, same inputs, same moments: bit-identical
