"""The REAL reference at the widths it is run at: tests/golden/rw_*.{npz,json} were captured by importing /root/reference
(tests/golden/make_realwidth_golden.py) with a GPT-2-small-width (two layers, and the full 12), a Llama-2-7B-width and a Llama-3-8B-width backbone (two
layers each), on the metric workload's window geometry ([L = 1024, C = 12], d_model 32, d_ff 128, 8 heads, 1024 prototypes, dataset + task prompt) plus
`independent` / `add` / `weighted-average` covariates, the `truncate` down-sample, input-statistics prompts and the segmentation / reconstruction heads. Every weight is formula-generated
(helpers.rw_backbone_state / rw_trainable_values — the generator imported the same functions), so the fixtures hold inputs, expected outputs,
sampled stage tensors and gradient summaries only.

Until round 4 the real-width comparisons (tests/test_gpu_realwidth.py) were HIP vs the ORACLE, and the oracle itself was pinned to the reference
only at d_llm 128: these tests close that gap.
  * CPU suite: the oracle against the reference on the four GPT-2-small-width fixtures (fp32, <= 2e-5 — the L1 rung of SURVEY.md 8c);
  * GPU suite: the HIP path against the reference on all seven (bar = 1.5 x the reference's own bf16-autocast deviation, exactly as
    tests/test_gpu_golden.py), and the oracle against the three Llama-width fixtures on the GPU box's host cores (6 - 12 GB, minutes: too large for the CPU suite).
"""
import numpy as np
import pytest
import torch

from helpers import RW_CASES, RW_CASES_CPU, LONGT_GRAD_FACTOR, load_rw_case, oracle_mcfg, golden_loss, rel_err, abs_err, big_grad_summary, fixture_tokenizer

TOL = 2e-5         # fp32 reductions over K = 50 257 / 16 384 / 11 008 in two different summation orders


def _sample(meta, key, t):
    s = (meta.get("sampled") or {}).get(key)
    return t if s is None else torch.as_tensor(t).detach().float().flatten()[::s]


def _oracle_vs_reference(name):
    from oracle import medtsllm_oracle as O
    # the oracle is the CPU restatement, also on the GPU box: every tensor it sees lives on the host. Only the SYNTHESIS of the formula-generated backbone
    # weights runs on the device when there is one (bit-identical to the host generator: test_device_weight_generator_is_the_host_generator) — 6.6 G
    # numbers take minutes on the host and seconds in HBM
    meta, data, bcfg, backbone, params = load_rw_case(name, device="cuda" if torch.cuda.is_available() else "cpu")
    backbone = {k: v.cpu() for k, v in backbone.items()}
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    torch.set_num_threads(min(64, torch.get_num_threads()))      # (256 host threads on [512, 4096]-sized operands only contend)
    m = oracle_mcfg(meta)
    # full-depth stacks: fp32 summation-order noise (the oracle's GEMMs on 64 threads vs the reference's on the build container's 8) is amplified through
    # 32 random-weight layers exactly as the bf16 noise is (x 12 at the last hidden state: profiles/r05_fulldepth_reference_parity.txt) — 4 x the bars of
    # the two-layer cases (measured: 7.9e-5 on one gradient projection of the Llama-3-8B stack, everything else < 5e-5)
    deep = 4.0 if bcfg.get("num_hidden_layers", 0) >= 16 else 1.0
    TOL = globals()["TOL"] * deep
    GTOL = 5e-5 * deep
    p = {n: v.clone().requires_grad_(True) for n, v in params.items()}
    x = torch.from_numpy(data["x_enc"])
    mean, stdev = O.revin_stats(x)
    assert rel_err(mean, data["revin_mean"]) < 1e-6 and rel_err(stdev, data["revin_stdev"]) < 1e-6
    pe = O.patch_embed(O.revin_norm(x, mean, stdev), p["patch_embedding.value_embedding.tokenConv.weight"], meta["patch_len"], meta["stride"])
    assert rel_err(_sample(meta, "patch_embed_out", pe), data["patch_embed_out"]) < TOL
    we_kw = {"word_emb": p["word_embeddings"]} if "word_embeddings" in p else {}        # trainable table (vocabulary > 100 000)
    pred, inter = O.medtsllm_forward(x, p, backbone, bcfg, m, token_ids=meta["prompt_token_ids"], pad_token_id=meta["pad_token_id"],
                                     training=True, return_intermediates=True, **we_kw)
    assert inter["llm_inputs_embeds"].shape[1] == meta["T"]
    assert rel_err(_sample(meta, "llm_inputs_embeds", inter["llm_inputs_embeds"]), data["llm_inputs_embeds"]) < TOL
    with torch.no_grad():
        last = O.backbone_forward(inter["llm_inputs_embeds"], backbone, bcfg)[:, -meta["n_patches"]:, :]
        src = O.source_embeddings(p["word_embeddings"] if we_kw else O.word_embeddings_of(backbone, bcfg), p["mapping_layer.weight"], p["mapping_layer.bias"])
    assert rel_err(_sample(meta, "llm_last_hidden", last), data["llm_last_hidden"]) < TOL
    assert rel_err(_sample(meta, "source_embeddings", src), data["source_embeddings"]) < TOL
    assert rel_err(_sample(meta, "pred_train", pred), data["pred_train"]) < TOL
    loss = golden_loss(pred, data["target"], meta["task"])
    assert abs(loss.item() - float(data["loss"])) < 1e-5 * deep * max(1.0, abs(float(data["loss"])))
    loss.backward()
    n_checked = 0
    for k, v in data.items():
        if k.startswith("grad."):
            n = k[len("grad."):]
            # (the key bias has an analytically-zero gradient — softmax shift invariance — hence the absolute floor)
            assert abs_err(p[n].grad, v) < GTOL * float(np.linalg.norm(v)) + 1e-7, (n, rel_err(p[n].grad, v))
            n_checked += 1
        elif k.startswith("gradnorm."):
            n = k[len("gradnorm."):]
            g = p[n].grad
            norm, prow, pcol, sample = big_grad_summary(g.reshape(g.shape[0], -1), meta["synth"]["stride"])
            assert abs(norm - float(v)) < GTOL * float(v), n
            for got, want in ((prow, data["gradproj_rows." + n]), (pcol, data["gradproj_cols." + n]), (sample, data["gradsample." + n])):
                assert abs_err(got, want) < GTOL * float(np.linalg.norm(want)) + 1e-7, (n, rel_err(got, want))
            n_checked += 1
    assert n_checked == len(p)
    with torch.no_grad():
        pe_eval = O.medtsllm_forward(x, p, backbone, bcfg, m, token_ids=meta["prompt_token_ids"], pad_token_id=meta["pad_token_id"], training=False, **we_kw)
    assert rel_err(pe_eval, data["pred_eval"]) < TOL


@pytest.mark.parametrize("name", RW_CASES_CPU)
def test_oracle_vs_reference_real_width(name):
    """GPT-2-small width: the metric model cut to two layers and at its full 12, `independent` covariates with input-statistics prompts, `add` with the
    `truncate` down-sample and the boundary-segmentation head"""
    _oracle_vs_reference(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in RW_CASES if n not in RW_CASES_CPU])
def test_oracle_vs_reference_real_width_large(name):
    """The oracle against the reference at the Llama widths — and at the FULL depth of BASELINE.json configs[2] (32 layers of Llama-2-7B, 6.6 G
    formula-generated weights) — on the GPU box's host cores for their size only (no device code runs): 6 - 30 GB of fp32 weights, a [1024, 32000]
    mapping layer, Llama-3-8B's 100 000 TRAINABLE vocabulary rows with their gradients. Every case runs in the default `-m gpu` suite since round 5
    (the weights come from helpers.synth_table_torch, multi-threaded: the numpy generator alone took minutes per case); the build container's 8 cores
    take 1 - 10 minutes per case, hence the gpu marker."""
    _oracle_vs_reference(name)


@pytest.mark.gpu
def test_device_weight_generator_is_the_host_generator():
    """the formula-generated weights written straight into HBM (what the full-depth cases load) == numpy's, bit for bit"""
    import helpers as H
    for rows, cols, salt, scale, row0 in [(777, 4096, 1003, 0.07, 0), (5, 11008, 1005, 0.07, 4090), (1, 4096, 1290, 0.2, 0), (300, 64, 1000, 0.07, 127990)]:
        assert np.array_equal(H.synth_table(rows, cols, salt, scale, row0=row0), H.synth_table_torch(rows, cols, salt, scale, row0=row0, device="cuda").cpu().numpy())
    a = H.synth_tensor(20000, 96, 9, 0.3, device="cuda").cpu()
    assert torch.equal(a, H.synth_tensor(20000, 96, 9, 0.3))


@pytest.mark.gpu
@pytest.mark.parametrize("name", RW_CASES)
def test_hip_vs_reference_real_width(name):
    from med_ts_llm_amd.models import model_lookup
    from test_gpu_golden import check_hip_vs_golden, _cfg_from_meta, _DS
    meta, data, bcfg, backbone, params = load_rw_case(name, device="cuda")       # (backbone weights generated in HBM: bit-identical to the host generator)
    model = model_lookup["medtsllm"](_cfg_from_meta(meta), _DS(meta), backbone_state=(bcfg, backbone))
    model.tokenizer = fixture_tokenizer()
    assert {n: tuple(q.shape) for n, q in model.named_parameters() if q.requires_grad} == {n: tuple(s) for n, s in meta["param_table"].items()}
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected and set(missing) <= {"word_embeddings"}, (missing, unexpected)
    del backbone, params
    # long sequences (interleave: ~1 600 query rows per sample against 64 shared prototypes): helpers.LONGT_GRAD_FACTOR, as tests/test_gpu_longT.py
    # fixtures whose second reference run is the all-bf16 one (R:tasks/base.py:261-262): the HIP model in its native bf16 mode (bf16 parameters, bf16
    # inputs, bf16 residual stream) against the fp32 reference, yardstick = the reference-bf16 run's own deviation
    model = model.to("cuda", torch.bfloat16) if meta.get("second_run") == "bf16" else model.to("cuda")
    check_hip_vs_golden(model, meta, data, bcfg, name, **({"grad_bar": LONGT_GRAD_FACTOR} if meta["covariate_mode"] == "interleave" else {}))


# ----------------------------------------------------------------------------- a10 at the real width: the reference TRAINER's 8-step run
def _rw_trainer_setup(tmp_path, device, dtype, model_key="medtsllm"):
    """what the product trainer needs to replay tests/golden/rw_trainer_gpt2s_2l_fc (make_realwidth_golden.run_trainer): the formula-generated backbone as
    an HF directory, a dataset that yields the reference's 16 windows in its order, the reference run's configuration"""
    import json
    from pathlib import Path
    from torch.utils.data import Dataset
    from helpers import GOLDEN, RW_BACKBONES, rw_backbone_state, write_hf_dir
    from med_ts_llm_amd.tasks.synthetic import register_dataset
    from med_ts_llm_amd.utils import dict_to_object
    meta = json.loads((GOLDEN / "rw_trainer_gpt2s_2l_fc.json").read_text())
    z = np.load(GOLDEN / "rw_trainer_gpt2s_2l_fc.npz")
    bcfg = RW_BACKBONES[meta["backbone"]]
    d = write_hf_dir(Path(tmp_path) / "llm_rw", bcfg, rw_backbone_state(bcfg))
    xs, ys = torch.from_numpy(z["x_enc"]), torch.from_numpy(z["y"])

    class GoldenWindows(Dataset):
        description, n_features, n_classes, task_description = meta["dataset_description"], meta["C"], 0, None

        def __len__(self):
            return xs.shape[0]

        def __getitem__(self, i):
            return {"x_enc": xs[i], "y": ys[i]}

    register_dataset("golden_trainer_rw", lambda config, split: GoldenWindows())
    cfg = dict_to_object({
        "DEBUG": True, "task": "forecasting", "model": model_key, "history_len": meta["L"], "pred_len": meta["pred_len"],
        "data": {"dataset": "golden_trainer_rw"},
        "training": {"epochs": meta["epochs"], "batch_size": meta["batch_size"], "optimizer": "adam", "learning_rate": meta["learning_rate"], "dropout": 0.0,
                     "loss": "mse", "eval_metric": "loss", "eval_metric_direction": "min", "shuffle": False},
        "tasks": {"segmentation": {"mode": "boundary-prediction"}},
        "models": {"timellm": {
            "d_model": meta["d_model"], "d_ff": meta["d_ff"], "n_heads": meta["n_heads"], "num_tokens": meta["num_tokens"],
            "covariate_mode": "concat", "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8}, "prompting": meta["prompting"],
            "llm": {"enabled": True, "llm": str(d), "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}},
        "setup": {"seed": 0, "device": device, "dtype": dtype, "num_workers": 0, "logger": "print", "quiet": True}})
    return cfg, z, meta


def _rw_trainer_run(trainer, z, meta):
    """formula-generated initial weights in, train, -> (losses, {name: distance to the reference's final value / distance the reference moved, over the stored sample})"""
    from helpers import rw_trainable_values
    init = rw_trainable_values([(n, tuple(s)) for n, s in meta["param_table"].items()])
    assert {n: tuple(q.shape) for n, q in trainer.model.named_parameters() if q.requires_grad} == {n: tuple(s) for n, s in meta["param_table"].items()}
    missing, unexpected = trainer.model.load_state_dict(init, strict=False)
    assert not unexpected and set(missing) <= {"word_embeddings"}, (missing, unexpected)
    trainer.train()
    losses = np.array([h["train/loss"] for h in trainer.logger.history if "train/loss" in h])
    p = dict(trainer.model.named_parameters())
    off = {}
    for n, s in meta["sampled"].items():
        if n.endswith("key_projection.bias"):
            continue       # analytically-zero gradient: Adam turns pure round-off into +-lr steps (not reproducible, also not in the reference)
        got = p[n].detach().float().cpu().flatten()[::s].double()
        off[n] = float((got - torch.from_numpy(z["final." + n]).double()).norm()) / (float(z["movedsample." + n]) + 1e-30)
    assert trainer.step == int(z["step_counter"])
    return losses, off


def test_product_trainer_replays_reference_trajectory_at_gpt2_small_width(tmp_path):
    """CPU suite: the product trainer (loop, loss, optimiser construction, step order) with the device math swapped for the pinned oracle"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer
    from helpers import register_oracle_math_model
    key = register_oracle_math_model()
    try:
        cfg, z, meta = _rw_trainer_setup(tmp_path, "cpu", "fp32", key)
        losses, off = _rw_trainer_run(get_trainer("DEBUG-golden-rw", cfg), z, meta)
    finally:
        del model_lookup[key]
    print("\nlosses", np.round(losses, 6).tolist(), "\nreference", np.round(z["losses"], 6).tolist(), "\nfinal weights off / moved:", {k: round(v, 5) for k, v in off.items()})
    # (importing the reference's tasks package sets float32 matmul precision "medium": the golden trajectory carries ~3e-4 of CPU matmul noise)
    assert len(losses) == len(z["losses"]) and np.allclose(losses, z["losses"], rtol=1e-3, atol=1e-6), (losses, z["losses"])
    assert max(off.values()) < 0.08, off       # (measured <= 0.048: the value projection, whose gradient is a cancellation-prone sum — DESIGN §3)


@pytest.mark.gpu
def test_hip_trainer_replays_reference_trajectory_at_gpt2_small_width(tmp_path):
    """GPU suite: tasks.get_trainer(...).train() with the HIP model, HipAdam and bf16 mixed arithmetic against the reference trainer's fp32 run; the
    reference's own dtype = "mixed" run of the same 8 steps (losses_mixed in the fixture) deviates by up to 8e-4 relative per loss"""
    from med_ts_llm_amd.tasks import get_trainer
    cfg, z, meta = _rw_trainer_setup(tmp_path, "cuda", "mixed")
    trainer = get_trainer("DEBUG-golden-rw-gpu", cfg)
    assert trainer.device.type == "cuda" and trainer.mixed and type(trainer.optimizer).__name__ == "HipAdam"
    losses, off = _rw_trainer_run(trainer, z, meta)
    mixed_dev = float(np.max(np.abs(z["losses_mixed"] / z["losses"] - 1.0)))
    print("\nlosses", np.round(losses, 6).tolist(), "\nreference", np.round(z["losses"], 6).tolist(), f"\nreference mixed-vs-fp32 max relative loss deviation {mixed_dev:.2e}",
          "\nfinal weights off / moved:", {k: round(v, 4) for k, v in off.items()})
    assert len(losses) == len(z["losses"]) and np.allclose(losses, z["losses"], rtol=max(3e-3, 3 * mixed_dev)), (losses, z["losses"])
    assert max(off.values()) < 0.10, off
