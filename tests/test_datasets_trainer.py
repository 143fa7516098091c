"""CPU tests of the dataset side of the path and of the trainer features around the hot path (8f-3), pinned to
goldens produced by the REAL reference classes (tests/golden/make_dataset_golden.py):

  clip-aware window indexing, mixed-dataset pre-training windows, the PretrainingTask and fine-tuning trainers (loss trajectories,
  parameter groups, per-epoch learning rates), optimiser state in checkpoints + resume, the SIGUSR1 pre-emption hook.

The product trainer runs on the CPU here with the device math swapped for the pinned oracle (helpers.register_oracle_math_model);
tests/test_gpu_golden.py replays the same trainer goldens with the HIP model."""
import json
import os
import signal

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_case, register_oracle_math_model, write_hf_dir

PARTS = {"ECG": 1, "ventilator": 5, "bidmc": 3, "ludb": 2}     # component datasets of the mix -> channels


@pytest.fixture(scope="module")
def G():
    z = np.load(GOLDEN / "datasets.npz")
    return {k: z[k] for k in z.files}, json.loads((GOLDEN / "datasets.json").read_text())


def base_cfg(task, dataset, llm_dir="unused", step=16, epochs=1, model="medtsllm", extra=None):
    from med_ts_llm_amd.utils import dict_to_object
    prompting = {"dataset": True, "task": True, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
    d = {
        "DEBUG": True, "task": task, "model": model, "history_len": 32, "pred_len": 32,
        "data": {"dataset": dataset, "mode": "multivariate", "cols": "all", "normalize": True, "step": step},
        "training": {"epochs": epochs, "batch_size": 4, "optimizer": "adam", "learning_rate": 1e-3, "dropout": 0.0, "loss": "mse",
                     "eval_metric": "mse", "eval_metric_direction": "min", "shuffle": False},
        "tasks": {"segmentation": {"mode": "boundary-prediction"}, "pretraining": {"downsample_pct": 0.5, "n_features": 3}},
        "models": {"timellm": {"d_model": 8, "d_ff": 64, "n_heads": 2, "num_tokens": 64, "covariate_mode": "concat",
                               "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8}, "prompting": prompting,
                               "llm": {"enabled": True, "llm": llm_dir, "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}},
        "setup": {"seed": 0, "device": "cpu", "dtype": "fp32", "num_workers": 0, "logger": "print", "quiet": True},
    }
    d.update(extra or {})
    return dict_to_object(d)


def register_clip_source(G):
    from med_ts_llm_amd.tasks.windows import register_series
    data, meta = G

    def source(config, split):
        return {"data": data[f"clip.raw.{split}"], "labels": data[f"clip.labels.{split}"], "clip_ids": data[f"clip.ids.{split}"],
                "clip_descriptions": {int(k): v for k, v in meta[f"clip.desc.{split}"].items()},
                "description": "synthetic clips of physiological waveforms."}
    register_series("synthetic_clips", source)


def register_part_sources(G):
    from med_ts_llm_amd.tasks.windows import register_series
    data, _ = G
    for name in PARTS:
        register_series(name, lambda config, split, name=name: {"data": data[f"part.{name}.{split}"],
                                                                  "description": f"synthetic stand-in for the {name} dataset."})


# ------------------------------------------------------------------------------------------------ clip-aware windows (R:datasets/base.py:284-335)
@pytest.mark.parametrize("task", ["reconstruction", "semantic_segmentation"])
@pytest.mark.parametrize("step", [8, 40])
@pytest.mark.parametrize("split", ["val", "test"])
def test_clip_window_index_matches_reference(G, task, step, split):
    from med_ts_llm_amd.tasks.windows import make_series_dataset
    data, meta = G
    register_clip_source(G)
    ds = make_series_dataset(base_cfg(task, "synthetic_clips", step=step), split)
    k = f"clip.{task}.s{step}.{split}."
    assert ds.clip_dataset and len(ds) == int(data[k + "len"])
    assert np.array_equal(np.array([ds.inverse_index(i) for i in range(len(ds))]), data[k + "ranges"])
    assert np.array_equal(ds.mask.numpy(), data[k + "mask"]) and len(ds.mask) == ds.n_points
    items = [ds[i] for i in range(len(ds))]
    assert np.allclose([it["x_enc"][0, 0].item() for it in items], data[k + "x0"], rtol=0, atol=1e-6)       # train-split normalisation too
    assert [it["descriptions"] for it in items] == meta[k + "descriptions"]
    if task == "semantic_segmentation":
        assert ds.n_classes == int(data[k + "n_classes"]) and all(it["labels"].shape == (32,) for it in items)
    with pytest.raises(AssertionError):        # R:datasets/base.py:290: clip datasets do not support forecasting windows ...
        from med_ts_llm_amd.tasks.windows import ForecastSeries, _with_clips
        _with_clips(ForecastSeries)(base_cfg("forecasting", "synthetic_clips", step=step), split)


@pytest.mark.parametrize("step", [8, 40])
def test_clip_stitched_predict_matches_reference_task(G, step):
    """the product ReconstructionTask.predict (device-side stitching, tasks/evalpath.py) over a clip dataset == the reference task's"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer
    data, _ = G
    register_clip_source(G)

    class FakeRecon(torch.nn.Module):           # the deterministic window -> output function the reference run used
        supported_tasks = ["reconstruction"]

        def __init__(self, config, dataset):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))

        def forward(self, inputs):
            x = inputs["x_enc"]
            return 0.9 * x + 0.05 * x.roll(1, dims=1) + 0.01 * x[:, :1, :]

    model_lookup["fake_recon"] = FakeRecon
    try:
        tr = get_trainer("DEBUG-clip", base_cfg("reconstruction", "synthetic_clips", step=step, model="fake_recon"))
        for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
            p, t = tr.predict(dl)
            assert np.array_equal(t.numpy(), data[f"clip.predict.s{step}.{split}.targets"])
            assert np.allclose(p.numpy(), data[f"clip.predict.s{step}.{split}.preds"], rtol=0, atol=1e-6)
    finally:
        del model_lookup["fake_recon"]


def test_univariate_view_indexing():
    """R:datasets/util.py:10-43 (the reference's own class cannot index its datasets — see make_dataset_golden.py — so this is a
    property test): sample i = (window i // C, feature i % C), inverse_index returns (time range, feature)"""
    from med_ts_llm_amd.tasks.windows import register_series, make_series_dataset
    rng = np.random.default_rng(0)
    raw = rng.standard_normal((100, 3)).astype(np.float32)
    register_series("uni_src", lambda config, split: {"data": raw})
    cfg = base_cfg("reconstruction", "uni_src", step=8)
    multi = make_series_dataset(cfg, "val")
    cfg.data.mode = "univariate"
    uni = make_series_dataset(cfg, "val")
    assert uni.univariate and uni.n_features == 1 and uni.real_features == 3 and len(uni) == 3 * len(multi)
    for i in range(len(uni)):
        (t0, t1), f = uni.inverse_index(i)
        assert (t0, t1) == multi.inverse_index(i // 3) and f == i % 3
        assert torch.equal(uni[i]["x_enc"], multi[i // 3]["x_enc"][:, f:f + 1])


# ------------------------------------------------------------------------------------------------ mixed windows (R:datasets/util.py:46-118)
@pytest.mark.parametrize("nf", [3, "auto"])
def test_mixed_windows_match_reference_pretraining_dataset(G, nf):
    from med_ts_llm_amd.tasks.windows import MixedWindows, make_series_dataset
    data, meta = G
    register_part_sources(G)
    parts = {name: make_series_dataset(base_cfg("reconstruction", name), "train") for name in PARTS}
    torch.manual_seed(77)
    mix = MixedWindows(parts, downsample_pct=0.5, n_features=nf)
    k = f"mix.nf{nf}."
    assert len(mix) == int(data[k + "len"]) and mix.n_features == int(data[k + "n_features"]) and mix.n_points == int(data[k + "n_points"])
    assert mix.lens == data[k + "lens"].tolist() and mix.cumsums == data[k + "cumsums"].tolist()
    for j, inds in enumerate(mix.dataset_inds):
        assert np.array_equal(inds.numpy(), data[k + f"inds{j}"])
    items = [mix[i] for i in range(len(mix))]
    assert np.allclose(np.stack([it["x_enc"].numpy() for it in items]), data[k + "x"], rtol=0, atol=1e-6)
    assert [it["dataset"] for it in items] == meta[k + "names"]
    assert sorted({it["dataset_description"] for it in items}) == meta[k + "descriptions"]
    assert np.array_equal(np.array([[mix.inverse_index_full(i)[0], *mix.inverse_index_full(i)[1]] for i in range(len(mix))]), data[k + "full_index"])
    assert np.array_equal(np.array([mix.inverse_index(i) for i in range(len(mix))]), data[k + "index"])
    assert mix.description == meta["mix.description"] and mix.task == "pretraining" and not mix.clip_dataset and not mix.univariate


# ------------------------------------------------------------------------------------------------ trainers
def _golden_backbone_dir(tmp_path):
    _, _, bcfg, backbone = load_case("gpt2_concat_fc")
    return write_hf_dir(tmp_path / "llm_gpt2", bcfg, backbone)


def _load_init(trainer, z, prefix="init."):
    sd = {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    missing, unexpected = trainer.model.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"word_embeddings"}, (missing, unexpected)


def _check_final(trainer, z, tol_each=0.15, tol_all=0.05):
    """every weight ends within `tol_each` of the distance it moved from the reference's final value, all of them together within `tol_all`.
    (The reference's `tasks` package sets torch.set_float32_matmul_precision("medium"), so the golden gradients carry ~3e-4 of
    reduced-precision matmul noise, which Adam's normalisation turns into full +-lr steps on elements whose gradient is near zero — up to
    ~10 % on a bias vector over 4-6 steps; a wrong step order, a stale gradient or a wrong parameter group moves these by O(1).)"""
    p = dict(trainer.model.named_parameters())
    err2 = moved2 = 0.0
    for k in z.files:
        if k.startswith("final."):
            n = k[len("final."):]
            if n.endswith("key_projection.bias"):
                continue       # analytically-zero gradient: Adam turns round-off into +-lr steps
            moved = float(np.linalg.norm(z[k] - z["init." + n]))
            err = float((p[n].detach().cpu() - torch.from_numpy(z[k])).norm())
            assert err < tol_each * moved + 1e-6, (k, err / (moved + 1e-30))
            err2, moved2 = err2 + err ** 2, moved2 + moved ** 2
    assert err2 ** 0.5 < tol_all * moved2 ** 0.5, (err2 ** 0.5 / moved2 ** 0.5)


def pretraining_config(tmp_path, device="cpu", dtype="fp32", model="medtsllm"):
    return base_cfg("pretraining", "pretrain-mix", llm_dir=_golden_backbone_dir(tmp_path), model=model,
                    extra={"setup": {"seed": 0, "device": device, "dtype": dtype, "num_workers": 0, "logger": "print", "quiet": True}})


def test_pretraining_task_replays_reference_trajectory(G, tmp_path):
    """R:tasks/pretraining.py through the product trainer: the mix the reference drew (same RNG draws in the same order), its batches,
    its per-step losses and final weights"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer, task_lookup
    from med_ts_llm_amd.tasks.tasks import ReconstructionTask
    z = np.load(GOLDEN / "trainer_pretraining.npz")
    register_part_sources(G)
    assert issubclass(task_lookup["pretraining"], ReconstructionTask)         # R:tasks/pretraining.py:5 — target = x_enc
    key = register_oracle_math_model()
    try:
        tr = get_trainer("DEBUG-pretrain", pretraining_config(tmp_path, model=key))
        assert tr.task == "pretraining" and tr.model.task == "pretraining" and tr.train_dataset.lens == z["mix.lens"].tolist()
        for j, inds in enumerate(tr.train_dataset.dataset_inds):
            assert np.array_equal(inds.numpy(), z[f"mix.inds{j}"])
        for i, b in enumerate(tr.train_dataloader):
            assert np.allclose(b["x_enc"].numpy(), z[f"batch{i}.x_enc"], rtol=0, atol=1e-6) and len(b["dataset"]) == b["x_enc"].shape[0]
        _load_init(tr, z)
        tr.train()
    finally:
        del model_lookup[key]
    losses = [h["train/loss"] for h in tr.logger.history if "train/loss" in h]
    assert np.allclose(losses, z["losses"], rtol=1e-3, atol=1e-6), (losses, z["losses"])
    _check_final(tr, z)
    assert tr.step == int(z["step_counter"])


def finetune_config(tmp_path, device="cpu", dtype="fp32", model="medtsllm", debug=True):
    return base_cfg("reconstruction", "bidmc", llm_dir=_golden_backbone_dir(tmp_path), epochs=3, model=model,
                    extra={"DEBUG": debug, "paths": {"logdir": str(tmp_path / "logs")},
                           "setup": {"seed": 0, "device": device, "dtype": dtype, "num_workers": 0, "logger": "print", "quiet": True},
                           "finetuning": {"enabled": True, "pretrained_id": "pretrain-golden", "pretrained_ckpt": "latest", "frozen_epochs": 0,
                                          "warmup_epochs": 2, "warmup_factor": 0.1}})


def write_pretrained_checkpoint(tmp_path, z):
    d = tmp_path / "logs" / "pretrain-golden" / "checkpoints"
    d.mkdir(parents=True, exist_ok=True)
    torch.save({"run_id": "pretrain-golden", "epoch": 1, "step": 16,
                "model": {k[len("pretrained."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("pretrained.")}}, d / "latest.pt")


def test_finetuning_from_pretrained_replays_reference(G, tmp_path):
    """R:tasks/base.py:88-91,118-155: the reference's pre-training checkpoint is loaded minus the output head, the loaded parameters
    form their own optimiser group with a warm-up schedule; groups, per-epoch learning rates, losses and final weights == the reference's"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer
    z = np.load(GOLDEN / "trainer_finetune.npz")
    meta = json.loads((GOLDEN / "trainer_finetune.json").read_text())
    register_part_sources(G)
    write_pretrained_checkpoint(tmp_path, z)
    key = register_oracle_math_model("timellm_oracle_math")
    try:
        tr = get_trainer("DEBUG-finetune", finetune_config(tmp_path, model=key))
    finally:
        del model_lookup[key]
    assert tr.finetuning and tr.loaded_params == meta["loaded_params"]
    named = {id(p): n for n, p in tr.model.named_parameters()}
    assert [[named[id(p)] for p in g["params"]] for g in tr.optimizer.param_groups] == meta["groups"]
    # the pre-trained weights arrived (everything but the output head, which keeps its fresh init)
    for k in z.files:
        if k.startswith("pretrained.") and not k.startswith("pretrained.output_projection"):
            assert torch.equal(dict(tr.model.named_parameters())[k[len("pretrained."):]].detach(), torch.from_numpy(z[k])), k
    _load_init(tr, z)                          # (the head's random init is the reference run's)
    lrs = []
    orig = tr.log_epoch
    tr.log_epoch = lambda scores={}, **kw: (lrs.append(list(tr.scheduler.get_last_lr())), orig(scores, **kw))[1]
    tr.train()
    assert np.allclose(lrs, z["lrs"], rtol=1e-6), (lrs, z["lrs"])
    assert any("train/finetune_lr" in h for h in tr.logger.history)
    losses = [h["train/loss"] for h in tr.logger.history if "train/loss" in h]
    assert np.allclose(losses, z["losses"], rtol=1e-3, atol=1e-6), (losses, z["losses"])
    _check_final(tr, z)


def test_frozen_epochs_schedule_and_exclusivity(G, tmp_path):
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer
    z = np.load(GOLDEN / "trainer_finetune.npz")
    register_part_sources(G)
    write_pretrained_checkpoint(tmp_path, z)
    key = register_oracle_math_model("timellm_oracle_math")
    try:
        cfg = finetune_config(tmp_path, model=key)
        cfg.finetuning.frozen_epochs, cfg.finetuning.warmup_epochs = 1, 0
        tr = get_trainer("DEBUG-frozen", cfg)
        before = {n: p.detach().clone() for n, p in tr.model.named_parameters() if n in tr.loaded_params}
        head0 = tr.model.output_projection.linear.weight.detach().clone()
        tr.model.train()
        for batch in tr.train_dataloader:      # epoch 0: the pre-trained group is frozen (lr factor 0), the new head trains
            tr.train_step(batch)
        assert tr.scheduler.get_last_lr() == [1e-3, 0.0]
        assert all(torch.equal(dict(tr.model.named_parameters())[n].detach(), v) for n, v in before.items())
        assert not torch.equal(tr.model.output_projection.linear.weight.detach(), head0)
        tr.scheduler.step()
        assert tr.scheduler.get_last_lr() == [1e-3, 1e-3]
        cfg.finetuning.warmup_epochs = 2
        with pytest.raises(AssertionError):
            get_trainer("DEBUG-both", cfg)
    finally:
        del model_lookup[key]


def test_checkpoint_carries_optimizer_state_and_resume_continues_exactly(G, tmp_path):
    """8f-3: `latest.pt` = the reference's fields + the optimiser state; from_run_id rebuilds the trainer from the run directory (config.json
    + checkpoint) and the resumed run continues exactly like an uninterrupted one (the reference restarts Adam's moments at zero)"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer, task_lookup
    register_part_sources(G)
    key = register_oracle_math_model()
    try:
        def cfg(epochs):
            c = base_cfg("reconstruction", "bidmc", llm_dir=_golden_backbone_dir(tmp_path) if not (tmp_path / "llm_gpt2").exists() else str(tmp_path / "llm_gpt2"),
                         epochs=epochs, model=key, extra={"DEBUG": False, "paths": {"logdir": str(tmp_path / "logs")}})
            return c
        torch.manual_seed(0)
        full = get_trainer("run-full", cfg(2))
        init = {k: v.detach().clone() for k, v in full.model.state_dict().items()}
        full.train()
        part = get_trainer("run-part", cfg(1))
        part.model.load_state_dict(init, strict=False)
        part.train()
        ck = torch.load(tmp_path / "logs" / "run-part" / "checkpoints" / "latest.pt")
        assert set(ck) == {"run_id", "epoch", "step", "datetime", "model", "optimizer", "epochs_done"} and ck["epochs_done"] == 1
        assert set(ck["optimizer"]["state"]) == {n for n, p in part.model.named_parameters() if p.requires_grad}
        assert all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in ck["optimizer"]["state"].values())
        assert json.loads((tmp_path / "logs" / "run-part" / "config.json").read_text())["training"]["epochs"] == 1
        resumed = task_lookup["reconstruction"].from_run_id("run-part", cfg={"training": cfg(2).training.to_dict()}, basepath=str(tmp_path / "logs"))
        assert resumed.step == part.step and resumed.config.training.epochs == 2 and resumed.epochs_done == 1
        resumed.train()                           # continues with epoch 2 of 2 (the reference would start over at epoch 1)
        assert resumed.epochs_done == 2
        for (n, a), (_, b) in zip(full.model.named_parameters(), resumed.model.named_parameters()):
            if a.requires_grad:
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-8), n
    finally:
        del model_lookup[key]


def test_sigusr1_checkpoints_and_exits(G, tmp_path):
    """R:tasks/base.py:55,277-281: a pre-emption notice saves "latest", closes the logger and exits with status 0"""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer
    register_part_sources(G)
    key = register_oracle_math_model()
    old = signal.getsignal(signal.SIGUSR1)
    try:
        cfg = base_cfg("reconstruction", "bidmc", llm_dir=_golden_backbone_dir(tmp_path), model=key,
                       extra={"DEBUG": False, "paths": {"logdir": str(tmp_path / "logs")}})
        tr = get_trainer("run-preempted", cfg)
        assert signal.getsignal(signal.SIGUSR1) == tr.handle_termination
        with pytest.raises(SystemExit) as e:
            os.kill(os.getpid(), signal.SIGUSR1)
            for _ in range(100):                 # the handler runs between bytecodes
                pass
        assert e.value.code == 0
        assert (tmp_path / "logs" / "run-preempted" / "checkpoints" / "latest.pt").exists()
    finally:
        signal.signal(signal.SIGUSR1, old)
        del model_lookup[key]
