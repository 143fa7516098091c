"""CPU tests of the host-side logic (no GPU): prompt construction, construction/checkpoint surface, C-ABI symbols,
and the optimisation-step order (a10) pinned to the reference trainer's loss trajectory."""
import json
import os
import re
import sys
from pathlib import Path

import numpy as np

import pytest
import torch

from helpers import CASES, GOLDEN, load_case, oracle_mcfg, rel_err, fixture_tokenizer

ROOT = Path(__file__).resolve().parent.parent


def _cfg_from_meta(meta):
    from med_ts_llm_amd.utils import dict_to_object
    return dict_to_object({
        "DEBUG": True, "task": meta["task"], "model": "medtsllm", "history_len": meta["L"], "pred_len": meta["pred_len"],
        "training": {"dropout": 0.0}, "setup": {"dtype": "fp32"}, "tasks": {"segmentation": {"mode": "boundary-prediction"}},
        "models": {"timellm": {
            "d_model": meta["d_model"], "d_ff": meta["d_ff"], "n_heads": meta["n_heads"], "num_tokens": meta["num_tokens"],
            "covariate_mode": meta["covariate_mode"], "embedding_downsample_mode": meta["embedding_downsample_mode"],
            "patching": {"patch_len": meta["patch_len"], "stride": meta["stride"]}, "prompting": meta["prompting"],
            "llm": {"enabled": True, "llm": "fixture", "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}}})


class _DS:
    def __init__(self, meta):
        self.description, self.n_features, self.n_classes, self.task_description = meta["dataset_description"], meta["C"], meta["n_classes"], None


def _model(name):
    from med_ts_llm_amd.models import model_lookup
    meta, data, bcfg, backbone = load_case(name)
    m = model_lookup["medtsllm"](_cfg_from_meta(meta), _DS(meta), backbone_state=(bcfg, backbone))
    m.tokenizer = fixture_tokenizer(meta["backbone"])
    return m, meta, data


@pytest.mark.parametrize("name", CASES)
def test_prompt_strings_and_token_ids_exact(name):
    """a6: identical part list, order, trailing-space rule, stats formatting; per-part token ids (reference goldens)."""
    model, meta, data = _model(name)
    inputs = {"x_enc": torch.from_numpy(data["x_enc"])}
    if meta["descriptions"]:
        inputs["descriptions"] = meta["descriptions"]
    if "examples" in data:
        inputs["examples"] = [("Example segment:", torch.from_numpy(data["examples"][b:b + 1])) for b in range(meta["B"])]
    parts = model.build_prompt(inputs)
    if "examples" in data:      # the tensor part sits where the reference puts it
        for b, ps in enumerate(parts):
            assert sum(torch.is_tensor(p) for p in ps) == 1 and torch.equal(next(p for p in ps if torch.is_tensor(p)), inputs["examples"][b][1])
    parts = [[p if isinstance(p, str) else "<TENSOR>" for p in ps] for ps in parts]
    assert parts == meta["prompts"]
    ids = [[model.tokenizer(p, padding=False, truncation=False).input_ids if p != "<TENSOR>" else None for p in ps] for ps in parts]
    assert ids == meta["prompt_token_ids"]
    if "examples" in data:
        return
    assert model.task_description == meta["task_description"]
    if parts[0]:
        from med_ts_llm_amd.models.prompt import left_pad_ids
        rows = left_pad_ids(ids, model.tokenizer.pad_token_id)
        n = max(sum(len(p) for p in ps) for ps in ids)
        assert all(len(r) == n for r in rows)
        for r, ps in zip(rows, ids):
            flat = [i for p in ps for i in p]
            assert r[n - len(flat):] == flat and all(v == meta["pad_token_id"] for v in r[: n - len(flat)])


def test_calc_lags_exact():
    from med_ts_llm_amd.models.prompt import calc_lags
    z = np.load(GOLDEN / "stats.npz")
    x = torch.from_numpy(z["x"])
    assert np.array_equal(calc_lags(x, 5).numpy(), z["lags_3d"])
    assert np.array_equal(calc_lags(x[:, :, 1], 5).numpy(), z["lags_2d"])


@pytest.mark.parametrize("name", CASES)
def test_construction_and_checkpoint_surface(name):
    """a11: derived sizes, parameter names/shapes/requires_grad, state_dict filtering, load_pretrained key dropping."""
    model, meta, _ = _model(name)
    table = {n: {"shape": list(p.shape), "requires_grad": bool(p.requires_grad)} for n, p in model.named_parameters()}
    assert table == meta["param_table"]
    assert list(model.state_dict().keys()) == meta["state_dict_keys"]
    assert model.n_patches == meta["n_patches"] and model.n_outputs == meta["n_outputs"] and model.d_model == meta["d_model_eff"]
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert model.load_pretrained(sd) == meta["load_pretrained_keys"]
    with pytest.raises(RuntimeError):      # the product path has no CPU fallback
        model({"x_enc": torch.zeros(2, meta["L"], meta["C"])})


def test_registry_and_config_object():
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import task_lookup, get_trainer  # noqa: F401
    from med_ts_llm_amd.utils import dict_to_object
    assert model_lookup["timellm"] is model_lookup["medtsllm"]
    assert set(task_lookup) == {"forecasting", "anomaly_detection", "reconstruction", "segmentation", "semantic_segmentation", "pretraining"}
    c = dict_to_object({"a": {"b": 1}, "c": 2})
    assert c.a.b == 1 and c["c"] == 2 and "a" in c and c.get("zz", 7) == 7 and c.copy().to_dict() == {"a": {"b": 1}, "c": 2}


def test_c_abi_exports_every_declared_symbol():
    from med_ts_llm_amd.hip import _native
    header = (ROOT / "include" / "medtsllm_hip.h").read_text()
    declared = set(re.findall(r"\b(mtl_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = _native.lib()                    # loads without a GPU; getattr raises on a missing export
    for name in declared:
        getattr(lib, name)
    assert lib.mtl_abi_version() == _native.ABI_VERSION
    assert b"alignment" in lib.mtl_strerror(-2)


def test_gemm_tile_order_visits_every_tile_once():
    """the persistent GEMM's work distribution (XCD chunks, tile groups, per-XCD column rotation: csrc/mtl_gemm.hip) restated on
    the host for the orders mtl_gemm_tile_order() picks: every tile of every grid must be owned by exactly one workgroup"""
    from med_ts_llm_amd.hip import _native
    lib = _native.lib()
    n_cu = 256
    seen_rot = 0
    for bm, bn in ((128, 64), (128, 96), (128, 192), (256, 96), (256, 192), (256, 256)):
        for per_cu in (1, 2, 3):
            for tiles_m in list(range(1, 20)) + [24, 32, 40, 64, 72, 128]:
                for tiles_n in (1, 2, 3, 5, 8, 9, 12, 16, 24, 43):
                    order = lib.mtl_gemm_tile_order(tiles_m, tiles_n, bm, bn, per_cu, 768, 0)
                    assert order > 0
                    g, col_rot = order & 0xff, bool(order & (1 << 9))
                    seen_rot += col_rot
                    nt = tiles_m * tiles_n
                    nblk = min(nt, per_cu * n_cu)
                    owned = []
                    for bid in range(nblk):
                        xcd, slot = bid & 7, bid >> 3
                        xblocks = (nblk - xcd + 7) >> 3
                        q, r8 = nt >> 3, nt & 7
                        t0 = xcd * (q + 1) if xcd < r8 else r8 * (q + 1) + (xcd - r8) * q
                        cnt = q + (1 if xcd < r8 else 0)
                        t = t0 + slot
                        while t < t0 + cnt:
                            grp, rem = divmod(t, g * tiles_n)
                            first_m = grp * g
                            gm = min(tiles_m - first_m, g)
                            tm, tn = first_m + rem % gm, rem // gm
                            if col_rot:
                                tn = (tn + ((xcd * tiles_n) >> 3)) % tiles_n
                            owned.append(tm * tiles_n + tn)
                            t += xblocks
                    assert sorted(owned) == list(range(nt)), (bm, bn, per_cu, tiles_m, tiles_n, order)
    assert seen_rot > 50            # the rotated orders were exercised
    assert lib.mtl_gemm_tile_order(32, 8, 128, 96, 1, 3072, 1) & (1 << 8) and not lib.mtl_gemm_tile_order(32, 8, 128, 96, 1, 8192, 1) & (1 << 8)


def golden_trainer_setup(tmp_path, device, dtype, model_key="medtsllm"):
    """Everything the product trainer needs to replay the reference trainer's golden run (tests/golden/make_golden.py
    run_trainer_golden): an on-disk HF-format backbone directory (golden weights + fixture tokenizer), a registered dataset
    that yields the golden batches in the golden order, and the reference run's config. -> (config, z)"""
    from torch.utils.data import Dataset
    from helpers import write_hf_dir
    from med_ts_llm_amd.tasks.synthetic import register_dataset
    from med_ts_llm_amd.utils import dict_to_object
    z = np.load(GOLDEN / "trainer_gpt2_concat_fc.npz")
    meta, _, bcfg, backbone = load_case("gpt2_concat_fc")
    d = write_hf_dir(Path(tmp_path) / "llm_gpt2", bcfg, backbone)
    n_batches = len([k for k in z.files if k.endswith(".x_enc")])
    xs = torch.cat([torch.from_numpy(z[f"batch{i}.x_enc"]) for i in range(n_batches)])
    ys = torch.cat([torch.from_numpy(z[f"batch{i}.y"]) for i in range(n_batches)])

    class GoldenWindows(Dataset):
        description, n_features, n_classes, task_description = meta["dataset_description"], meta["C"], 0, None

        def __len__(self):
            return xs.shape[0]

        def __getitem__(self, i):
            return {"x_enc": xs[i], "y": ys[i]}

    register_dataset("golden_trainer", lambda config, split: GoldenWindows())
    cfg = dict_to_object({
        "DEBUG": True, "task": "forecasting", "model": model_key, "history_len": meta["L"], "pred_len": meta["pred_len"],
        "data": {"dataset": "golden_trainer"},
        "training": {"epochs": 2, "batch_size": 4, "optimizer": "adam", "learning_rate": 1e-3, "dropout": 0.0, "loss": "mse",
                     "eval_metric": "loss", "eval_metric_direction": "min", "shuffle": False},
        "tasks": {"segmentation": {"mode": "boundary-prediction"}},
        "models": {"timellm": {
            "d_model": meta["d_model"], "d_ff": meta["d_ff"], "n_heads": meta["n_heads"], "num_tokens": meta["num_tokens"],
            "covariate_mode": "concat", "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8},
            "prompting": meta["prompting"],
            "llm": {"enabled": True, "llm": str(d), "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}},
        "setup": {"seed": 0, "device": device, "dtype": dtype, "num_workers": 0, "logger": "print", "quiet": True}})
    return cfg, z, n_batches


def load_golden_init(trainer, z):
    """the reference trainer's initial trainable weights -> the product model (in place: optimiser / bf16 shadows keep their references)"""
    sd = {k[len("init."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init.")}
    missing, unexpected = trainer.model.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"word_embeddings"}, (missing, unexpected)


def test_optimisation_step_order_matches_reference_trajectory(tmp_path):
    """a10: the PRODUCT trainer — tasks.get_trainer(...).train(), i.e. BaseTask.train_step / prepare_batch / build_optimizer /
    log_step — replays the REFERENCE trainer's golden run (8 Adam steps over 2 epochs) and must land on the reference's per-step
    losses and final weights. The device math is swapped for the pinned oracle (the HIP model has no CPU path) by a model class
    that keeps the product's constructor, parameter names and prompt builder; tests/test_gpu_golden.py runs the same replay with
    the HIP model itself."""
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.tasks import get_trainer
    from helpers import register_oracle_math_model
    key = register_oracle_math_model()
    try:
        cfg, z, n_batches = golden_trainer_setup(tmp_path, "cpu", "fp32", key)
        trainer = get_trainer("DEBUG-golden", cfg)
        load_golden_init(trainer, z)
        trainer.train()
    finally:
        del model_lookup[key]
    losses = [h["train/loss"] for h in trainer.logger.history if "train/loss" in h]
    # importing the reference's tasks package sets torch.set_float32_matmul_precision("medium") (R:tasks/base.py:19-22),
    # so the golden trajectory itself carries ~3e-4 of reduced-precision CPU matmul noise; a wrong step order
    # (e.g. zero_grad before step, stale gradients) moves these numbers by O(1e-1).
    assert len(losses) == 2 * n_batches == len(z["losses"])
    assert np.allclose(losses, z["losses"], rtol=1e-3, atol=1e-6), (losses, z["losses"])
    p = dict(trainer.model.named_parameters())
    for k in z.files:
        if k.startswith("final."):
            name = k[len("final."):]
            if name.endswith("key_projection.bias"):
                continue   # analytically-zero gradient: Adam normalises pure round-off noise into +-lr steps (not reproducible)
            moved = float(np.linalg.norm(z[k] - z["init." + name]))
            assert float((p[name].detach() - torch.from_numpy(z[k])).norm()) < 0.05 * moved + 1e-6, k
    assert trainer.step == int(z["step_counter"]) == 4 * 2 * n_batches   # BaseTask.step advances by batch_size per step (R:tasks/base.py:217)


def test_backbone_config_options_that_change_numerics_are_rejected():
    """a checkpoint whose config asks for arithmetic the kernels do not implement must not load silently (HF would apply it)"""
    from med_ts_llm_amd.models.backbone import normalise_config
    llama = {"model_type": "llama", "vocab_size": 512, "hidden_size": 128, "intermediate_size": 192, "num_hidden_layers": 2,
             "num_attention_heads": 2, "num_key_value_heads": 2, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
    gpt2 = {"model_type": "gpt2", "vocab_size": 512, "n_positions": 256, "n_embd": 128, "n_layer": 2, "n_head": 2}
    assert normalise_config(llama)["rope_theta"] == 10000.0 and normalise_config(gpt2)["ffn"] == 512
    assert normalise_config({**llama, "rope_scaling": None, "attention_bias": False, "mlp_bias": False, "hidden_act": "silu"})["arch"] == "llama"
    for bad in ({"rope_scaling": {"rope_type": "llama3", "factor": 8.0}}, {"rope_scaling": {"type": "linear", "factor": 2.0}},
                {"rope_parameters": {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0}},
                {"attention_bias": True}, {"mlp_bias": True}, {"hidden_act": "gelu"}, {"attention_dropout": 0.1}):
        with pytest.raises(NotImplementedError):
            normalise_config({**llama, **bad})
    for bad in ({"activation_function": "gelu"}, {"scale_attn_by_inverse_layer_idx": True}, {"reorder_and_upcast_attn": True},
                {"scale_attn_weights": False}):
        with pytest.raises(NotImplementedError):
            normalise_config({**gpt2, **bad})


def test_prompt_token_ids_are_range_checked_on_the_host():
    """the assembly kernel gathers embedding rows by id and cannot raise: out-of-range ids (a '[PAD]' token beyond an un-resized
    table, bad fixed_prompt_ids) are refused before the launch, like nn.Embedding's IndexError in the reference"""
    model, meta, data = _model("gpt2_concat_fc")
    model._check_ids([[0, 5, 511]])
    for rows in ([[0, 512]], [[-1, 3]]):
        with pytest.raises(IndexError):
            model._check_ids(rows)
    model.fixed_prompt_ids = torch.tensor([[1, 2, 9999]], dtype=torch.int32)
    with pytest.raises(IndexError):
        model._prompt_ids({"x_enc": torch.zeros(1, 64, 3)}, torch.device("cpu"))


def test_dropout_mask_function_statistics():
    """the counter-based dropout mask (csrc/mtl_common.h::drop_quad, replicated in helpers.drop_u16): keep rate at the nominal 1 - p,
    no correlation between neighbouring elements / rows / streams, uniform 16-bit fields — on the index patterns the kernels use
    (a [rows, cols] matrix; [B*H, Tq, Tk] attention probabilities)"""
    import numpy as np
    from helpers import drop_u16, drop_threshold
    thr = drop_threshold(0.1)
    for seed in (1, 777, 123456789):
        r, c = np.meshgrid(np.arange(2048, dtype=np.uint64), np.arange(768, dtype=np.uint64), indexing="ij")
        keep = (drop_u16(seed, 0, r, c) >= thr).astype(np.float64)
        assert abs(keep.mean() - 0.9) < 1.5e-3
        z = keep - keep.mean()
        for sh in (1, 2, 3, 4, 8):
            assert abs((z[:, :-sh] * z[:, sh:]).mean() / z.var()) < 4e-3, ("col", sh)
            assert abs((z[:-sh, :] * z[sh:, :]).mean() / z.var()) < 4e-3, ("row", sh)
        # row / column keep rates scatter like independent Bernoulli draws
        assert 0.9 < keep.mean(1).std() / np.sqrt(0.09 / 768) < 1.1 and 0.9 < keep.mean(0).std() / np.sqrt(0.09 / 2048) < 1.1
        bh, q, k = np.meshgrid(np.arange(24, dtype=np.uint64), np.arange(128, dtype=np.uint64), np.arange(256, dtype=np.uint64), indexing="ij")
        k3 = (drop_u16(seed, bh, q, k) >= thr).astype(np.float64)
        z3 = k3 - k3.mean()
        assert abs(k3.mean() - 0.9) < 1.5e-3
        assert abs((z3[:-1] * z3[1:]).mean() / z3.var()) < 4e-3            # the same (q, k) of neighbouring heads
    r, c = np.meshgrid(np.arange(1024, dtype=np.uint64), np.arange(1024, dtype=np.uint64), indexing="ij")
    v = drop_u16(99, 0, r, c).astype(np.int64)
    for byte in (v >> 8, v & 255):
        hist = np.bincount(byte.ravel(), minlength=256)
        e = byte.size / 256
        assert ((hist - e) ** 2 / e).sum() / 255 < 1.5                     # chi-square per degree of freedom


def test_pure_16bit_dtype_is_rejected_at_construction():
    """R:tasks/base.py:261-264 casts model and inputs to bf16 / fp16; the MI355X path keeps fp32 masters ("mixed"): the product
    trainer refuses such a config in its constructor instead of failing inside the first optimiser step"""
    from med_ts_llm_amd.tasks.base import BaseTask

    class Probe(BaseTask):
        def build_loss(self):
            return torch.nn.MSELoss()

    from med_ts_llm_amd.utils import dict_to_object
    for name in ("fp16", "half"):
        cfg = dict_to_object({"task": "forecasting", "setup": {"device": "cpu", "dtype": name, "seed": 0}})
        with pytest.raises(ValueError, match="mixed"):
            Probe("r", cfg)
    with pytest.raises(ValueError):          # bf16 on a CPU device: "Invalid dtype selection", as in the reference (R:tasks/base.py:269-270)
        Probe("r", dict_to_object({"task": "forecasting", "setup": {"device": "cpu", "dtype": "bf16", "seed": 0}}))


def test_derived_window_datasets_pickle():
    """DataLoader workers under spawn / forkserver pickle the dataset: clip / univariate classes are module-level, no closures kept"""
    import pickle
    import numpy as np
    from med_ts_llm_amd.tasks import windows
    from med_ts_llm_amd.utils import dict_to_object
    cfg = dict_to_object({"task": "reconstruction", "history_len": 8, "pred_len": 8,
                          "data": {"dataset": "pk", "step": 4, "normalize": False, "mode": "univariate"}})
    rng = np.random.default_rng(0)
    raw = {"data": rng.standard_normal((64, 3)).astype("float32"), "clip_ids": np.repeat(np.arange(4), 16)}
    ds = windows.make_series_dataset(cfg, "train", source=lambda c, sp: raw)
    assert type(ds).__name__ == "UnivariateClipReconstructionSeries" and getattr(windows, type(ds).__name__) is type(ds)
    back = pickle.loads(pickle.dumps(ds))
    assert len(back) == len(ds) and torch.equal(back[5]["x_enc"], ds[5]["x_enc"])
    # a spawn / forkserver worker imports the module AFRESH (make_series_dataset never ran there): the class must exist from import time on
    import subprocess
    import sys
    blob = tmp = None
    code = ("import pickle, sys, torch; sys.path.insert(0, %r); import med_ts_llm_amd; ds = pickle.load(open(sys.argv[1], 'rb')); "
            "print(type(ds).__name__, len(ds), float(ds[5]['x_enc'].sum()))" % str(ROOT))
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as f:
        pickle.dump(ds, f)
        tmp = f.name
    try:
        r = subprocess.run([sys.executable, "-c", code, tmp], capture_output=True, text=True, timeout=240)
    finally:
        os.unlink(tmp)
    assert r.returncode == 0, r.stderr[-800:]
    name, n, tot = r.stdout.split()
    assert name == "UnivariateClipReconstructionSeries" and int(n) == len(ds) and abs(float(tot) - float(ds[5]["x_enc"].sum())) < 1e-5


def test_bench_compact_line_is_small_and_complete():
    """the driver parses the LAST stdout line of bench.py: r03's 20 KB line (kernel instances, prose) could not be parsed. The compact form of a
    real full record (profiles/r03_v5c_bench.json: headline + two Llama configs) must stay under 4 KB and carry the contract's fields, the
    roofline objects as numbers and the CPU baseline."""
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    with open(ROOT / "profiles" / "r03_v5c_bench.json") as f:
        full = json.load(f)
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < 4096, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in c, k
    r = c["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and isinstance(r["traffic"], int)
    assert c["roofline_hbm"]["bound"] == "hbm" and c["roofline_hbm"]["unit"] == "GB/s"
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1
    assert "workload" in c["config"] and "model" not in c["config"]
    assert len(c["configs"]) == 2 and all("value" in e and "roofline" in e for e in c["configs"])
    # r03's Llama lines claimed a whole-step MFMA fraction (0.64) above their fastest GEMM's (0.55): the check flags exactly that record
    assert c["configs"][0]["checks"]["step_frac_le_best_kernel_frac"] is False
    assert c["checks"]["step_frac_le_best_kernel_frac"] is True


def test_bench_round5_fields_on_the_compact_line():
    """roofline_step (bytes of ALL dispatches of a step against 8 TB/s, FLOP per byte next to the ridge), the list of HBM-bound kernels and,
    under N > 1, both DP modes' figures ride on the ONE line and it stays under 4 KB"""
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    with open(ROOT / "profiles" / "r03_v5c_bench.json") as f:
        full = json.load(f)
    step = {"hbm_read_bytes": 10.4e9, "hbm_write_bytes": 5.6e9, "hbm_bytes": 16.0e9, "dispatches_per_step": 219.5}
    rs = bench.roofline_step(step, full, "live: two rocprofv3 passes")
    sec = full["ms_per_step"] * 1e-3
    assert rs["bound"] == "hbm" and rs["bytes_per_step"] == 16_000_000_000 and abs(rs["achieved"] - 16.0 / sec) < 0.1
    assert abs(rs["frac"] - rs["achieved"] / 8000.0) < 1e-3 and abs(rs["ridge_flop_per_byte"] - 312.5) < 1e-9
    assert abs(rs["flop_per_byte"] - full["executed_tflop_per_step_per_gpu"] * 1e12 / 16.0e9) < 0.1
    assert bench.roofline_step(None, full, "x") is None
    full["roofline_step"] = rs
    full["dp_modes"] = {"plain_allreduce": 40000.0, "sharded": 43000.0, "what": "both modes back to back"}
    c = json.loads(json.dumps(bench.compact_line(full), separators=(",", ":")))
    assert len(json.dumps(c, separators=(",", ":"))) < 4096
    assert c["roofline_step"]["bytes_per_step"] == 16_000_000_000 and "source" not in c["roofline_step"]
    assert c["dp_modes"]["plain_allreduce"] == 40000.0
    assert all(h["kernel"].startswith(("norm", "adam")) and 0 < h["frac"] < 1 for h in c["hbm_kernels"]) and len(c["hbm_kernels"]) >= 2
    assert all("step_mfma_frac_algorithmic" not in e for e in c["configs"])       # (with the prompt-row cache that figure is not a utilisation: detail file only)
    # round 6: every BASELINE.json config rides on the default line (five extra entries): still one line under 4 KB, each entry with its
    # value, ms per step, dominant GEMM and an (extrapolated-flagged) CPU figure
    full["configs"] = [dict(full["configs"][i % 2], cpu_baseline={"value": 0.0123, "cores": 16, "kind": "port", "extrapolated": True}) for i in range(5)]
    line = bench.fit_line(full)                    # (sheds per-config extras in a fixed order when the line would not fit)
    assert len(line) < 4000, len(line)
    c5 = json.loads(line)["configs"]
    assert len(c5) == 5 and all(e["cpu_baseline"]["extrapolated"] and e["roofline"]["frac"] and e["ms_per_step"] for e in c5)


def test_causal_fraction_of_the_rectangle():
    sys.path.insert(0, str(ROOT))
    import bench
    # T = 4 keys, the last 2 queries: they see 3 and 4 keys of 4 -> 7 / 8
    assert abs(bench.causal_fraction("attn_fwd_kernel<128, true, false, 8>", 4, 2) - 7 / 8) < 1e-12
    assert bench.causal_fraction("attn_fwd_kernel<128, false, true, 8>", 4, 2) == 1.0
    assert abs(bench.causal_fraction("attn_bwd_dq_res_kernel<64, 8, true>", 256, 256) - 257 / 512) < 1e-12
    # the 32-rows-per-wave long-sequence kernels are causal-only and carry no "true" in their launch labels
    assert abs(bench.causal_fraction("attn_bwd_dq_w32_kernel<128, 4>", 4, 2) - 7 / 8) < 1e-12
    assert abs(bench.causal_fraction("attn_fwd_w32_kernel<64, true, 4>", 4, 2) - 7 / 8) < 1e-12


def test_bench_pmc_traffic_arithmetic_and_fallback(tmp_path, monkeypatch):
    """bench.py's roofline.traffic: (i) the counter arithmetic of tools/pmc_traffic.py on a synthetic rocprofv3 counter table (FETCH_SIZE in KiB,
    doubled on the read side for gfx950; WRITE_SIZE in KiB; mean per dispatch, several counter rows of one dispatch summed), (ii) the live pass
    reports WHY it has nothing instead of raising when rocprofv3 cannot run the workload (no GPU here / tool absent) — bench.py then keeps the
    committed table of the same kernel sources."""
    import os
    import sqlite3
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    sys.path.insert(0, str(ROOT))
    import pmc_traffic as pt
    db = str(tmp_path / "r_results.db")
    c = sqlite3.connect(db)
    c.execute("create table counters_collection (kernel_name text, counter_name text, value real, dispatch_id integer)")
    kn = "void (anonymous namespace)::norm_fwd_kernel<3, false>(float const*, float*)"
    rows = [(kn, "FETCH_SIZE", 100.0, 1), (kn, "FETCH_SIZE", 28.0, 1), (kn, "FETCH_SIZE", 120.0, 2), (kn, "WRITE_SIZE", 64.0, 1), ("other(int)", "FETCH_SIZE", 7.0, 3)]
    c.executemany("insert into counters_collection values (?, ?, ?, ?)", rows)
    c.commit()
    c.close()
    f = {pt.clean(k): v for k, v in pt.per_kernel(db, "FETCH_SIZE").items()}
    assert f["norm_fwd_kernel<3, false>"] == (124.0, 2) and f["other"] == (7.0, 1)
    assert pt.per_kernel(db, "WRITE_SIZE")[kn] == (64.0, 1)
    # warm-up / set-up dispatches are dropped at the optimiser launch that ends the warm-up step (same rule as tools/rocprof_summary.py)
    c = sqlite3.connect(db)
    adam = "(anonymous namespace)::adam_multi_kernel(AdamTable, AdamHyper)"
    c.executemany("insert into counters_collection values (?, ?, ?, ?)", [(adam, "FETCH_SIZE", 1.0, 4), (kn, "FETCH_SIZE", 50.0, 5), (adam, "FETCH_SIZE", 1.0, 6)])
    c.commit()
    c.close()
    f1 = {pt.clean(k): v for k, v in pt.per_kernel(db, "FETCH_SIZE", warmup_steps=1).items()}
    assert f1["norm_fwd_kernel<3, false>"] == (50.0, 1) and f1["adam_multi_kernel"] == (1.0, 1) and "other" not in f1

    import bench
    monkeypatch.setattr("shutil.which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p, _orig=os.path.exists: False if p.endswith("rocprofv3") else _orig(p))
    table, why, step = bench.live_pmc_traffic("gpt2s_B32_L1024_C12", ["norm_fwd_kernel<3, false>"])
    assert step is None
    assert table == {} and "not found" in why


def test_synth_generators_agree():
    """helpers.synth_table_torch / synth_tensor (int64 torch arithmetic, multi-threaded, device-capable: what regenerates the 6.7 G weights of the
    full-depth Llama fixtures) == helpers.synth_table (numpy uint64: what the committed fixtures were generated with), bit for bit"""
    import numpy as np
    import helpers as H
    for rows, cols, salt, scale, row0 in [(1000, 4096, 1003, 0.07, 0), (17, 11008, 5, 0.2, 123456), (1, 100000, 11, 2.0, 0), (300, 3, 7, 0.035, 99990),
                                          (128256 - 128000, 64, 1000, 0.07, 128000)]:
        assert np.array_equal(H.synth_table(rows, cols, salt, scale, row0=row0), H.synth_table_torch(rows, cols, salt, scale, row0=row0).numpy())
    assert np.array_equal(H.synth_tensor(20000, 5, 9, 0.3, block=8192).numpy(), H.synth_table(20000, 5, 9, 0.3))


def test_product_library_reads_no_environment_variable():
    """the shipped library has no hidden state. (1) every environment read of the sources goes through mtl_env_int() (a constant in
    the product build) or sits inside an #ifdef MTL_DIAG block; (2) the header names exactly the switches DIAGNOSTIC builds read; (3) the built
    product .so does not even import getenv, and reports mtl_build_flags() == 0"""
    import re
    import subprocess
    files = ("mtl_gemm.hip", "mtl_attention.hip", "mtl_backbone.hip", "mtl_norm.hip", "mtl_elementwise.hip", "mtl_tokenizer.hip", "mtl_optim.hip",
             "mtl_stats.hip", "mtl_common.h")
    names = set()
    for f in files:
        text = (ROOT / "med-ts-llm_amd" / "csrc" / f).read_text()
        names |= set(re.findall(r'mtl_env_int\("([A-Z0-9_]+)"', text))
        depth = []                                   # stack of preprocessor conditions
        for line in text.splitlines():
            st = line.strip()
            if st.startswith("#if"):
                depth.append(st)
            elif st.startswith("#endif") and depth:
                depth.pop()
            elif st.startswith("#else") and depth:
                depth[-1] = "#else of " + depth[-1]
            if "getenv(" in line:
                assert any(d.startswith("#ifdef MTL_DIAG") for d in depth), (f, line)
                names |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', line))
    head = (ROOT / "include" / "medtsllm_hip.h").read_text()
    listed = set(re.findall(r"\bMTL_[A-Z0-9_]+\b", head[head.index("the dispatch switches"):head.index("#ifndef MEDTSLLM_HIP_H")]))
    assert names == listed, (sorted(names - listed), sorted(listed - names))
    from med_ts_llm_amd.hip import _native
    if _native.available() and not os.environ.get("MTL_LIB_PATH"):
        assert _native.lib().mtl_build_flags() == 0
        nm = subprocess.run(["nm", "-D", "--undefined-only", _native.LIB_PATH], capture_output=True, text=True)
        if nm.returncode == 0:
            assert "getenv" not in nm.stdout, "the product library imports getenv"
