#!/usr/bin/env python3
"""Golden vectors for the dataset side of the path and the pre-training / fine-tuning trainer (8f-3), produced by the
REAL reference classes in THIS container only (same stub recipe as make_golden.py):

  * `ClipDataset` (R:datasets/base.py:284-335) mixed into the reference's reconstruction / anomaly / semantic-segmentation datasets over
    synthetic clip data: length, every inverse_index range, the scoring mask, per-item descriptions — and the stitched `predict()`
    outputs of the reference ReconstructionTask over it (clip mask branch of the evaluation path);
  * `PretrainingDataset` (R:datasets/util.py:46-118): subset draws, block layout, adjust_n_features, inverse_index_full;
  * `PretrainingTask` (R:tasks/pretraining.py): per-step losses of the reference trainer over a mix of four synthetic datasets;
  * fine-tuning (R:tasks/base.py:88-91,118-155): parameter groups, per-epoch learning rates and the loss trajectory of the reference
    trainer started from a pre-training checkpoint, with a warm-up schedule.

Outputs: tests/golden/datasets.npz + datasets.json, trainer_pretraining.npz, trainer_finetune.npz (data only)."""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402
from make_eval_golden import FakeRecon  # noqa: E402

OUT = Path(__file__).resolve().parent
CLIP_LENS = [70, 33, 120, 32, 57]          # one clip exactly pred_len long, one barely longer


def clip_series(split, C):
    g = torch.Generator().manual_seed({"train": 61, "val": 62, "test": 63}[split])
    n = sum(CLIP_LENS)
    t = torch.arange(n, dtype=torch.float32)
    data = torch.stack([torch.sin(t / (3.0 + 2 * c)) + 0.05 * c for c in range(C)], dim=-1) + 0.1 * torch.randn(n, C, generator=g)
    clip_ids = np.repeat(np.array([3, 4, 7, 11, 12]) + (0 if split == "train" else 100), CLIP_LENS)   # ids need not be 0..n-1
    labels = (np.arange(n) // 9) % 4
    desc = {int(c): f"Patient information: synthetic subject {int(c)}; ECG lead: II" for c in np.unique(clip_ids)}
    return data.numpy().astype(np.float32), clip_ids.astype(np.int64), labels.astype(np.int64), desc


def part_series(name, split, n, C):
    g = torch.Generator().manual_seed(hash((name, split)) % 1000 + 7 if False else {"ECG": 1, "ventilator": 2, "bidmc": 3, "ludb": 4}[name] * 10 + {"train": 1, "val": 2, "test": 3}[split])
    t = torch.arange(n, dtype=torch.float32)
    return (torch.stack([torch.sin(t / (2.0 + c + len(name))) * (1 + 0.3 * c) for c in range(C)], dim=-1) + 0.1 * torch.randn(n, C, generator=g)).numpy().astype(np.float32)


PARTS = {"ECG": (150, 1), "ventilator": (190, 5), "bidmc": (130, 3), "ludb": (170, 2)}      # name -> (points, channels)


def trainer_cfg(llm_dir, task, dataset, epochs, extra=None):
    cfgd = MG.base_config(llm_dir, task, 32, 32, "concat", "linear", MG.PROMPT_CONST)
    cfgd.update({
        "data": {"dataset": dataset, "mode": "multivariate", "cols": "all", "normalize": True, "step": 16},
        "training": {"epochs": epochs, "batch_size": 4, "optimizer": "adam", "learning_rate": 1e-3, "dropout": 0.0,
                     "loss": "mse", "eval_metric": "mse", "eval_metric_direction": "min"},
        "setup": {"seed": 0, "device": "cpu", "dtype": "fp32", "num_workers": 0, "logger": "print"},
        "datasets": {dataset: {}},
    })
    cfgd["tasks"]["pretraining"] = {"downsample_pct": 0.5, "n_features": 3}
    cfgd.update(extra or {})
    return cfgd


def record_run(trainer, prefix, out):
    """run trainer.train() with the loader order fixed, recording batches, per-step losses, per-epoch LRs and the weights"""
    from torch.utils.data import DataLoader
    losses, lrs = [], []
    orig_step, orig_epoch = trainer.log_step, trainer.log_epoch
    trainer.log_step = lambda loss: (losses.append(loss), orig_step(loss))[1]
    trainer.log_epoch = lambda scores={}, **kw: (lrs.append(list(trainer.scheduler.get_last_lr())), orig_epoch(scores, **kw))[1]
    trainer.train_dataloader = DataLoader(trainer.train_dataset, batch_size=4, shuffle=False, num_workers=0)
    for i, b in enumerate(trainer.train_dataloader):
        out[f"{prefix}batch{i}.x_enc"] = MG.t2n(b["x_enc"])
    for n, p in trainer.model.named_parameters():
        if p.requires_grad:
            out[f"{prefix}init.{n}"] = MG.t2n(p)
    trainer.train()
    for n, p in trainer.model.named_parameters():
        if p.requires_grad:
            out[f"{prefix}final.{n}"] = MG.t2n(p)
    out[f"{prefix}losses"] = np.array(losses, dtype=np.float64)
    out[f"{prefix}lrs"] = np.array(lrs, dtype=np.float64)
    out[f"{prefix}step_counter"] = np.int64(trainer.step)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        MG.setup_imports(tmp)
        d = str(Path(tmp) / "llm_gpt2")
        os.makedirs(d)
        MG.make_backbone("gpt2", d, seed=100)
        MG.make_tokenizer(d)
        import datasets as ref_datasets
        import tasks as ref_tasks
        import tasks.base as ref_base
        from utils import dict_to_object
        from datasets.base import (BaseDataset, ClipDataset, ReconstructionDataset, AnomalyDetectionDataset, SemanticSegmentationDataset)
        from datasets.util import PretrainingDataset, multi_2_uni_dataset

        out, meta = {}, {}
        C = 2

        class ClipBase(BaseDataset):
            """synthetic clips of physiological waveforms."""
            supported_tasks = ["reconstruction", "anomaly_detection", "semantic_segmentation"]

            def get_data(self, split=None):
                data, ids, labels, desc = clip_series(split or self.split, C)
                return {"data": data, "labels": labels if self.task != "reconstruction" else None, "clip_ids": ids, "clip_descriptions": desc}

        def mk(base):
            return type("Clip" + base.__name__, (ClipBase, ClipDataset, base), {"__doc__": ClipBase.__doc__})

        ref_datasets.dataset_lookup["synthetic_clips"] = {"reconstruction": mk(ReconstructionDataset), "anomaly_detection": mk(AnomalyDetectionDataset),
                                                          "semantic_segmentation": mk(SemanticSegmentationDataset)}
        for split in ("train", "val", "test"):
            data, ids, labels, desc = clip_series(split, C)
            out[f"clip.raw.{split}"], out[f"clip.ids.{split}"], out[f"clip.labels.{split}"] = data, ids, labels
            meta[f"clip.desc.{split}"] = {str(k): v for k, v in desc.items()}
        # ---- clip indexing: step 8 < pred 32 (overlap), step 40 > pred 32 (gaps), test split (step forced to pred_len)
        for task in ("reconstruction", "semantic_segmentation"):
            for step in (8, 40):
                for split in ("val", "test"):
                    cfgd = trainer_cfg(d, task, "synthetic_clips", 1)
                    cfgd["data"]["step"] = step
                    cfgd["models"]["timellm"]["prompting"] = MG.PROMPT_CLIP
                    if task == "semantic_segmentation":
                        cfgd["training"]["loss"] = "ce"
                    ds = ref_datasets.get_dataset(dict_to_object(cfgd), split)
                    k = f"clip.{task}.s{step}.{split}."
                    out[k + "len"] = np.int64(len(ds))
                    out[k + "ranges"] = np.array([ds.inverse_index(i) for i in range(len(ds))], dtype=np.int64)
                    out[k + "mask"] = ds.mask.numpy()
                    out[k + "x0"] = np.array([ds[i]["x_enc"][0, 0].item() for i in range(len(ds))], dtype=np.float32)
                    meta[k + "descriptions"] = [ds[i]["descriptions"] for i in range(len(ds))]
                    if task == "semantic_segmentation":
                        out[k + "n_classes"] = np.int64(ds.n_classes)
        # ---- the reference ReconstructionTask.predict over the clip dataset (clip-mask branch of the stitching)
        for step in (8, 40):
            cfgd = trainer_cfg(d, "reconstruction", "synthetic_clips", 1)
            cfgd["data"]["step"] = step
            tr = ref_tasks.get_trainer("DEBUG-clip-golden", dict_to_object(cfgd))
            tr.model = FakeRecon()
            for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
                p, t = tr.predict(dl)
                out[f"clip.predict.s{step}.{split}.preds"], out[f"clip.predict.s{step}.{split}.targets"] = p.numpy(), t.numpy()

        # ---- stand-ins for the four datasets the reference PretrainingTask hard-codes
        class Part(BaseDataset):
            supported_tasks = ["reconstruction"]

            def get_data(self, split=None):
                n, c = PARTS[self.name]
                return {"data": part_series(self.name, split or self.split, n, c)}

        for name in PARTS:
            ref_datasets.dataset_lookup[name] = {"reconstruction": type("Part" + name, (Part, ReconstructionDataset),
                                                                        {"__doc__": f"synthetic stand-in for the {name} dataset."})}
        # (the reference's univariate view, R:datasets/util.py:10-43, cannot index these datasets: its __getitem__ calls the base class's,
        #  which calls the OVERRIDDEN inverse_index and slices with the (range, feature) tuple — TypeError; no golden can be taken)
        for name, (n, c) in PARTS.items():
            for split in ("train", "val", "test"):
                out[f"part.{name}.{split}"] = part_series(name, split, n, c)

        # ---- PretrainingDataset on its own: subset draws, layout, channel adjustment
        torch.manual_seed(77)
        parts = {}
        for name in PARTS:
            c2 = trainer_cfg(d, "reconstruction", name, 1)
            parts[name] = ref_datasets.get_dataset(dict_to_object(c2), "train")
        for nf in (3, "auto"):
            torch.manual_seed(77)
            pds = PretrainingDataset(parts, downsample_pct=0.5, n_features=nf)
            k = f"mix.nf{nf}."
            out[k + "len"], out[k + "n_features"], out[k + "n_points"] = np.int64(len(pds)), np.int64(pds.n_features), np.int64(pds.n_points)
            out[k + "lens"], out[k + "cumsums"] = np.array(pds.lens, dtype=np.int64), np.array(pds.cumsums, dtype=np.int64)
            for j, inds in enumerate(pds.dataset_inds):
                out[k + f"inds{j}"] = inds.numpy()
            out[k + "x"] = np.stack([pds[i]["x_enc"].numpy() for i in range(len(pds))])
            out[k + "full_index"] = np.array([[pds.inverse_index_full(i)[0], *pds.inverse_index_full(i)[1]] for i in range(len(pds))], dtype=np.int64)
            out[k + "index"] = np.array([pds.inverse_index(i) for i in range(len(pds))], dtype=np.int64)
            meta[k + "names"] = [pds[i]["dataset"] for i in range(len(pds))]
            meta[k + "descriptions"] = sorted({pds[i]["dataset_description"] for i in range(len(pds))})
        meta["mix.description"] = PretrainingDataset.description
        np.savez_compressed(OUT / "datasets.npz", **out)
        (OUT / "datasets.json").write_text(json.dumps(meta, indent=1))
        print("[golden] datasets.npz:", len(out), "arrays")

        # ---- the reference PretrainingTask: loss trajectory over the mix (1 epoch), checkpoint kept for the fine-tuning run below
        logdir = Path(tmp) / "logs"
        ref_base.Path(__file__)   # (the reference resolves checkpoints relative to its own tasks/ directory: pass paths.logdir instead)
        tout = {}
        cfgd = trainer_cfg(d, "pretraining", "pretrain-mix", 1, extra={"DEBUG": False, "paths": {"logdir": str(logdir)}})
        cfgd["model"] = "timellm"
        torch.manual_seed(0)
        tr = ref_tasks.get_trainer("pretrain-golden", dict_to_object(cfgd))
        tout["mix.lens"] = np.array(tr.train_dataset.lens, dtype=np.int64)
        for j, inds in enumerate(tr.train_dataset.dataset_inds):
            tout[f"mix.inds{j}"] = inds.numpy()
        record_run(tr, "", tout)
        np.savez_compressed(OUT / "trainer_pretraining.npz", **tout)
        print(f"[golden] pretraining trainer: {len(tout['losses'])} steps, losses={tout['losses'][:3]}..., step={tr.step}")
        ckpt = torch.load(logdir / "pretrain-golden" / "checkpoints" / "latest.pt")
        assert set(ckpt) >= {"model", "epoch", "step", "run_id"}

        # ---- fine-tuning from that checkpoint (reconstruction on one dataset), warm-up schedule over 3 epochs
        fout = {}
        # the reference hard-codes <its tasks dir>/../outputs/logs/<id>/checkpoints/<ckpt>.pt (R:tasks/base.py:152): hand it the state directly
        real_load = torch.load
        torch.load = lambda path, *a, **k: real_load(logdir / "pretrain-golden" / "checkpoints" / "latest.pt", *a, **k) if "pretrain-golden" in str(path) else real_load(path, *a, **k)
        try:
            cfgd = trainer_cfg(d, "reconstruction", "bidmc", 3, extra={"finetuning": {"enabled": True, "pretrained_id": "pretrain-golden", "pretrained_ckpt": "latest",
                                                                                      "frozen_epochs": 0, "warmup_epochs": 2, "warmup_factor": 0.1}})
            cfgd["model"] = "timellm"
            torch.manual_seed(0)
            tr = ref_tasks.get_trainer("DEBUG-finetune-golden", dict_to_object(cfgd))
        finally:
            torch.load = real_load
        fmeta = {"loaded_params": list(tr.loaded_params), "groups": [[n for n, p in tr.model.named_parameters() if any(p is q for q in g["params"])] for g in tr.optimizer.param_groups]}
        record_run(tr, "", fout)
        np.savez_compressed(OUT / "trainer_finetune.npz", **fout)
        (OUT / "trainer_finetune.json").write_text(json.dumps(fmeta, indent=1))
        for k, v in ckpt["model"].items():
            fout_k = "pretrained." + k
        np.savez_compressed(OUT / "trainer_finetune.npz", **fout, **{"pretrained." + k: MG.t2n(v) for k, v in ckpt["model"].items()})
        print(f"[golden] fine-tuning trainer: {len(fout['losses'])} steps, lrs per epoch={fout['lrs'].tolist()}, groups={[len(g) for g in fmeta['groups']]}")


if __name__ == "__main__":
    main()
