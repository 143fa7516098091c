#!/usr/bin/env python3
"""Golden vectors for the evaluation path, produced by the REAL reference task classes
(tasks/forecasting.py, tasks/reconstruction.py, tasks/anomaly_detection.py, tasks/semantic_segmentation.py, tasks/segmentation.py) in THIS container only.

The reference trainer is built on CPU around a tiny local GPT-2 exactly as make_golden.py does; its model is then
swapped for a deterministic window -> output function, so that the vectors pin the stitching / scoring logic
(last-write-wins on overlapping windows, context cut, step > pred_len crop, point scores, thresholds, point-adjust)
independently of backbone numerics. Outputs: tests/golden/eval_*.npz (data only)."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402

OUT = Path(__file__).resolve().parent


def series(split, n, C):
    g = torch.Generator().manual_seed({"train": 21, "val": 22, "test": 23}[split])
    t = torch.arange(n, dtype=torch.float32)
    cols = [torch.sin(t / (4.0 + 3 * c)) * (1 + 0.2 * c) + 0.02 * c * t / n for c in range(C)]
    return (torch.stack(cols, dim=-1) + 0.1 * torch.randn(n, C, generator=g)).numpy().astype(np.float32)


def labels_for(split, n):
    lab = np.zeros(n, dtype=np.int64)
    g = np.random.default_rng({"train": 31, "val": 32, "test": 33}[split])
    lab[0:5] = 1                      # a segment that starts at index 0 (point-adjust quirk)
    for s in g.integers(10, n - 12, size=6):
        lab[s:s + int(g.integers(2, 9))] = 1
    return lab


def semseg_labels(split, n, n_classes):
    g = np.random.default_rng({"train": 41, "val": 42, "test": 43}[split] + n_classes)
    lab = np.zeros(n, dtype=np.int64)
    i = 0
    while i < n:
        ln = int(g.integers(5, 30))
        lab[i:i + ln] = int(g.integers(0, n_classes))
        i += ln
    lab[:n_classes] = np.arange(n_classes)      # every class present in every split
    return lab


def boundary_labels(split, n):
    """binary boundary marks every 17-33 points (segmentation task)"""
    g = np.random.default_rng({"train": 51, "val": 52, "test": 53}[split])
    lab = np.zeros(n, dtype=np.int64)
    i = int(g.integers(6, 20))
    while i < n - 3:
        lab[i] = 1
        i += int(g.integers(17, 34))
    return lab


class FakeBoundary(torch.nn.Module):
    """eval-mode output of a boundary-prediction model: per-point scores in (0, 1), [B, L]"""

    def forward(self, inputs):
        x = inputs["x_enc"]
        return torch.sigmoid(1.5 * x[:, :, 0] - 0.5 * x[:, :, 1] + 0.2 * x[:, :1, 2])


class FakeRamp(torch.nn.Module):
    """eval-mode output of a steps-to-boundary model: a ramp-like regression, [B, L]"""

    def forward(self, inputs):
        x = inputs["x_enc"]
        return 0.5 + 0.6 * x[:, :, 0] + 0.15 * x[:, :, 2]


class FakeForecast(torch.nn.Module):
    def __init__(self, pred_len):
        super().__init__()
        self.pred_len = pred_len

    def forward(self, inputs):
        x = inputs["x_enc"]
        ramp = torch.arange(self.pred_len, dtype=x.dtype)[None, :, None] * 0.01
        return x[:, -1:, :] + 0.25 * x[:, :self.pred_len, :].flip(1) + ramp


class FakeSemSeg(torch.nn.Module):
    """eval-mode outputs of a semantic-segmentation model: class probabilities [B, L, n_cls], or P(class 1) [B, L] when binary"""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs):
        x = inputs["x_enc"]
        if self.n_classes == 2:
            return torch.sigmoid(1.3 * x[:, :, 0] + 0.1 * x[:, :1, 1])
        w = torch.linspace(-1.0, 1.0, self.n_classes, dtype=x.dtype)
        return torch.softmax(x[:, :, :1] * w + 0.3 * x[:, :1, 1:2] * w.flip(0), dim=-1)


class FakeRecon(torch.nn.Module):
    def forward(self, inputs):
        x = inputs["x_enc"]
        return 0.9 * x + 0.05 * x.roll(1, dims=1) + 0.01 * x[:, :1, :]


def build(ref_tasks, dict_to_object, llm_dir, task, L, pred, step, n, C, extra_tasks=None):
    cfgd = MG.base_config(llm_dir, task, L, pred, "concat", "linear", MG.PROMPT_CONST)
    cfgd.update({
        "data": {"dataset": "synthetic_eval", "mode": "multivariate", "cols": "all", "normalize": True, "step": step},
        "training": {"epochs": 1, "batch_size": 3, "optimizer": "adam", "learning_rate": 1e-3, "dropout": 0.0,
                     "loss": "mse", "eval_metric": "mse", "eval_metric_direction": "min"},
        "setup": {"seed": 0, "device": "cpu", "dtype": "fp32", "num_workers": 0, "logger": "print"},
        "datasets": {"synthetic_eval": {"n": n, "C": C}},
    })
    if task == "semantic_segmentation":
        cfgd["training"]["loss"] = "ce"
    if task == "segmentation":
        cfgd["training"]["loss"] = "bce" if (extra_tasks or {}).get("segmentation", {}).get("mode") == "boundary-prediction" else "mse"
    cfgd["tasks"].update(extra_tasks or {})
    return ref_tasks.get_trainer("DEBUG-eval-golden", dict_to_object(cfgd))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        MG.setup_imports(tmp)
        d = str(Path(tmp) / "llm_gpt2")
        os.makedirs(d)
        MG.make_backbone("gpt2", d, seed=100)
        MG.make_tokenizer(d)
        import datasets as ref_datasets
        import tasks as ref_tasks
        from utils import dict_to_object
        from datasets.base import (BaseDataset, ForecastDataset, ReconstructionDataset, AnomalyDetectionDataset, SemanticSegmentationDataset,
                                   SegmentationDataset)
        from tasks.anomaly_detection import adjust_anomalies as ref_adjust, running_mean as ref_running_mean

        N, C = 230, 3

        class SynthBase(BaseDataset):
            """synthetic multichannel physiological waveforms sampled at 125 Hz."""
            supported_tasks = ["forecasting", "reconstruction", "anomaly_detection", "semantic_segmentation", "segmentation"]
            semseg_classes = 4

            def get_data(self, split=None):
                split = split or self.split
                out = {"data": series(split, N, C)}
                if self.task == "anomaly_detection":
                    out["labels"] = labels_for(split, N)
                if self.task == "semantic_segmentation":
                    out["labels"] = semseg_labels(split, N, type(self).semseg_classes)
                if self.task == "segmentation":
                    out["labels"] = boundary_labels(split, N)
                return out

        def mk(base):
            return type("Synth" + base.__name__, (SynthBase, base), {"__doc__": SynthBase.__doc__})

        ref_datasets.dataset_lookup["synthetic_eval"] = {"forecasting": mk(ForecastDataset), "reconstruction": mk(ReconstructionDataset),
                                                         "anomaly_detection": mk(AnomalyDetectionDataset),
                                                         "semantic_segmentation": mk(SemanticSegmentationDataset),
                                                         "segmentation": mk(SegmentationDataset)}
        out = {"N": np.int64(N), "C": np.int64(C)}
        for split in ("train", "val", "test"):
            out[f"raw.{split}"] = series(split, N, C)
            out[f"labels.{split}"] = labels_for(split, N)

        # ---- forecasting: val (step 8 < pred 16: overlapping windows) and test (step forced to pred_len)
        tr = build(ref_tasks, dict_to_object, d, "forecasting", 64, 16, 8, N, C)
        tr.model = FakeForecast(16)
        for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
            p, t = tr.predict(dl)
            out[f"fc.{split}.preds"], out[f"fc.{split}.targets"] = p.numpy(), t.numpy()
            sc = tr.score(p, t)
            out[f"fc.{split}.mse"], out[f"fc.{split}.mae"] = np.float64(sc["mse"]), np.float64(sc["mae"])
        out["fc.val.len"], out["fc.test.len"] = np.int64(len(tr.val_dataset)), np.int64(len(tr.test_dataset))

        # ---- reconstruction: overlap (step 8), test, and step 40 > pred 32 (crop branch)
        for tag, step in (("s8", 8), ("s40", 40)):
            tr = build(ref_tasks, dict_to_object, d, "reconstruction", 32, 32, step, N, C)
            tr.model = FakeRecon()
            for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
                p, t = tr.predict(dl)
                out[f"rc.{tag}.{split}.preds"], out[f"rc.{tag}.{split}.targets"] = p.numpy(), t.numpy()

        # ---- anomaly detection: threshold / normalisation variants
        variants = {"auto_nf": {"threshold": "auto", "normalize_by_feature": True, "normalize_moving_window": 0},
                    "f10_win5": {"threshold": 0.1, "normalize_by_feature": False, "normalize_moving_window": 5},
                    "f05_nf_win4": {"threshold": 0.05, "normalize_by_feature": True, "normalize_moving_window": 4}}
        for tag, tcfg in variants.items():
            tr = build(ref_tasks, dict_to_object, d, "anomaly_detection", 32, 32, 8, N, C,
                       extra_tasks={"anomaly_detection": {"score_metric": "mse", **tcfg}})
            tr.model = FakeRecon()
            for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
                r = tr.predict(dl, split=split)
                for k in ("recon_preds", "recon_targets", "anomaly_labels", "anomaly_scores", "anomaly_preds"):
                    out[f"ad.{tag}.{split}.{k}"] = r[k].numpy()
                out[f"ad.{tag}.{split}.quantile"] = np.float64(r["anomaly_quantile"])
                out[f"ad.{tag}.{split}.threshold"] = np.float64(r["anomaly_threshold"])
                sc = tr.score_anomalies(r.anomaly_preds, r.anomaly_labels)
                for k, v in sc.items():
                    out[f"ad.{tag}.{split}.score.{k}"] = np.float64(v)

        # ---- semantic segmentation: 4 classes and binary; overlap (step 8) and step 40 > pred 32
        for ncls in (4, 2):
            SynthBase.semseg_classes = ncls
            for split in ("train", "val", "test"):
                out[f"ss{ncls}.labels.{split}"] = semseg_labels(split, N, ncls)
            for tag, step in (("s8", 8), ("s40", 40)):
                tr = build(ref_tasks, dict_to_object, d, "semantic_segmentation", 32, 32, step, N, C)
                tr.model = FakeSemSeg(ncls)
                for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
                    p, t = tr.predict(dl)
                    out[f"ss{ncls}.{tag}.{split}.preds"], out[f"ss{ncls}.{tag}.{split}.targets"] = p.numpy(), t.numpy()
                    for k, v in tr.score(p, t).items():
                        out[f"ss{ncls}.{tag}.{split}.score.{k}"] = np.float64(v)

        # ---- segmentation (boundary detection): both label modes, distance threshold "auto" and fixed, overlap and step > pred
        for split in ("train", "val", "test"):
            out[f"sg.labels.{split}"] = boundary_labels(split, N)
        seg_variants = {"bp_auto": ({"mode": "boundary-prediction", "distance_thresh": "auto"}, FakeBoundary()),
                        "bp_d12": ({"mode": "boundary-prediction", "distance_thresh": 12}, FakeBoundary()),
                        "stb": ({"mode": "steps-to-boundary", "distance_thresh": "auto"}, FakeRamp())}
        for tag, (tcfg, fake) in seg_variants.items():
            for stag, step in (("s8", 8), ("s40", 40)):
                tr = build(ref_tasks, dict_to_object, d, "segmentation", 32, 32, step, N, C, extra_tasks={"segmentation": tcfg})
                tr.model = fake
                if tag == "stb" and stag == "s8":
                    out["sg.stb.converted_labels.val"] = tr.val_dataset.labels.numpy()
                for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
                    r = tr.predict(dl)
                    k = f"sg.{tag}.{stag}.{split}."
                    for name in ("preds_raw", "pred_points", "pred_labels", "pred_segments", "labels", "label_points", "label_segments"):
                        out[k + name] = r[name].numpy()
                    for name, v in tr.score(r).items():
                        out[k + "score." + name] = np.float64(v)

        # ---- point-adjust and running mean on their own (incl. a segment starting at index 0)
        rng = np.random.default_rng(5)
        for i in range(8):
            n = int(rng.integers(1, 60))
            gt = (rng.random(n) < 0.45).astype(np.int32)
            if i % 2 == 0 and n > 3:
                gt[:3] = 1
            pred = (rng.random(n) < 0.2).astype(np.int32)
            adj = ref_adjust(torch.tensor(pred, dtype=torch.int), torch.tensor(gt, dtype=torch.int))
            out[f"adj.{i}.pred"], out[f"adj.{i}.gt"], out[f"adj.{i}.out"] = pred, gt, adj.numpy()
        xs = torch.tensor(rng.random(37), dtype=torch.float32)
        out["rm.x"] = xs.numpy()
        out["rm.w4"], out["rm.w5"] = ref_running_mean(xs, 4).numpy(), ref_running_mean(xs, 5).numpy()
        np.savez_compressed(OUT / "eval_stitching.npz", **out)
        print("[golden] eval_stitching.npz:", len(out), "arrays; fc.val", out["fc.val.preds"].shape, "rc.s40.val", out["rc.s40.val.preds"].shape)


if __name__ == "__main__":
    main()
