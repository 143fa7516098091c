"""Evaluation path: device-side window stitching + scoring vs goldens produced by the REAL reference
task classes (tests/golden/make_eval_golden.py). The model is the same deterministic window -> output function on both
sides, so every comparison is exact (stitching is a pure copy) or to fp32 round-off (scores)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from med_ts_llm_amd.tasks import evalpath as E  # noqa: E402
from med_ts_llm_amd.tasks import get_trainer  # noqa: E402
from med_ts_llm_amd.tasks.windows import register_series  # noqa: E402
from med_ts_llm_amd.utils import dict_to_object  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "eval_stitching.npz"))


class FakeForecast(torch.nn.Module):
    supported_tasks = ["forecasting"]

    def __init__(self, pred_len):
        super().__init__()
        self.pred_len = pred_len
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, inputs):
        x = inputs["x_enc"]
        ramp = torch.arange(self.pred_len, dtype=x.dtype, device=x.device)[None, :, None] * 0.01
        return x[:, -1:, :] + 0.25 * x[:, :self.pred_len, :].flip(1) + ramp


class FakeRecon(torch.nn.Module):
    supported_tasks = ["reconstruction", "anomaly_detection"]

    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, inputs):
        x = inputs["x_enc"]
        return 0.9 * x + 0.05 * x.roll(1, dims=1) + 0.01 * x[:, :1, :]


SEMSEG_CLASSES = [4]


def _source(config, split):
    out = {"data": G[f"raw.{split}"]}
    if config.task == "anomaly_detection":
        out["labels"] = G[f"labels.{split}"]
    if config.task == "semantic_segmentation":
        out["labels"] = G[f"ss{SEMSEG_CLASSES[0]}.labels.{split}"]
    if config.task == "segmentation":
        out["labels"] = G[f"sg.labels.{split}"]
    return out


class FakeBoundary(torch.nn.Module):
    supported_tasks = ["segmentation"]

    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, inputs):
        x = inputs["x_enc"]
        return torch.sigmoid(1.5 * x[:, :, 0] - 0.5 * x[:, :, 1] + 0.2 * x[:, :1, 2])


class FakeRamp(FakeBoundary):
    def forward(self, inputs):
        x = inputs["x_enc"]
        return 0.5 + 0.6 * x[:, :, 0] + 0.15 * x[:, :, 2]


class FakeSemSeg(torch.nn.Module):
    supported_tasks = ["semantic_segmentation"]

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, inputs):
        x = inputs["x_enc"]
        if self.n_classes == 2:
            return torch.sigmoid(1.3 * x[:, :, 0] + 0.1 * x[:, :1, 1])
        w = torch.linspace(-1.0, 1.0, self.n_classes, dtype=x.dtype, device=x.device)
        return torch.softmax(x[:, :, :1] * w + 0.3 * x[:, :1, 1:2] * w.flip(0), dim=-1)


register_series("series_eval", _source)


def _trainer(task, L, pred, step, fake, extra_tasks=None, loss="mse"):
    from med_ts_llm_amd.tasks.base import BaseTask
    cfg = {"DEBUG": True, "task": task, "model": "medtsllm", "history_len": L, "pred_len": pred,
           "data": {"dataset": "series_eval", "mode": "multivariate", "cols": "all", "normalize": True, "step": step},
           "training": {"epochs": 1, "batch_size": 3, "optimizer": "adam", "learning_rate": 1e-3, "dropout": 0.0,
                        "loss": loss, "eval_metric": "mse", "eval_metric_direction": "min"},
           "setup": {"seed": 0, "device": "cpu", "dtype": "fp32", "num_workers": 0},
           "tasks": {"segmentation": {"mode": "boundary-prediction"}, **(extra_tasks or {})}}
    orig = BaseTask.build_model
    BaseTask.build_model = lambda self: setattr(self, "model", fake) or fake    # the stitching is model-agnostic
    try:
        return get_trainer("eval-golden", dict_to_object(cfg))
    finally:
        BaseTask.build_model = orig


def test_forecast_predict_matches_reference():
    tr = _trainer("forecasting", 64, 16, 8, FakeForecast(16))
    assert len(tr.val_dataset) == int(G["fc.val.len"]) and len(tr.test_dataset) == int(G["fc.test.len"])
    for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
        p, t = tr.predict(dl)
        assert np.array_equal(p.numpy(), G[f"fc.{split}.preds"]) and np.array_equal(t.numpy(), G[f"fc.{split}.targets"])
        sc = tr.score(p, t)
        assert sc["mse"] == pytest.approx(float(G[f"fc.{split}.mse"]), rel=1e-6) and sc["mae"] == pytest.approx(float(G[f"fc.{split}.mae"]), rel=1e-6)
    scores = tr.val()
    assert scores["val/mse"] == pytest.approx(float(G["fc.val.mse"]), rel=1e-6)


@pytest.mark.parametrize("tag,step", [("s8", 8), ("s40", 40)])
def test_reconstruction_predict_matches_reference(tag, step):
    tr = _trainer("reconstruction", 32, 32, step, FakeRecon())
    for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
        p, t = tr.predict(dl)
        assert np.array_equal(p.numpy(), G[f"rc.{tag}.{split}.preds"]) and np.array_equal(t.numpy(), G[f"rc.{tag}.{split}.targets"])


@pytest.mark.parametrize("tag,tcfg", [("auto_nf", {"threshold": "auto", "normalize_by_feature": True, "normalize_moving_window": 0}),
                                      ("f10_win5", {"threshold": 0.1, "normalize_by_feature": False, "normalize_moving_window": 5}),
                                      ("f05_nf_win4", {"threshold": 0.05, "normalize_by_feature": True, "normalize_moving_window": 4})])
def test_anomaly_predict_matches_reference(tag, tcfg):
    tr = _trainer("anomaly_detection", 32, 32, 8, FakeRecon(), {"anomaly_detection": {"score_metric": "mse", **tcfg}})
    for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
        r = tr.predict(dl, split=split)
        k = f"ad.{tag}.{split}."
        assert np.array_equal(r.recon_preds.numpy(), G[k + "recon_preds"]) and np.array_equal(r.recon_targets.numpy(), G[k + "recon_targets"])
        assert np.array_equal(r.anomaly_labels.numpy(), G[k + "anomaly_labels"])
        np.testing.assert_allclose(r.anomaly_scores.numpy(), G[k + "anomaly_scores"], rtol=1e-6, atol=1e-8)
        assert r.anomaly_quantile == pytest.approx(float(G[k + "quantile"]), rel=1e-12)
        assert r.anomaly_threshold == pytest.approx(float(G[k + "threshold"]), rel=1e-6)
        assert np.array_equal(r.anomaly_preds.numpy(), G[k + "anomaly_preds"])
        for name, v in tr.score_anomalies(r.anomaly_preds, r.anomaly_labels).items():
            assert v == pytest.approx(float(G[k + "score." + name]), rel=1e-9)


@pytest.mark.parametrize("ncls", [4, 2])
@pytest.mark.parametrize("tag,step", [("s8", 8), ("s40", 40)])
def test_semantic_segmentation_predict_matches_reference(ncls, tag, step):
    SEMSEG_CLASSES[0] = ncls
    tr = _trainer("semantic_segmentation", 32, 32, step, FakeSemSeg(ncls), loss="ce")
    assert tr.val_dataset.n_classes == ncls
    for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
        p, t = tr.predict(dl)
        k = f"ss{ncls}.{tag}.{split}."
        assert np.array_equal(p.numpy(), G[k + "preds"]) and np.array_equal(t.numpy(), G[k + "targets"])
        for name, v in tr.score(p, t).items():
            assert v == pytest.approx(float(G[k + "score." + name]), rel=1e-9)


@pytest.mark.parametrize("tag,tcfg,fake,loss", [("bp_auto", {"mode": "boundary-prediction", "distance_thresh": "auto"}, FakeBoundary, "bce"),
                                                ("bp_d12", {"mode": "boundary-prediction", "distance_thresh": 12}, FakeBoundary, "bce"),
                                                ("stb", {"mode": "steps-to-boundary", "distance_thresh": "auto"}, FakeRamp, "mse")])
@pytest.mark.parametrize("stag,step", [("s8", 8), ("s40", 40)])
def test_segmentation_predict_matches_reference(tag, tcfg, fake, loss, stag, step):
    """boundary detection (R:tasks/segmentation.py): stitched scores, detected points / labels / segments exact; metrics to round-off"""
    tr = _trainer("segmentation", 32, 32, step, fake(), {"segmentation": tcfg}, loss=loss)
    if tag == "stb" and stag == "s8":
        assert np.array_equal(tr.val_dataset.labels.numpy(), G["sg.stb.converted_labels.val"], equal_nan=True)
    for split, dl in (("val", tr.val_dataloader), ("test", tr.test_dataloader)):
        r = tr.predict(dl)
        k = f"sg.{tag}.{stag}.{split}."
        for name in ("preds_raw", "pred_points", "pred_labels", "pred_segments", "labels", "label_points", "label_segments"):
            assert np.array_equal(r[name].numpy(), G[k + name]), (k, name)
        sc = tr.score(r)
        assert set(sc) == {n[len(k) + 6:] for n in G.files if n.startswith(k + "score.")}
        for name, v in sc.items():
            assert v == pytest.approx(float(G[k + "score." + name]), rel=1e-6)
    assert tr.test()[f"test/segment_miou"] == pytest.approx(float(G[f"sg.{tag}.{stag}.test.score.segment_miou"]), rel=1e-6)


def test_segmentation_loss_mode_contract():
    """R:tasks/segmentation.py:58-71: bce <-> boundary-prediction, mse / mae <-> steps-to-boundary"""
    with pytest.raises(AssertionError):
        _trainer("segmentation", 32, 32, 8, FakeBoundary(), {"segmentation": {"mode": "steps-to-boundary", "distance_thresh": "auto"}}, loss="bce")
    with pytest.raises(ValueError):
        _trainer("segmentation", 32, 32, 8, FakeBoundary(), {"segmentation": {"mode": "boundary-prediction", "distance_thresh": "auto"}}, loss="ce")
    tr = _trainer("segmentation", 32, 32, 8, FakeBoundary(), {"segmentation": {"mode": "boundary-prediction", "distance_thresh": "optimize"}}, loss="bce")
    with pytest.raises(NotImplementedError):
        tr.predict(tr.val_dataloader)


def test_point_adjust_and_running_mean_match_reference():
    for i in range(8):
        out = E.adjust_anomalies(torch.tensor(G[f"adj.{i}.pred"], dtype=torch.int), torch.tensor(G[f"adj.{i}.gt"], dtype=torch.int))
        assert np.array_equal(out.numpy(), G[f"adj.{i}.out"]), i
    x = torch.tensor(G["rm.x"])
    np.testing.assert_allclose(E.running_mean(x, 4).numpy(), G["rm.w4"], rtol=1e-6)
    np.testing.assert_allclose(E.running_mean(x, 5).numpy(), G["rm.w5"], rtol=1e-6)


def test_stitch_last_wins_equals_sequential_assignment():
    """property: the one-gather stitch == the reference's ordered slice assignment, for random strides/lengths"""
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        W, n, C = int(torch.randint(1, 9, (1,), generator=g)), int(torch.randint(1, 12, (1,), generator=g)), 2
        step = int(torch.randint(1, 15, (1,), generator=g))
        starts = [3 + w * step for w in range(W)]
        npts = starts[-1] + n + 4
        win = torch.randn(W, n, C, generator=g)
        ref = torch.full((npts, C), float("nan"))
        for w in range(W):
            ref[starts[w]:starts[w] + n] = win[w]
        out = E.stitch_last_wins(win, starts, npts, float("nan"))
        assert torch.equal(torch.nan_to_num(out, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))


@pytest.mark.gpu
def test_forecast_and_anomaly_stitching_on_device():
    """same goldens with every tensor on the GPU: the stitch is one device-side gather, no per-window host copy"""
    import med_ts_llm_amd.tasks.base as B
    tr = _trainer("forecasting", 64, 16, 8, FakeForecast(16))
    tr.device = torch.device("cuda")
    tr.model = tr.model.cuda()
    p, t = tr.predict(tr.val_dataloader)
    assert np.array_equal(p.numpy(), G["fc.val.preds"]) and np.array_equal(t.numpy(), G["fc.val.targets"])
    tr = _trainer("anomaly_detection", 32, 32, 8, FakeRecon(), {"anomaly_detection": {"score_metric": "mse", "threshold": 0.1,
                                                                  "normalize_by_feature": False, "normalize_moving_window": 5}})
    tr.device = torch.device("cuda")
    tr.model = tr.model.cuda()
    r = tr.predict(tr.test_dataloader, split="test")
    assert np.array_equal(r.recon_preds.numpy(), G["ad.f10_win5.test.recon_preds"])
    assert np.array_equal(r.anomaly_preds.numpy(), G["ad.f10_win5.test.anomaly_preds"])
    tr = _trainer("segmentation", 32, 32, 8, FakeRamp(), {"segmentation": {"mode": "steps-to-boundary", "distance_thresh": "auto"}}, loss="mse")
    tr.device = torch.device("cuda")
    tr.model = tr.model.cuda()
    r = tr.predict(tr.val_dataloader)
    np.testing.assert_allclose(r["preds_raw"].numpy(), G["sg.stb.s8.val.preds_raw"], rtol=1e-6, atol=1e-7)   # fake model's fma order
    assert np.array_equal(r["pred_points"].numpy(), G["sg.stb.s8.val.pred_points"])
    assert np.array_equal(r["label_segments"].numpy(), G["sg.stb.s8.val.label_segments"])
