"""Task classes: per-task loss construction and target selection of the reference's train loops
(R:tasks/forecasting.py:22-24,103-113; reconstruction.py:27-29; anomaly_detection.py:44-46;
segmentation.py:32-34; semantic_segmentation.py:38-45,123-136; pretraining.py)."""
import torch

from . import evalpath as E
from .base import BaseTask


def _regression_loss(name):
    if name == "mse":
        return torch.nn.MSELoss()
    if name == "mae":
        return torch.nn.L1Loss()
    if name in ("smooth_l1", "smooth_mae"):
        return torch.nn.SmoothL1Loss()
    raise ValueError(f"Invalid loss function selection: {name}")


class _StitchedEval:
    """val()/test()/predict() of the regression tasks over a sliding-window dataset (one with `inverse_index`):
    model outputs stay on the device, windows are stitched by evalpath.stitch_dataset. Datasets of independent
    windows (tasks/synthetic.py) keep BaseTask's batch-mean loss evaluation."""

    def _windows(self, dataloader, keys):
        self.model.eval()
        cols = {k: [] for k in ("pred", *keys)}
        with torch.no_grad():
            for inputs in dataloader:           # sequential loader: sample index = position (R: (idx * bs) + j)
                inputs = self.prepare_batch(inputs)
                cols["pred"].append(self.model(inputs).float())
                for k in keys:
                    cols[k].append(inputs[k])
        return {k: torch.cat(v, dim=0) for k, v in cols.items()}

    def _stitched(self, dataloader):
        return hasattr(dataloader.dataset, "inverse_index")

    def score(self, pred, target):
        return E.regression_scores(pred, target)

    def _val_test(self, dataloader, prefix):
        if not self._stitched(dataloader):
            return self._eval_loss(dataloader, prefix)
        preds, targets = self.predict(dataloader)
        scores = {f"{prefix}/{k}": v for k, v in self.score(preds, targets).items()}
        self.log_scores(scores)
        return scores

    def val(self):
        return self._val_test(self.val_dataloader, "val")

    def test(self):
        return self._val_test(self.test_dataloader, "test")


class ForecastTask(_StitchedEval, BaseTask):
    target_key = "y"

    def build_loss(self):
        return _regression_loss(self.config.training.loss)

    def predict(self, dataloader):
        """R:tasks/forecasting.py:52-95 -> (preds, targets) fp32 [n_scored_points, n_features] on the host."""
        ds = dataloader.dataset
        if not self._stitched(dataloader):
            return BaseTask.predict(self, dataloader)
        pred_len, ctx_len, step = self.config.pred_len, self.config.history_len, ds.step_size
        n_points = ds.n_points if ds.clip_dataset else pred_len + ctx_len + ((len(ds) - 1) * step)
        w = self._windows(dataloader, ("y",))
        nan = float("nan")
        preds = E.stitch_dataset(ds, n_points, ds.real_features, w["pred"], lambda t: t[1], nan)[ctx_len:]
        targets = E.stitch_dataset(ds, n_points, ds.real_features, w["y"].float(), lambda t: t[1], nan)[ctx_len:]
        preds, targets = E.crop_to_scored_points(ds, [preds, targets], n_points, step, pred_len)
        assert not preds.isnan().any() and not targets.isnan().any()
        return preds.cpu(), targets.cpu()


class ReconstructionTask(_StitchedEval, BaseTask):
    def build_loss(self):
        return _regression_loss(self.config.training.loss)

    def compute_loss(self, inputs):
        return self.loss_fn(self.model(inputs), inputs["x_enc"].detach())

    def _stitch_recon(self, dataloader, extra=()):
        ds = dataloader.dataset
        pred_len, step = self.config.pred_len, ds.step_size
        n_points = ds.n_points if ds.clip_dataset else pred_len + ((len(ds) - 1) * step)
        w = self._windows(dataloader, ("x_enc", *extra))
        nan = float("nan")
        out = [E.stitch_dataset(ds, n_points, ds.real_features, w["pred"], lambda t: t, nan),
               E.stitch_dataset(ds, n_points, ds.real_features, w["x_enc"].float(), lambda t: t, nan)]
        for k in extra:   # labels: time only
            starts = [(ds.inverse_index(i)[0] if ds.univariate else ds.inverse_index(i))[0] for i in range(w[k].shape[0])]
            out.append(E.stitch_last_wins(w[k].reshape(w[k].shape[0], -1).to(torch.int), starts, n_points, -1))
        return E.crop_to_scored_points(ds, out, n_points, step, pred_len), n_points

    def predict(self, dataloader):
        """R:tasks/reconstruction.py:52-91"""
        if not self._stitched(dataloader):
            return BaseTask.predict(self, dataloader)
        (preds, targets), _ = self._stitch_recon(dataloader)
        assert not preds.isnan().any() and not targets.isnan().any()
        return preds.cpu(), targets.cpu()


class PretrainingTask(ReconstructionTask):
    """R:tasks/pretraining.py: reconstruction over a MIX of datasets — every split is a `MixedWindows` over the component datasets
    (each built as a plain reconstruction dataset of its own name), batches mix windows of all of them, channel counts are
    brought to one width by tiling / truncation. The component names come from `tasks.pretraining.datasets` (default: the
    reference's hard-coded four); each must be registered with the dataset registry (tasks/windows.register_series)."""
    DEFAULT_DATASETS = ("ECG", "ventilator", "bidmc", "ludb")

    def __init__(self, run_id, config, newrun=True):
        super().__init__(run_id, config, newrun)
        self.task = "pretraining"

    def build_datasets(self):
        from .synthetic import get_dataset
        from .windows import MixedWindows
        tc = self.config.tasks.pretraining
        names = list(tc.get("datasets", self.DEFAULT_DATASETS))
        parts = {"train": {}, "val": {}, "test": {}}
        for name in names:      # (component order = the order of the reference's loop: it decides the RNG draws of the subsets)
            cfg = self.config.copy()
            cfg.data.dataset = name
            cfg.task = "reconstruction"
            for split in parts:
                parts[split][name] = get_dataset(cfg, split)
        mk = lambda split: MixedWindows(parts[split], downsample_pct=tc.downsample_pct, n_features=tc.n_features)
        self.train_dataset, self.val_dataset, self.test_dataset = mk("train"), mk("val"), mk("test")


class AnomalyDetectionTask(ReconstructionTask):
    def predict(self, dataloader, split=None):
        """R:tasks/anomaly_detection.py:86-163 -> dict_to_object of stitched reconstructions, point scores, the
        quantile threshold and the point-adjusted anomaly predictions (all on the host, as the reference returns)."""
        from ..utils import dict_to_object
        if not self._stitched(dataloader):
            return BaseTask.predict(self, dataloader)
        tc = self.config.tasks.anomaly_detection
        (preds, targets, labels), n_points = self._stitch_recon(dataloader, extra=("labels",))
        assert not preds.isnan().any() and not targets.isnan().any() and not (labels < 0).any()
        scores = E.anomaly_scores(preds, targets, tc.normalize_by_feature, tc.get("normalize_moving_window", 0))
        thr = tc.threshold
        if thr == "optimize" or (thr == "optimize-test" and split == "test"):
            raise NotImplementedError("threshold optimisation needs bayes_opt (R:tasks/anomaly_detection.py:246-262), "
                                      "which this image does not have; use 'auto' or a float")
        if thr in ("auto", "optimize-test"):
            quantile = 1 - (labels.sum().item() / (n_points + self.train_dataset.n_points))
        elif isinstance(thr, float):
            quantile = 1 - thr
        else:
            raise ValueError(f"Invalid threshold selection: {thr}")
        threshold = scores.quantile(quantile)
        anomalies = E.adjust_anomalies((scores > threshold).to(torch.int), labels)
        return dict_to_object({"recon_preds": preds.cpu(), "recon_targets": targets.cpu(), "anomaly_labels": labels.cpu(),
                               "anomaly_scores": scores.cpu(), "anomaly_preds": anomalies.cpu(),
                               "anomaly_quantile": quantile, "anomaly_threshold": threshold.item()})

    def score(self, pred, target):
        return E.regression_scores(pred, target, "recon_")

    def score_anomalies(self, pred, target):
        """R:tasks/anomaly_detection.py:165-175"""
        from sklearn.metrics import accuracy_score, f1_score, jaccard_score, precision_score, recall_score, roc_auc_score
        pred, target = pred.cpu().numpy(), target.cpu().numpy()
        return {"accuracy": accuracy_score(target, pred), "f1": f1_score(target, pred, average="binary", zero_division=0),
                "auroc": roc_auc_score(target, pred), "precision": precision_score(target, pred, average="binary", zero_division=0),
                "recall": recall_score(target, pred, average="binary", zero_division=0),
                "iou": jaccard_score(target, pred, average="binary", zero_division=0)}

    def _val_test(self, dataloader, prefix):
        if not self._stitched(dataloader):
            return self._eval_loss(dataloader, prefix)
        r = self.predict(dataloader, split=prefix)
        scores = {**self.score_anomalies(r.anomaly_preds, r.anomaly_labels), **self.score(r.recon_preds, r.recon_targets),
                  "anomaly_quantile": r.anomaly_quantile, "anomaly_threshold": r.anomaly_threshold}
        scores = {f"{prefix}/{k}": v for k, v in scores.items()}
        self.log_scores(scores)
        return scores


class SegmentationTask(_StitchedEval, BaseTask):
    """boundary detection (R:tasks/segmentation.py): per-point boundary scores (BCE) or a steps-to-boundary ramp (MSE / MAE)"""

    def build_loss(self):
        mode, loss = self.config.tasks.segmentation.mode, self.config.training.loss
        if loss == "bce":                                            # R:tasks/segmentation.py:58-71
            assert mode == "boundary-prediction"
            return torch.nn.BCEWithLogitsLoss()
        if loss in ("mse", "mae"):
            assert mode == "steps-to-boundary"
            return torch.nn.MSELoss() if loss == "mse" else torch.nn.L1Loss()
        raise ValueError(f"Invalid loss function selection: {loss}")

    def compute_loss(self, inputs):
        return self.loss_fn(self.model(inputs), inputs["labels"].to(self.dtype))

    def predict(self, dataloader):
        """R:tasks/segmentation.py:73-113 -> dict of stitched scores, detected boundary points / labels / segments and the
        true ones (host tensors, as the reference returns)."""
        ds = dataloader.dataset
        if not self._stitched(dataloader):
            return BaseTask.predict(self, dataloader)
        if ds.univariate:
            raise NotImplementedError("segmentation over univariate datasets")
        mode = self.config.tasks.segmentation.mode
        pred_len, step = self.config.pred_len, ds.step_size
        n_points = ds.n_points if ds.clip_dataset else pred_len + ((len(ds) - 1) * step)
        w = self._windows(dataloader, ("labels",))
        W = w["pred"].shape[0]
        starts = [ds.inverse_index(i)[0] for i in range(W)]
        preds = E.stitch_last_wins(w["pred"].reshape(W, -1), starts, n_points, float("nan"))
        tdt = torch.int if mode == "boundary-prediction" else torch.float
        targets = E.stitch_last_wins(w["labels"].reshape(W, -1).to(tdt), starts, n_points, -1)
        preds, targets = E.crop_to_scored_points(ds, [preds, targets], n_points, step, pred_len)
        assert not preds.isnan().any() and not (targets < 0).any()
        preds, targets = preds.cpu(), targets.cpu()
        if mode == "boundary-prediction":
            return E.boundaries_from_scores(preds, targets, self.config.tasks.segmentation.distance_thresh)
        if mode == "steps-to-boundary":
            return E.boundaries_from_ramps(preds, targets)
        raise ValueError(f"Segmentation mode {mode} not supported")

    def score(self, results):
        return E.segmentation_scores(results)

    def _val_test(self, dataloader, prefix):
        if not self._stitched(dataloader):
            return self._eval_loss(dataloader, prefix)
        scores = {f"{prefix}/{k}": v for k, v in self.score(self.predict(dataloader)).items()}
        self.log_scores(scores)
        return scores


class SemanticSegmentationTask(_StitchedEval, BaseTask):
    def predict(self, dataloader):
        """R:tasks/semantic_segmentation.py:78-121 -> (class probabilities [n_scored_points, n_classes], int targets)"""
        ds = dataloader.dataset
        if not self._stitched(dataloader):
            return BaseTask.predict(self, dataloader)
        pred_len, step, n_classes = self.config.pred_len, ds.step_size, ds.n_classes
        n_points = ds.n_points if ds.clip_dataset else pred_len + ((len(ds) - 1) * step)
        w = self._windows(dataloader, ("labels",))
        tr = (lambda t: t)
        if n_classes == 2:      # the model returns P(class 1): column 0 is its complement
            p1 = E.stitch_dataset(ds, n_points, 1, w["pred"].reshape(w["pred"].shape[0], -1, 1), tr, float("nan"))[:, 0]
            preds = torch.stack([1 - p1, p1], dim=1)
        else:
            preds = E.stitch_dataset(ds, n_points, n_classes, w["pred"], tr, float("nan"))
        starts = [(ds.inverse_index(i)[0] if ds.univariate else ds.inverse_index(i))[0] for i in range(w["labels"].shape[0])]
        targets = E.stitch_last_wins(w["labels"].reshape(w["labels"].shape[0], -1).to(torch.int), starts, n_points, -1)
        preds, targets = E.crop_to_scored_points(ds, [preds, targets], n_points, step, pred_len)
        assert not preds.isnan().any() and not (targets < 0).any()
        return preds.cpu(), targets.cpu()

    def score(self, pred_scores, target):
        """R:tasks/semantic_segmentation.py:138-148"""
        from sklearn.metrics import accuracy_score, f1_score, jaccard_score, precision_score, recall_score
        avg = "binary" if pred_scores.size(1) == 2 else "macro"
        pred, target = pred_scores.argmax(dim=1).int().numpy(), target.numpy()
        return {"accuracy": accuracy_score(target, pred), "f1": f1_score(target, pred, average=avg, zero_division=0),
                "precision": precision_score(target, pred, average=avg, zero_division=0),
                "recall": recall_score(target, pred, average=avg, zero_division=0),
                "iou": jaccard_score(target, pred, average=avg, zero_division=0)}

    def build_loss(self):
        is_binary = self.train_dataset.n_classes == 2
        name = self.config.training.loss
        if is_binary and name in ("bce", "ce", "cross_entropy", "auto"):
            return torch.nn.BCEWithLogitsLoss()
        if not is_binary and name in ("ce", "cross_entropy", "auto"):
            return torch.nn.CrossEntropyLoss()
        raise ValueError(f"Invalid loss function selection: {name}")

    def compute_loss(self, inputs):
        pred = self.model(inputs)
        if pred.ndim == 3:
            return self.loss_fn(pred.permute(0, 2, 1), inputs["labels"])
        return self.loss_fn(pred, inputs["labels"].to(self.dtype))
