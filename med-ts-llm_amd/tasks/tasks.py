"""Task classes: per-task loss construction and target selection of the reference's train loops
(R:tasks/forecasting.py:22-24,103-113; reconstruction.py:27-29; anomaly_detection.py:44-46;
segmentation.py:32-34; semantic_segmentation.py:38-45,123-136; pretraining.py)."""
import torch

from .base import BaseTask


def _regression_loss(name):
    if name == "mse":
        return torch.nn.MSELoss()
    if name == "mae":
        return torch.nn.L1Loss()
    if name in ("smooth_l1", "smooth_mae"):
        return torch.nn.SmoothL1Loss()
    raise ValueError(f"Invalid loss function selection: {name}")


class ForecastTask(BaseTask):
    target_key = "y"

    def build_loss(self):
        return _regression_loss(self.config.training.loss)


class PretrainingTask(ForecastTask):
    pass


class ReconstructionTask(BaseTask):
    def build_loss(self):
        return _regression_loss(self.config.training.loss)

    def compute_loss(self, inputs):
        return self.loss_fn(self.model(inputs), inputs["x_enc"].detach())


class AnomalyDetectionTask(ReconstructionTask):
    pass


class SegmentationTask(BaseTask):
    def build_loss(self):
        mode = self.config.tasks.segmentation.mode
        return torch.nn.BCEWithLogitsLoss() if mode == "boundary-prediction" else _regression_loss(self.config.training.loss)

    def compute_loss(self, inputs):
        return self.loss_fn(self.model(inputs), inputs["labels"].to(self.dtype))


class SemanticSegmentationTask(BaseTask):
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
