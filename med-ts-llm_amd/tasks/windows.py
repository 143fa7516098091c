"""Sliding-window datasets over one long [n_points, C] series with the reference's indexing contract
(R:datasets/base.py): `__len__`, `inverse_index(idx)`, `step_size` (forced to pred_len on the test split, :41-42),
`n_points`, `n_features`, `real_features`, `univariate`, `clip_dataset`, train-split StandardScaler normalisation
(:79-88, sklearn's scaler as in the reference).

The reference's real loaders (ETT/PSM/LUDB/... files) are host-side I/O and stay out of scope; a `SeriesSource`
callable hands the raw arrays in, so any of them can be plugged in behind `register_series`."""
import numpy as np
import torch
from torch.utils.data import Dataset

from .synthetic import register_dataset

_SOURCES = {}


def register_series(name, source):
    """source(config, split) -> {"data": float array [n, C], optional "labels": int array [n]}; also registers the
    dataset name with the trainer's dataset registry."""
    _SOURCES[name] = source
    register_dataset(name, lambda config, split: make_series_dataset(config, split))


class SeriesDataset(Dataset):
    """generic multichannel time series."""
    univariate = False
    clip_dataset = False

    def __init__(self, config, split, source=None):
        self.config, self.split, self.task = config, split, config.task
        self.name = config.data.dataset
        self.history_len, self.pred_len = config.history_len, config.pred_len
        self.step_size = config.data.step if split != "test" else self.pred_len          # R:datasets/base.py:39-42
        self.source = source or _SOURCES[self.name]
        raw = self.source(config, split)
        data = np.asarray(raw["data"])
        self.normalizer = None
        if config.data.normalize:                                                          # R:datasets/base.py:79-88
            from sklearn.preprocessing import StandardScaler                               # the reference's own scaler
            train = data if split == "train" else np.asarray(self.source(config, "train")["data"])
            self.normalizer = StandardScaler().fit(train)
            data = self.normalizer.transform(data)
        self.data = torch.tensor(data, dtype=torch.float32)
        self.labels = None
        if raw.get("labels") is not None:
            lab = np.asarray(raw["labels"])
            self.labels = torch.tensor(lab, dtype=torch.long if len(np.unique(lab)) > 2 else torch.int32)
        self.description = raw.get("description", type(self).__doc__)
        self.task_description = raw.get("task_description")

    def denormalize(self, data):
        return self.normalizer.inverse_transform(data)

    n_points = property(lambda self: self.data.shape[0])
    n_features = property(lambda self: self.data.shape[1])
    real_features = property(lambda self: self.data.shape[1])

    @property
    def n_classes(self):
        return 0


class ForecastSeries(SeriesDataset):
    """R:datasets/base.py:116-143"""

    def __len__(self):
        return (self.n_points - self.history_len - self.pred_len + 1) // self.step_size

    def inverse_index(self, idx):
        i = idx * self.step_size
        x_range = (i, i + self.history_len)
        return x_range, (x_range[1], x_range[1] + self.pred_len)

    def __getitem__(self, idx):
        xr, yr = self.inverse_index(idx)
        return {"x_enc": self.data[slice(*xr), :], "y": self.data[slice(*yr), :]}


class ReconstructionSeries(SeriesDataset):
    """R:datasets/base.py:146-171 (AnomalyDetectionDataset :174-203 adds labels; SemanticSegmentationDataset :206-236)"""

    def __init__(self, config, split, source=None):
        super().__init__(config, split, source)
        assert self.pred_len == self.history_len

    def __len__(self):
        return (self.n_points - self.pred_len) // self.step_size + 1

    def inverse_index(self, idx):
        i = idx * self.step_size
        return (i, i + self.pred_len)

    def __getitem__(self, idx):
        r = self.inverse_index(idx)
        out = {"x_enc": self.data[slice(*r), :]}
        if self.labels is not None and self.task != "reconstruction":
            out["labels"] = self.labels[slice(*r)]
        return out


class SemSegSeries(ReconstructionSeries):
    @property
    def n_classes(self):
        return len(self.labels.unique())


def steps_to_boundary_labels(boundary):
    """R:datasets/base.py:265-277 — binary boundary marks -> regression target: at point i the distance to the next
    boundary c (first boundary >= i; the series end closes the last segment) divided by that segment's length
    (c minus the previous boundary; the first segment starts at 0). A boundary point itself gets 0."""
    b = np.asarray(boundary)
    n = len(b)
    cps = np.append(np.nonzero(b)[0], n)
    i = np.arange(n)
    j = np.searchsorted(cps, i, side="left")
    nxt = cps[j]
    prev = np.where(j > 0, cps[np.maximum(j - 1, 0)], 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((nxt - i) / (nxt - prev)).astype(np.float32)


class SegmentationSeries(ReconstructionSeries):
    """R:datasets/base.py:236-281: windows of (x_enc, labels); labels are boundary marks (boundary-prediction) or the
    normalised steps-to-boundary ramp."""

    def __init__(self, config, split, source=None):
        super().__init__(config, split, source)
        assert self.task == "segmentation" and self.labels is not None
        mode = config.tasks.segmentation.mode
        if mode == "steps-to-boundary":
            self.labels = torch.tensor(steps_to_boundary_labels(self.labels.numpy()))
        elif mode != "boundary-prediction":
            raise ValueError(f"Segmentation mode {mode} not supported")


def make_series_dataset(config, split, source=None):
    cls = {"forecasting": ForecastSeries, "pretraining": ForecastSeries, "reconstruction": ReconstructionSeries,
           "anomaly_detection": ReconstructionSeries, "semantic_segmentation": SemSegSeries, "segmentation": SegmentationSeries}[config.task]
    return cls(config, split, source)
