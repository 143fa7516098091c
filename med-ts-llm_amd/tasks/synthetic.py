"""Synthetic window datasets honouring the reference's batch-dict contract (R:datasets/base.py:122-133,180-192):
{"x_enc": [L, C] float32, "y" | "labels": ...}. The reference's real loaders are host-side I/O and out of scope
(SURVEY.md §2 #8); `register_dataset` lets a caller plug any Dataset class in under a config name."""
import torch
from torch.utils.data import Dataset

_REGISTRY = {}


def register_dataset(name, factory):
    """factory(config, split) -> Dataset with .description, .n_features, .n_classes"""
    _REGISTRY[name] = factory


def get_dataset(config, split):
    name = config.data.dataset
    if name not in _REGISTRY:
        raise ValueError(f"dataset {name!r} is not registered (built-in: {sorted(_REGISTRY)})")
    return _REGISTRY[name](config, split)


class SyntheticWindows(Dataset):
    """synthetic multichannel physiological waveforms sampled at 125 Hz."""

    def __init__(self, config, split):
        ds = config.get("datasets", {}).get("synthetic", {}) if hasattr(config, "get") else {}
        get = ds.get if hasattr(ds, "get") else (lambda k, d=None: d)
        self.n_features = get("n_features", 3)
        self.n_classes = get("n_classes", 4 if config.task == "semantic_segmentation" else 0)
        self.n = get("n_windows", 64)
        self.task, self.L, self.pred = config.task, config.history_len, config.pred_len
        self.description = self.__doc__
        self.task_description = None
        g = torch.Generator().manual_seed({"train": 11, "val": 12, "test": 13}[split])
        T = self.L + self.pred
        t = torch.arange(T, dtype=torch.float32)
        phase = torch.rand(self.n, 1, self.n_features, generator=g) * 6.28
        freq = 0.05 + 0.2 * torch.rand(self.n, 1, self.n_features, generator=g)
        self.data = torch.sin(t[None, :, None] * freq + phase) + 0.1 * torch.randn(self.n, T, self.n_features, generator=g)
        # learnable labels: the amplitude bucket of channel 0 at the same time step
        nb = max(self.n_classes, 2)
        self.labels = torch.clamp(((self.data[:, :self.pred, 0] + 1.2) / 2.4 * nb).long(), 0, nb - 1)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        x = self.data[i, :self.L]
        if self.task in ("forecasting", "pretraining"):
            return {"x_enc": x, "y": self.data[i, self.L:]}
        if self.task in ("reconstruction", "anomaly_detection"):
            return {"x_enc": x}
        if self.task == "semantic_segmentation":
            return {"x_enc": x, "labels": self.labels[i]}
        return {"x_enc": x, "labels": (self.labels[i] == 0).float()}


register_dataset("synthetic", SyntheticWindows)
