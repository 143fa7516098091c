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
        src = source or _SOURCES[self.name]      # used during construction only: not kept on the instance (the dataset must pickle for
        raw = src(config, split)                  # DataLoader workers under the spawn / forkserver start methods)
        data = np.asarray(raw["data"])
        self.normalizer = None
        if config.data.normalize:                                                          # R:datasets/base.py:79-88
            from sklearn.preprocessing import StandardScaler                               # the reference's own scaler
            train = data if split == "train" else np.asarray(src(config, "train")["data"])
            self.normalizer = StandardScaler().fit(train)
            data = self.normalizer.transform(data)
        self.data = torch.tensor(data, dtype=torch.float32)
        self.labels = None
        if raw.get("labels") is not None:
            lab = np.asarray(raw["labels"])
            self.labels = torch.tensor(lab, dtype=torch.long if len(np.unique(lab)) > 2 else torch.int32)
        self.description = raw.get("description", type(self).__doc__)
        self.task_description = raw.get("task_description")
        # per-point clip ids + per-clip descriptions (R:datasets/base.py:71-74): windows then never straddle two clips (ClipIndex)
        self.clip_ids = torch.tensor(np.asarray(raw["clip_ids"]), dtype=torch.int32) if raw.get("clip_ids") is not None else None
        self.clip_descriptions = raw.get("clip_descriptions")

    def _describe(self, out, start):
        """R:datasets/base.py:129-131 — the description of the clip a window starts in rides along as "descriptions" """
        if self.clip_descriptions is not None:
            out["descriptions"] = self.clip_descriptions[self.clip_ids[start].item()]
        return out

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
        return self._describe({"x_enc": self.data[slice(*xr), :], "y": self.data[slice(*yr), :]}, xr[0])


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
        return self._describe(out, r[0])


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


class ClipIndex:
    """Clip-aware window indexing (R:datasets/base.py:284-335, `ClipDataset`): the series is a concatenation of clips (runs of equal
    `clip_ids`), windows of pred_len points step through every clip separately and never cross a clip boundary; `mask` marks the points
    a stitched evaluation scores (per clip: the covered prefix, and with step > pred_len only the first pred_len points of every step).
    Mixed in FRONT of a window dataset class (the reference's `class LUDB(ClipDataset, SemanticSegmentationDataset)` pattern)."""
    clip_dataset = True

    def _build_clip_index(self):
        assert self.task != "forecasting", "clip datasets do not support forecasting"
        ids = self.clip_ids
        assert ids is not None and bool((ids.diff() >= 0).all()), "clip ids must be given per point and ascend"
        clips, self.clip_inds, self.clip_lens = ids.unique_consecutive(return_inverse=True, return_counts=True)
        assert clips.numel() == ids.unique().numel(), "a clip id may not come back after another clip"
        self.clips = torch.arange(len(clips))
        zero = torch.zeros(1, dtype=torch.int64)
        self.clip_lens_cumsum = torch.cat([zero, self.clip_lens.cumsum(0)])
        self.clip_segs = (self.clip_lens - self.pred_len) // self.step_size + 1                  # windows per clip
        self.clip_segs_cumsum = torch.cat([zero, self.clip_segs.cumsum(0)])
        self.dataset_len = int(self.clip_segs_cumsum[-1])
        covered = (self.clip_segs - 1) * self.step_size + self.pred_len                          # points of a clip its windows reach
        assert bool((self.clip_lens - covered >= 0).all())
        # point i of a clip (i < covered) is scored iff (i mod step) < pred_len — one vectorised pass instead of per-clip concatenation
        pos = torch.arange(self.n_points) - self.clip_lens_cumsum[:-1].repeat_interleave(self.clip_lens)
        cov = covered.repeat_interleave(self.clip_lens)
        self.mask = (pos < cov) & ((pos % self.step_size) // self.pred_len == 0)

    def __len__(self):
        return self.dataset_len

    def inverse_index(self, seg_idx):
        clip = int(torch.searchsorted(self.clip_segs_cumsum, seg_idx, right=True)) - 1
        start = int(self.clip_lens_cumsum[clip]) + (seg_idx - int(self.clip_segs_cumsum[clip])) * self.step_size
        return (start, start + self.pred_len)


_DERIVED = {}      # (kind, base class) -> derived class, also published as a module attribute so that instances pickle by name


def _publish(kind, base, derived, name):
    derived.__name__ = derived.__qualname__ = name
    derived.__module__ = __name__
    globals()[name] = derived
    _DERIVED[(kind, base)] = derived
    return derived


def _with_clips(cls):
    """cls + ClipIndex, built once the raw data (and with it the clip ids) is known"""
    if ("clip", cls) in _DERIVED:
        return _DERIVED[("clip", cls)]

    class Clipped(ClipIndex, cls):
        __doc__ = cls.__doc__

        def __init__(self, config, split, source=None):
            super().__init__(config, split, source)
            self._build_clip_index()
    return _publish("clip", cls, Clipped, "Clip" + cls.__name__)


def univariate_view(cls):
    """R:datasets/util.py:10-43 (`multi_2_uni_dataset`): every (window, feature) pair is its own univariate sample; sample index =
    window * real_features + feature, `inverse_index` returns (time range(s), feature)."""
    if ("uni", cls) in _DERIVED:
        return _DERIVED[("uni", cls)]

    class Univariate(cls):
        __doc__ = cls.__doc__
        univariate = True
        n_features = property(lambda self: 1)
        real_features = property(lambda self: cls.n_features.fget(self))

        _inner = False      # the wrapped class's __getitem__ looks its window up through self.inverse_index: give it the plain one
                            # (the reference's class hands it the (ranges, feature) tuple there and fails with a TypeError)

        def __getitem__(self, index):
            ex, f = divmod(index, self.real_features)
            self._inner = True
            try:
                inputs = super().__getitem__(ex)
            finally:
                self._inner = False
            for k in ("x_enc", "y", "x_dec"):
                if k in inputs:
                    inputs[k] = inputs[k][:, f:f + 1]
            return inputs

        def __len__(self):
            return super().__len__() * self.real_features

        def inverse_index(self, index):
            if self._inner:
                return super().inverse_index(index)
            return super().inverse_index(index // self.real_features), index % self.real_features
    return _publish("uni", cls, Univariate, "Univariate" + cls.__name__)


class MixedWindows(Dataset):
    """This dataset consists of a mix of different biomedical time series datasets."""
    # R:datasets/util.py:46-118 (`PretrainingDataset`): the windows of several datasets behind one index — dataset d contributes a
    # random subset (torch.randperm, drawn in the order the datasets are given, from the global torch RNG as the reference does) of
    # max(1, int(downsample_pct * len(d))) of its windows, in blocks one after the other; every window is brought to `n_features`
    # channels by tiling its channels and truncating. Items carry the source dataset's name and description.
    supported_tasks = ["pretraining"]
    univariate = False
    clip_dataset = False
    n_classes = 0

    def __init__(self, datasets, downsample_pct=1.0, n_features=None):
        self.datasets, self.dataset_names = list(datasets.values()), list(datasets.keys())
        first = self.datasets[0]
        self.config, self.split, self.task = first.config, first.split, "pretraining"
        self.name = "pretrain:" + "+".join(self.dataset_names)
        self.description = type(self).__doc__
        self.task_description = None
        self.dataset_inds = [torch.randperm(len(d))[:max(1, int(downsample_pct * len(d)))] for d in self.datasets]
        self.lens = [len(i) for i in self.dataset_inds]
        self.cumsums = [sum(self.lens[:i]) for i in range(len(self.datasets))]
        if n_features is None or n_features == "auto":
            n_features = max(d.n_features for d in self.datasets)
        self.n_features = self.real_features = n_features
        self.pred_len, self.history_len, self.step_size = first.pred_len, first.history_len, first.step_size
        self.n_points = sum(self.step_size * n for n in self.lens)

    def __len__(self):
        return sum(self.lens)

    def _locate(self, index):
        d = int(np.searchsorted(self.cumsums, index, side="right")) - 1
        return d, int(self.dataset_inds[d][index - self.cumsums[d]])

    def adjust_n_features(self, x):
        """R:datasets/util.py:100-106 — [n, c] -> [n, n_features]: channels tiled ceil(n_features / c) times, then cut"""
        if x.shape[1] < self.n_features:
            x = x.repeat(1, -(-self.n_features // x.shape[1]))
        return x[:, :self.n_features] if x.shape[1] > self.n_features else x

    def __getitem__(self, index):
        d, i = self._locate(index)
        item = dict(self.datasets[d][i])
        for k in ("x_enc", "y"):
            if k in item:
                item[k] = self.adjust_n_features(item[k])
        item["dataset"], item["dataset_description"] = self.dataset_names[d], self.datasets[d].description
        return item

    def inverse_index_full(self, index):
        d, i = self._locate(index)
        return d, self.datasets[d].inverse_index(i)

    def inverse_index(self, idx):
        # (the reference lays the mixed windows out on ONE virtual time axis, step_size apart: R:datasets/util.py:115-118)
        return (idx * self.step_size, idx * self.step_size + self.pred_len)


# every combination exists from import time on (not only once make_series_dataset has run in the parent): a DataLoader worker started with
# spawn / forkserver imports this module afresh and must find "ClipSemSegSeries", "UnivariateClipReconstructionSeries", ... by name
for _base in (ForecastSeries, ReconstructionSeries, SemSegSeries, SegmentationSeries):
    univariate_view(_base)
    if _base is not ForecastSeries:          # (clip datasets do not support forecasting)
        univariate_view(_with_clips(_base))
del _base


def make_series_dataset(config, split, source=None):
    cls = {"forecasting": ForecastSeries, "pretraining": ForecastSeries, "reconstruction": ReconstructionSeries,
           "anomaly_detection": ReconstructionSeries, "semantic_segmentation": SemSegSeries, "segmentation": SegmentationSeries}[config.task]
    src = source or _SOURCES[config.data.dataset]
    raw = src(config, split)                                                                   # read once; handed on below
    if config.task != "forecasting" and raw.get("clip_ids") is not None:                       # clip data: windows stay inside their clip
        cls = _with_clips(cls)
    if config.data.get("mode", "multivariate") == "univariate":                                # R:datasets/__init__.py:31-32
        cls = univariate_view(cls)
    return cls(config, split, lambda c, sp: raw if sp == split else src(c, sp))
