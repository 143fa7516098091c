"""Evaluation path right after the hot path: window stitching + scoring of
R:tasks/forecasting.py:52-95, R:tasks/reconstruction.py:52-91 and R:tasks/anomaly_detection.py:86-163.

The reference copies every window to the host (`pred[j].squeeze().cpu()`) and slice-assigns it into a NaN-filled
[n_points, C] buffer, one sample at a time; overlapping windows (val split: step < pred_len) are resolved by write
order, i.e. the LAST window covering a point wins. Here the model outputs stay on the device and the same buffer is
produced by ONE gather: point t takes window w*(t) = the largest w with start[w] <= t (all windows have one length,
starts ascend), which is exactly the window whose write came last. Deterministic, no per-sample sync."""
import torch
import torch.nn.functional as F


def stitch_last_wins(windows, starts, n_points, fill):
    """windows [W, n, ...], starts int64 [W] ascending -> [n_points, ...]; uncovered points = fill."""
    W, n = windows.shape[0], windows.shape[1]
    dev = windows.device
    starts = torch.as_tensor(starts, dtype=torch.int64, device=dev)
    assert starts.numel() == W and (W < 2 or bool((starts[1:] > starts[:-1]).all())), "window starts must ascend"
    t = torch.arange(n_points, device=dev)
    w = torch.searchsorted(starts, t, right=True) - 1
    wc = w.clamp(min=0)
    off = t - starts[wc]
    valid = (w >= 0) & (off < n)
    out = torch.full((n_points, *windows.shape[2:]), fill, dtype=windows.dtype, device=dev)
    out[valid] = windows[wc[valid], off[valid]]
    return out


def stitch_dataset(dataset, n_points, n_features, windows, time_range_of, fill):
    """Stitch per-sample windows [W, n, C'] into [n_points, n_features] following dataset.inverse_index.
    `time_range_of(inds)` picks the (start, stop) time range out of inverse_index's return value.
    Univariate datasets return (time ranges, feature index) per sample: stitched per feature column."""
    W = windows.shape[0]
    inds = [dataset.inverse_index(i) for i in range(W)]
    if not getattr(dataset, "univariate", False):
        starts = [time_range_of(i)[0] for i in inds]
        w = windows if windows.ndim == 3 else windows.unsqueeze(-1)
        return stitch_last_wins(w, starts, n_points, fill)
    out = torch.full((n_points, n_features), fill, dtype=windows.dtype, device=windows.device)
    feats = [int(i[1]) for i in inds]
    for f in sorted(set(feats)):
        sel = [k for k, ff in enumerate(feats) if ff == f]
        starts = [time_range_of(inds[k][0])[0] for k in sel]
        out[:, f] = stitch_last_wins(windows[sel].reshape(len(sel), -1), starts, n_points, fill)
    return out


def crop_to_scored_points(dataset, tensors, n_points, step_size, pred_len):
    """R:tasks/forecasting.py:82-90 — clip mask, or drop the unscored gap when step > pred_len."""
    if getattr(dataset, "clip_dataset", False):
        mask = dataset.mask.to(tensors[0].device)
        return [t[mask] for t in tensors]
    if step_size > pred_len:
        cutoff = n_points - (n_points % step_size)
        return [t[:cutoff].reshape(-1, step_size, *t.shape[1:])[:, :pred_len].reshape(-1, *t.shape[1:]) for t in tensors]
    return tensors


def regression_scores(pred, target, prefix=""):
    return {f"{prefix}mse": F.mse_loss(pred, target).item(), f"{prefix}mae": F.l1_loss(pred, target).item()}


# ---- anomaly scoring (R:tasks/anomaly_detection.py:128-152,216-262)
def running_mean(xs, window_size):
    if window_size % 2 == 0:
        window_size += 1
    kernel = torch.ones(1, 1, window_size, dtype=xs.dtype, device=xs.device) / window_size
    return F.conv1d(xs.view(1, 1, -1), kernel, stride=1, padding="same").squeeze()


def adjust_anomalies(pred, gt):
    """Point-adjust: every ground-truth anomaly segment containing a detection is marked detected as a whole
    (vectorised restatement of the reference's sequential scan, including its quirk that the backward fill stops
    before index 0, so pred[0] is never changed)."""
    p, g = pred.to(torch.bool), gt == 1
    prev = torch.cat([torch.zeros(1, dtype=torch.bool, device=g.device), g[:-1]])
    seg = torch.cumsum((g & ~prev).to(torch.int64), 0) * g          # 1-based segment id, 0 outside segments
    hit = torch.zeros(int(seg.max().item()) + 1 if seg.numel() else 1, dtype=torch.bool, device=g.device)
    hit[seg[p & g]] = True
    hit[0] = False
    out = p | (hit[seg] & g)
    if out.numel():
        out[0] = p[0]
    return out.to(torch.int)


def anomaly_scores(preds, targets, normalize_by_feature, moving_window):
    scores = F.mse_loss(preds, targets, reduction="none")
    if normalize_by_feature:
        scores = scores / scores.mean(dim=0).unsqueeze(0)
    scores = scores.nanmean(dim=1)
    if moving_window > 0:
        scores = scores / running_mean(scores, moving_window)
    return scores


# ---------------------------------------------------------------- segmentation (boundary detection) post-processing
def _segments(points, n):
    """consecutive (start, end) pairs of 0 | points | n-1 (R:tasks/segmentation.py:142-146)"""
    edges = torch.cat([torch.tensor([0]), points.to(torch.int64).reshape(-1), torch.tensor([n - 1])])
    return torch.stack([edges[:-1], edges[1:]], dim=1)


def _boundary_result(preds, pred_points, targets):
    n = preds.shape[0]
    pred_labels = torch.zeros_like(targets)
    pred_labels[pred_points.to(torch.int64)] = 1
    label_points = targets.nonzero().squeeze()
    return {"preds_raw": preds, "pred_points": pred_points, "pred_labels": pred_labels, "pred_segments": _segments(pred_points, n),
            "labels": targets, "label_points": label_points, "label_segments": _segments(label_points, n)}


def boundaries_from_scores(preds, targets, distance_thresh):
    """R:tasks/segmentation.py:115-154 — boundary-prediction mode: peaks of the stitched per-point scores at least
    `distance_thresh` apart (scipy.signal.find_peaks, as the reference); "auto" = the 10 % quantile of the true
    segment lengths."""
    import scipy.signal
    if distance_thresh == "auto":
        seg_lens = targets.nonzero().squeeze().unfold(0, 2, 1).diff(dim=1).squeeze()
        distance_thresh = seg_lens.float().quantile(0.1).item()
    elif distance_thresh == "optimize":
        raise NotImplementedError("distance_thresh='optimize' needs bayes_opt (R:tasks/segmentation.py:297-323), which this "
                                  "image does not have; use 'auto' or a number")
    pts = scipy.signal.find_peaks(preds.numpy(), distance=distance_thresh)[0]
    return _boundary_result(preds, torch.tensor(pts, dtype=torch.int), targets)


def boundaries_from_ramps(preds, targets):
    """R:tasks/segmentation.py:156-192 — steps-to-boundary mode: a boundary shows as a maximum right after a minimum of
    the predicted ramp. Prominent maxima and minima are paired: each point of the larger family snaps to its nearest
    point of the other family when that one is closer than half the mean true segment length."""
    import scipy.signal
    targets = (targets == 0).int()
    half = targets.size(0) / targets.sum().item() / 2
    x = preds.numpy()
    hi, lo = scipy.signal.find_peaks(x, prominence=0.5)[0], scipy.signal.find_peaks(-x, prominence=0.5)[0]
    a, b = (hi, lo) if len(hi) >= len(lo) else (lo, hi)
    a, b = torch.tensor(a), torch.tensor(b)
    if a.numel() and b.numel():
        d = (b[None, :] - a[:, None]).abs()
        near = d.argmin(dim=1)                                   # first minimum on ties, as a sequential argmin
        a = torch.where(d.gather(1, near[:, None]).squeeze(1) > half, a, b[near])
    return _boundary_result(preds, a, targets)


def all_pairs_iou(seg1, seg2):
    """[n1, 2] x [n2, 2] (start, end) segments -> IoU matrix [n1, n2] (R:tasks/segmentation.py:261-273)"""
    s1, e1, s2, e2 = seg1[:, :1], seg1[:, 1:2], seg2[:, 0][None, :], seg2[:, 1][None, :]
    inter = (torch.minimum(e1, e2) - torch.maximum(s1, s2)).clamp(min=0)
    return inter / ((e1 - s1) + (e2 - s2) - inter)


def segmentation_scores(r):
    """R:tasks/segmentation.py:194-234"""
    pp, tp = r["pred_points"], r["label_points"]
    if len(pp) == 0:
        return {"point_mae": float("inf"), "point_rmse": float("inf"), "segment_miou": 0, "pred_label_ratio": 0.0}
    dist = (pp.reshape(-1, 1) - tp).abs()                       # [n_pred, n_true]
    iou = all_pairs_iou(r["pred_segments"], r["label_segments"])
    nearest = dist.min(dim=0).values.float()
    m = {"point_mae": nearest.mean().item(), "point_rmse": nearest.pow(2).mean().sqrt().item(),
         "segment_miou": iou.max(dim=0).values.float().mean().item(),
         "pred_label_ratio": r["pred_labels"].sum().item() / r["labels"].sum().item()}
    for t in (50, 100, 200):
        m[f"point_acc@{t}"] = (dist < t).any(dim=0).float().mean().item()
    for t in (0.5, 0.75, 0.9):
        m[f"segment_acc@{int(t * 100)}iou"] = (iou > t).any(dim=0).float().mean().item()
    return m
