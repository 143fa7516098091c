"""Shared test helpers: golden loading and norm-wise relative error."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"

CASES = [
    "gpt2_concat_fc", "gpt2_indep_recon", "gpt2_interleave_ad", "gpt2_uni_seg",
    "llama_concat_semseg", "llama_add_fc", "llama_wavg_fc", "llama_mergeend_fc", "llamagqa_concat_fc",
]


def _flat(a):
    return torch.as_tensor(a).detach().cpu().double().flatten()


def rel_err(a, b):
    a, b = _flat(a), _flat(b)
    return float((a - b).norm() / (b.norm() + 1e-30))


def abs_err(a, b):
    return float((_flat(a) - _flat(b)).norm())


def load_case(name):
    meta = json.loads((GOLDEN / f"case_{name}.json").read_text())
    z = np.load(GOLDEN / f"case_{name}.npz")
    data = {k: z[k] for k in z.files}
    bcfg = json.loads((GOLDEN / f"backbone_{meta['backbone']}.json").read_text())
    zb = np.load(GOLDEN / f"backbone_{meta['backbone']}.npz")
    backbone = {k: torch.from_numpy(zb[k]) for k in zb.files}
    return meta, data, bcfg, backbone


def oracle_mcfg(meta):
    task = meta["task"]
    if task in ("forecasting", "reconstruction", "anomaly_detection", "pretraining"):
        nops = meta["C"]
    elif task == "semantic_segmentation":
        nops = meta["n_classes"] if meta["n_classes"] > 2 else 1
    else:
        nops = 1
    return dict(task=task, pred_len=meta["pred_len"], patch_len=meta["patch_len"], stride=meta["stride"],
                n_heads=meta["n_heads"], d_ff=meta["d_ff"], covariate_mode=meta["covariate_mode"],
                embedding_downsample_mode=meta["embedding_downsample_mode"], n_outputs_per_step=nops,
                n_classes=meta["n_classes"], seg_mode="boundary-prediction")


def golden_loss(pred, target, task):
    import torch.nn.functional as F
    if task == "semantic_segmentation":
        return F.cross_entropy(pred.permute(0, 2, 1), torch.as_tensor(target).long())
    if task == "segmentation":
        return F.binary_cross_entropy_with_logits(pred, torch.as_tensor(target).float())
    return F.mse_loss(pred, torch.as_tensor(target).float())
