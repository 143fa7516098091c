"""Shared test helpers: golden loading and norm-wise relative error."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"

CASES = [
    "gpt2_concat_fc", "gpt2_indep_recon", "gpt2_interleave_ad", "gpt2_uni_seg",
    "llama_concat_semseg", "llama_add_fc", "llama_wavg_fc", "llama_mergeend_fc", "llamagqa_concat_fc",
    "gpt2_concat_fc_examples", "llama_add_semseg_examples",     # "examples" prompting: a tensor part inside the prompt
]


def prompt_parts_with_examples(meta, data):
    """golden per-part token ids, with the "<TENSOR>" parts replaced by the sample's example tensor [1, L_ex, C]"""
    out = []
    for b, parts in enumerate(meta["prompt_token_ids"]):
        out.append([torch.from_numpy(data["examples"][b:b + 1]) if ids is None else ids for ids in parts])
    return out


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


# ----------------------------------------------------------------------------- GPU-test model configs (HIP-friendly sizes)
def hf_cfg(kind, vocab=512):
    if kind == "gpt2":
        return {"model_type": "gpt2", "vocab_size": vocab, "n_positions": 256, "n_embd": 128, "n_layer": 2, "n_head": 2,
                "layer_norm_epsilon": 1e-5}
    if kind == "llama":
        return {"model_type": "llama", "vocab_size": vocab, "hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 2,
                "num_attention_heads": 4, "num_key_value_heads": 4, "head_dim": 64, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
    if kind == "llama_gqa":
        return {"model_type": "llama", "vocab_size": vocab, "hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 2,
                "num_attention_heads": 4, "num_key_value_heads": 2, "head_dim": 64, "rms_norm_eps": 1e-5, "rope_theta": 500000.0}
    raise ValueError(kind)


def model_config(task, L, pred, cov, down, prompting, d_model=8, d_ff=64, H=2, num_tokens=64, dropout=0.0, llm_layers=-1):
    return {
        "DEBUG": True, "task": task, "model": "medtsllm", "history_len": L, "pred_len": pred,
        "training": {"dropout": dropout}, "setup": {"dtype": "mixed"},
        "tasks": {"segmentation": {"mode": "boundary-prediction"}},
        "models": {"timellm": {
            "d_model": d_model, "d_ff": d_ff, "n_heads": H, "num_tokens": num_tokens, "covariate_mode": cov,
            "embedding_downsample_mode": down, "patching": {"patch_len": 16, "stride": 8}, "prompting": prompting,
            "llm": {"enabled": True, "llm": "in-memory", "llm_layers": llm_layers, "load_in_4bit": False, "load_in_8bit": False},
        }},
    }


class FakeDataset:
    def __init__(self, n_features, n_classes=0):
        self.description = "synthetic multichannel physiological waveforms sampled at 125 Hz."
        self.n_features, self.n_classes, self.task_description = n_features, n_classes, None


def fixture_tokenizer(kind="gpt2"):
    from transformers import PreTrainedTokenizerFast
    tok = PreTrainedTokenizerFast(tokenizer_file=str(GOLDEN / f"tokenizer_{kind}.json"), bos_token="<|endoftext|>",
                                  eos_token="<|endoftext|>")
    tok.pad_token = tok.eos_token
    return tok
