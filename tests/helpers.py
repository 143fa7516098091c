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
    "llamagqa_bigvocab_recon",                                  # vocabulary > 100 000: trainable sub-sampled word embeddings
]


def synth_table(rows, cols, salt, scale):
    """Deterministic [rows, cols] fp32 table from integer hashing — the same function tests/golden/make_golden.py used to FILL the
    100 000-row tensors of the vocabulary > 100 000 fixture, so they are regenerated here instead of stored."""
    i = np.arange(rows, dtype=np.uint64)[:, None]
    j = np.arange(cols, dtype=np.uint64)[None, :]
    h = (i * np.uint64(2654435761) + j * np.uint64(40503) + np.uint64(salt * 97 + 1)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return ((h.astype(np.float64) / 4294967296.0 - 0.5) * scale).astype(np.float32)


def big_grad_summary(g, stride):
    """(norm, row projection, column projection, strided sample) of a 100 000-wide gradient, as make_golden.big_grad_summary"""
    g = torch.as_tensor(g).detach().cpu().double().numpy()
    long_axis = 0 if g.shape[0] >= g.shape[1] else 1
    u = synth_table(1, g.shape[0], 11, 2.0)[0].astype(np.float64)
    v = synth_table(1, g.shape[1], 12, 2.0)[0].astype(np.float64)
    return (float(np.linalg.norm(g)), g @ v, u @ g, np.take(g, np.arange(0, g.shape[long_axis], stride), axis=long_axis))


def prompt_parts_with_examples(meta, data):
    """golden per-part token ids, with the "<TENSOR>" parts replaced by the sample's example tensor [1, L_ex, C]"""
    out = []
    for b, parts in enumerate(meta["prompt_token_ids"]):
        out.append([torch.from_numpy(data["examples"][b:b + 1]) if ids is None else ids for ids in parts])
    return out


# ----------------------------------------------------------------------------- THE end-to-end parity bars (stated once; SURVEY.md 8c ladder L3)
# The HIP path computes GEMM / attention operands in bf16 with fp32 accumulation — the reference's dtype = "mixed" arithmetic — and is compared with
# fp32 results (reference goldens or the oracle). north_star's "1e-3 rel on bf16 logits" is below what ANY bf16 path can reach against fp32 (SURVEY 0:
# the reference's own mixed run deviates by 7.8e-3 on a 12-layer stack), so the bar is relative to that self-error, measured per tensor on the same model:
MIXED_FACTOR = 1.5      # error(HIP vs fp32) <= 1.5 x error(reference-mixed vs fp32) of the same tensor ...
FWD_FLOOR = 4e-3        # ... but never tighter than one bf16 rounding of a stage output and its operands (2^-9 max, ~1.1e-3 rms each)
GRAD_FLOOR = 1e-2       # ... resp. the three to four roundings a gradient passes
SMALL_NUMEL = 4096      # tensors with fewer elements (bias vectors, the patch convolution, the 1 x C feature weighting) are sums with cancellation:
SMALL_FACTOR = 3.0      #     the RATIO of two single error samples scatters — 3 x per tensor, and every such sum is additionally pinned EXACTLY
EXACT_SUM = 2e-5        #     (cancellation_checks: equal to the fp64 reduction of the path's own upstream gradient to 2e-5 of its L1 mass)
LONGT_GRAD_FACTOR = 2.0 # long sequences: ~1 800 query rows per sample against 64 shared prototypes — the key / query projection gradients' error ratio between
                        #     two bf16 paths scatters 0.8 .. 2.0 from one tile configuration to the next (tests/test_gpu_longT.py; measured 1.58e-2 vs 7.9e-3)
SAME_ARITH_FWD = 1.2e-2   # two tilings / schedules of the SAME bf16 arithmetic at full size (pruned vs full backward, cached vs full forward, B = 32 row vs
SAME_ARITH_GRAD = 2.5e-2  #     B = 1 run): one-ulp flips of bf16 activations propagate through 12 - 32 layers; floor measured by tools/grad_noise.py
                          #     (profiles/r01_grad_noise_floor.txt: 1e-3 .. 9e-3 per tensor, 1.9e-2 on the analytically-zero key bias). A real bug is O(1).
SAME_ARITH_SAMPLE = 2e-2  #     sample independence (B = 32 row i vs the B = 1 run of sample i: different GEMM tile / split order)


def fwd_bar(self_err):
    """bar of a forward tensor given the reference-mixed arithmetic's own error on it"""
    return MIXED_FACTOR * max(self_err, FWD_FLOOR)


def grad_factor(numel, factor=MIXED_FACTOR):
    return SMALL_FACTOR if numel < SMALL_NUMEL else factor


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
    for k, spec in (bcfg.get("synth") or {}).items():       # formula-generated tables (vocabulary > 100 000 fixture)
        backbone[k] = torch.from_numpy(synth_table(**spec))
    if meta.get("synth"):
        data["param.mapping_layer.weight"] = synth_table(**meta["synth"]["mapping_layer.weight"])
        emb = backbone["embed_tokens.weight"]
        inds = torch.linspace(0, emb.shape[0] - 1, 100_000, dtype=torch.long)     # R:models/medtsllm.py:220-222
        data["param.word_embeddings"] = emb[inds].numpy().copy()
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
    ft = pred.dtype if pred.dtype == torch.bfloat16 else torch.float32      # (setup.dtype = "bf16": prepare_batch hands bf16 targets over)
    if task == "segmentation":
        return F.binary_cross_entropy_with_logits(pred, torch.as_tensor(target).to(ft))
    return F.mse_loss(pred, torch.as_tensor(target).to(ft))


# ----------------------------------------------------------------------------- GPU-test model configs (HIP-friendly sizes)
def hf_cfg(kind, vocab=512):
    if kind == "gpt2":
        return {"model_type": "gpt2", "vocab_size": vocab, "n_positions": 256, "n_embd": 128, "n_layer": 2, "n_head": 2,
                "layer_norm_epsilon": 1e-5, "embd_pdrop": 0.0, "attn_pdrop": 0.0, "resid_pdrop": 0.0}
    if kind == "llama":
        return {"model_type": "llama", "vocab_size": vocab, "hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 2,
                "num_attention_heads": 4, "num_key_value_heads": 4, "head_dim": 64, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
    if kind == "llama_gqa":
        return {"model_type": "llama", "vocab_size": vocab, "hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 2,
                "num_attention_heads": 4, "num_key_value_heads": 2, "head_dim": 64, "rms_norm_eps": 1e-5, "rope_theta": 500000.0}
    if kind == "llama_hd128":        # Llama-2-7B's head geometry (hd 128, MHA) on a small stack
        return {"model_type": "llama", "vocab_size": vocab, "hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 2,
                "num_attention_heads": 2, "num_key_value_heads": 2, "head_dim": 128, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
    if kind == "llama_gqa_hd128":    # Llama-3-8B's (hd 128, 4 query heads per KV head); n_heads * head_dim = 512 != hidden_size
        return {"model_type": "llama", "vocab_size": vocab, "hidden_size": 256, "intermediate_size": 384, "num_hidden_layers": 2,
                "num_attention_heads": 4, "num_key_value_heads": 1, "head_dim": 128, "rms_norm_eps": 1e-5, "rope_theta": 500000.0}
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
    """the BPE tokenizer trained in-process by make_golden.py (the same file for every fixture backbone)"""
    from transformers import PreTrainedTokenizerFast
    tok = PreTrainedTokenizerFast(tokenizer_file=str(GOLDEN / "tokenizer.json"), bos_token="<|endoftext|>",
                                  eos_token="<|endoftext|>")
    tok.pad_token = tok.eos_token
    return tok


# ----------------------------------------------------------------------------- host replicas of the library's dropout mask
def _drop_base(seed, stream):
    """drop_base() of csrc/mtl_common.h in uint32 arithmetic on numpy arrays / ints"""
    M, u = np.uint64(0xFFFFFFFF), np.uint64
    seed, stream = np.asarray(seed, dtype=np.uint64), np.asarray(stream, dtype=np.uint64)
    a = (seed ^ ((stream * u(0x9E3779B1)) & M)) & M
    a ^= a >> u(16)
    a = (a * u(0x85EBCA6B)) & M
    a ^= a >> u(13)
    a = (a * u(0xC2B2AE35)) & M
    a ^= a >> u(16)
    return a


def _mad24(a, b, c):
    M, u = np.uint64(0xFFFFFFFF), np.uint64
    return (((a & u(0xFFFFFF)) * (u(b) & u(0xFFFFFF))) + c) & M


def drop_u16(seed, stream, a, b):
    """the 16-bit mask field of element (stream, a, b): drop_field(drop_quad(drop_base(seed, stream), a, b >> 2), b) of mtl_common.h"""
    M, u = np.uint64(0xFFFFFFFF), np.uint64
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    h = (_mad24(a, 0x9E3779, _drop_base(seed, stream)) ^ _mad24(b >> u(2), 0x85EBCB, u(0))) & M
    h ^= h >> u(15)
    h = _mad24(h, 0xC2B2AF, h >> u(24))
    h ^= h >> u(13)
    g = _mad24(h, 0x27D4EB, h >> u(8))
    g ^= g >> u(15)
    w = np.where((b & u(2)) == 0, h, g)
    return np.where((b & u(1)) == 1, w >> u(16), w & u(0xFFFF))


def drop_threshold(p):
    t = np.float32(p) * np.float32(65536.0)
    return 65535 if t >= 65535 else (0 if t <= 0 else int(t))


def drop_scale(p):
    """the library scales kept values by the EXACT inverse keep rate of its 16-bit threshold"""
    return float(np.float32(65536.0) / np.float32(65536 - drop_threshold(p)))


def drop_site_seed(seed, layer, k):
    return (seed ^ ((0x9E3779B9 * (3 * layer + k + 1)) & 0xFFFFFFFF)) & 0xFFFFFFFF


def drop_mult_matrix(seed, p, rows, cols):
    """[rows, cols] multipliers keep * scale of the (seed, 0, row, col) mask (GEMM resid / norm-bwd / embd dropout)"""
    r, c = np.meshgrid(np.arange(rows, dtype=np.uint64), np.arange(cols, dtype=np.uint64), indexing="ij")
    keep = drop_u16(seed, 0, r, c) >= drop_threshold(p)
    return torch.from_numpy(keep.astype(np.float32)) * drop_scale(p)


def drop_mult_attention(seed, p, B, H, Tq, Tk):
    bh, q, k = np.meshgrid(np.arange(B * H, dtype=np.uint64), np.arange(Tq, dtype=np.uint64), np.arange(Tk, dtype=np.uint64), indexing="ij")
    keep = drop_u16(seed, bh, q, k) >= drop_threshold(p)
    return torch.from_numpy(keep.astype(np.float32).reshape(B, H, Tq, Tk)) * drop_scale(p)


# ----------------------------------------------------------------------------- gradients that are sums with cancellation
def cancellation_checks(tap, grads, floor):
    """For the gradients that are contractions of an upstream gradient A with a layer input X over many rows — bias vectors
    (X = ones) and the tiny feature-weighting layer — returns (exact, cond):
      exact[name] = (error of the returned gradient against the fp64 contraction of the HIP path's OWN upstream gradient, L1 mass)
      cond[name]  = floor * |A|_F * |X|_F / sqrt(rows): what a relative perturbation `floor` of A's elements moves the sum by,
                    however small the (cancelled) sum itself is — the absolute error allowance on that gradient.
    `tap` = MedTsLLM.debug_tap after backward (the i-th tensor tapped under a name is "name@i", its gradient "grad:name@i")."""
    def up(name):
        gs = [tap[k] for k in sorted(tap) if k.startswith(f"grad:{name}@")]
        return torch.cat([g.double().cpu().reshape(-1, g.shape[-1]) for g in gs], dim=0)

    exact, cond = {}, {}

    def add(pname, want64, mass, allowance):
        e = float((grads[pname].detach().cpu().double().flatten() - want64.flatten()).norm())
        exact[pname], cond[pname] = (e, mass), allowance

    rl = "reprogramming_layer."
    for pname, tname in ((rl + "query_projection.bias", "q"), (rl + "key_projection.bias", "k"), (rl + "value_projection.bias", "v"),
                         (rl + "out_projection.bias", "reprog"), ("embedding_downsample_layer.bias", "down"), ("output_projection.linear.bias", "head")):
        if pname in grads and any(k.startswith(f"grad:{tname}@") for k in tap):
            dy = up(tname)
            add(pname, dy.sum(0), float(dy.abs().sum(0).norm()), floor * float(dy.norm()))
    if any(k.startswith("grad:source@") for k in tap):
        ds = sum(tap[k].double().cpu() for k in tap if k.startswith("grad:source@"))
        add("mapping_layer.bias", ds.sum(1), float(ds.abs().sum(1).norm()), floor * float(ds.norm()))
    if "feature_weighting.weight" in grads:
        x = torch.cat([tap[k].double().cpu().reshape(-1, tap[k].shape[-1]) for k in sorted(tap) if k.startswith("fw_in@")], dim=0)
        dy = up("fw_out")
        add("feature_weighting.weight", dy.t() @ x, float((dy.abs().t() @ x.abs()).norm()), floor * float(dy.norm()) * float(x.norm()) / x.shape[0] ** 0.5)
        add("feature_weighting.bias", dy.sum(0), float(dy.abs().sum(0).norm()), floor * float(dy.norm()))
    return exact, cond


# ----------------------------------------------------------------------------- the product trainer on the CPU: device math by the oracle
def register_oracle_math_model(name="medtsllm_oracle_math"):
    """Registers (and returns the key of) a model class that keeps the PRODUCT's constructor, parameter names, checkpoint filters and
    prompt builder, and swaps only the device math of forward() for the pinned oracle — the HIP model has no CPU path, and this lets
    the product TRAINER (tasks/base.py, tasks/tasks.py) be driven against reference goldens in the CPU suite. The same replays run
    with the HIP model itself in tests/test_gpu_golden.py. Remove the key from model_lookup when done."""
    from oracle import medtsllm_oracle as O
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.medtsllm import MedTsLLM

    class OracleMath(MedTsLLM):
        def forward(self, inputs):
            x = inputs["x_enc"]
            m = dict(task=self.task, pred_len=self.pred_len, patch_len=self.patch_len, stride=self.stride, n_heads=self.n_attention_heads,
                     d_ff=self.d_ff, covariate_mode=self.covariate_mode, embedding_downsample_mode=self.embedding_downsample_mode,
                     n_outputs_per_step=self.n_outputs_per_step, n_classes=self.n_classes, seg_mode="boundary-prediction")
            ids = None
            parts = self.build_prompt(inputs)
            if len(parts[0]):
                tok = self._get_tokenizer()
                ids = [[tok(s, padding=False, truncation=False).input_ids for s in ps] for ps in parts]
            p = {n: t for n, t in self.named_parameters() if n != "word_embeddings"}
            pad = self._get_tokenizer().pad_token_id if ids is not None else 0
            return O.medtsllm_forward(x, p, self._hf_state, self._hf_cfg, m, token_ids=ids, pad_token_id=pad, training=self.training)

    model_lookup[name] = OracleMath
    return name


def write_hf_dir(d, bcfg, backbone):
    """HuggingFace-format backbone directory (config.json + model.safetensors + the fixture tokenizer), as config.models.*.llm.llm expects"""
    import shutil
    from safetensors.torch import save_file
    d = Path(d)
    d.mkdir(parents=True, exist_ok=True)
    (d / "config.json").write_text(json.dumps(bcfg))
    save_file({k: v.contiguous() for k, v in backbone.items()}, str(d / "model.safetensors"))
    shutil.copy(GOLDEN / "tokenizer.json", d / "tokenizer.json")
    (d / "tokenizer_config.json").write_text(json.dumps({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<|endoftext|>",
                                                         "eos_token": "<|endoftext|>"}))
    return str(d)
