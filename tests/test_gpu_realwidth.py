"""Oracle parity at the widths the reference actually runs (VERDICT r03 "missing 3 / 4"): every other end-to-end oracle / golden comparison is at
d_llm <= 256, ffn <= 384; at BASELINE widths there were only property tests, which cannot see a wrong-but-deterministic 256 x 256-tile epilogue
or a K = 11008 accumulation drift end to end.

Here the fp32 oracle runs on the host cores of the GPU box at the REAL width with the stack cut to two layers (`llm_layers`, which the
reference itself offers: R:models/medtsllm.py:145-146) — 1 to 10 s per pass — and the HIP path is compared with it exactly as the small cases of
tests/test_gpu_model.py are: forward, loss, EVERY trainable gradient, the eval output; bar = 1.5 x the oracle's own deviation under bf16
autocast (= the reference's dtype "mixed" arithmetic) on the same model, small tensors 3 x, cancellation-prone sums pinned exactly.

  * Llama-2-7B width (d 4096, 32 heads of 128, ffn 11008, vocabulary 32000) — 256 x 256 GEMM tiles, SwiGLU / dSwiGLU epilogues, resident hd-128
    attention, the 16384 -> 4096 flatten head — on the metric-shaped [L = 1024, C = 12] windows;
  * Llama-3-8B width (GQA 32 / 8, ffn 14336, vocabulary 128 256 -> 100 000 TRAINABLE sub-sampled rows);
  * GPT-2-small at its full depth (12 layers of 768);
  * BASELINE.json configs[3]'s PSM geometry (25 channels, L = 2048, the 1.68 G-parameter flatten head) and the `interleave` long-sequence mode
    (T = 1536 + prompt: the 32-rows-per-wave forward attention, chunked backward) at the Llama-2-7B width;
  * the four shipped reference configurations at their own hyper-parameters (R:configs/datasets/ludb.toml:6-7,35-44,
    bidmc.toml:6-7,36-45, ecgmit-anom.toml, ecgmit-seg.toml; the toml files do not travel, the numbers are typed in), on the Llama-2-7B width.
"""
import pytest
import torch

from helpers import FakeDataset, LONGT_GRAD_FACTOR
from test_gpu_model import _check_full_model

pytestmark = pytest.mark.gpu

LLAMA2_7B_2L = {"model_type": "llama", "vocab_size": 32000, "hidden_size": 4096, "intermediate_size": 11008, "num_hidden_layers": 2,
                "num_attention_heads": 32, "num_key_value_heads": 32, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
LLAMA3_8B_2L = {"model_type": "llama", "vocab_size": 128256, "hidden_size": 4096, "intermediate_size": 14336, "num_hidden_layers": 2,
                "num_attention_heads": 32, "num_key_value_heads": 8, "rms_norm_eps": 1e-5, "rope_theta": 500000.0}
GPT2_SMALL = {"model_type": "gpt2", "vocab_size": 50257, "n_positions": 1024, "n_embd": 768, "n_layer": 12, "n_head": 12,
              "layer_norm_epsilon": 1e-5, "embd_pdrop": 0.0, "attn_pdrop": 0.0, "resid_pdrop": 0.0}
_STATE = {}


def _state(name, hf):
    """seeded random-init CPU fp32 weights of the architecture (std 0.02, the scale of released checkpoints), built once per module"""
    if name not in _STATE:
        from med_ts_llm_amd.models.backbone import random_state_dict
        _STATE.clear()                                     # (one multi-GB state at a time)
        _STATE[name] = random_state_dict(hf, seed=0, std=0.02)
    return _STATE[name]


SHIPPED_PROMPTS = {"dataset": True, "task": True, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}


def test_llama2_7b_width_metric_windows_vs_oracle():
    """BASELINE.json configs[2] geometry: [L = 1024, C = 12] windows, 4-class semantic segmentation, concat covariates (query width 12 * 32 = 384),
    d_ff 128 x 8 heads, 1024 prototypes, dataset + task text prompt; B = 2"""
    _check_full_model("llama2_7b", "semantic_segmentation", 2, 1024, 12, 1024, "concat", "linear", True, d_model=32, d_ff=128, H=8, num_tokens=1024,
                      hf=LLAMA2_7B_2L, sd=_state("llama2", LLAMA2_7B_2L), prompting=SHIPPED_PROMPTS)


def test_llama2_7b_width_forecasting_no_prompt_vs_oracle():
    """the same width without a text prompt (T = P: no prompt rows at all) and a forecasting head with RevIN de-normalisation"""
    off = dict(SHIPPED_PROMPTS, dataset=False, task=False)
    _check_full_model("llama2_7b", "forecasting", 2, 512, 7, 96, "concat", "linear", False, d_model=32, d_ff=128, H=8, num_tokens=1024,
                      hf=LLAMA2_7B_2L, sd=_state("llama2", LLAMA2_7B_2L), prompting=off)


# R:configs/datasets/*.toml — (task, L = pred, C, covariate mode, d_ff, clip prompt)
SHIPPED = {
    "ludb": ("semantic_segmentation", 512, 1, "univariate", 128, False),          # ludb.toml:6-7,35-44 (one ECG lead per sample)
    "bidmc": ("segmentation", 256, 3, "concat", 64, False),                        # bidmc.toml:6-7,36-45 (ECG, PPG, respiration)
    "ecgmit-anom": ("anomaly_detection", 128, 2, "concat", 64, False),             # ecgmit-anom.toml (two-channel ambulatory ECG)
    "ecgmit-seg": ("segmentation", 256, 2, "concat", 64, True),                    # ecgmit-seg.toml: per-clip descriptions in the prompt
}


@pytest.mark.parametrize("name", list(SHIPPED))
def test_shipped_reference_configuration_vs_oracle(name):
    task, L, C, cov, d_ff, clip = SHIPPED[name]
    B = 4
    prompting = dict(SHIPPED_PROMPTS, clip=clip)
    desc = None
    if clip:      # per-sample prompts of different token counts: left padding, no prompt-row cache
        desc = [f"Record {100 + 7 * i} of a {30 + 11 * i} year old patient, leads MLII and V{1 + i}." for i in range(B)]
    _check_full_model("llama2_7b", task, B, L, C, L, cov, "linear", True, d_model=32, d_ff=d_ff, H=8, num_tokens=1024,
                      hf=LLAMA2_7B_2L, sd=_state("llama2", LLAMA2_7B_2L), prompting=prompting, descriptions=desc,
                      dataset=FakeDataset(C, 4 if task == "semantic_segmentation" else 0))


def test_psm_geometry_llama2_7b_width_vs_oracle():
    """BASELINE.json configs[3] geometry: PSM anomaly detection = reconstruction of 25-channel L = 2048 windows — concat width 25 * 32 = 800 (padded to 832 for
    the query GEMM), P = 256 patch rows (T = 256 + prompt: the resident hd-128 attention no longer fits, the chunked / 32-row kernels run), and the
    1.68 G-parameter flatten head (128 * 256 = 32768 -> 2048 * 25 = 51200: every fp32 tensor of it is > 2^31 bytes) with its gradient against the oracle's"""
    _check_full_model("llama2_7b", "anomaly_detection", 2, 2048, 25, 2048, "concat", "linear", True, d_model=32, d_ff=128, H=8, num_tokens=1024,
                      hf=LLAMA2_7B_2L, sd=_state("llama2", LLAMA2_7B_2L), prompting=SHIPPED_PROMPTS)


def test_interleave_covariates_llama2_7b_width_vs_oracle():
    """SURVEY 8f-4 at the real width: `interleave` covariates put every channel's patches into the LLM sequence — 12 * 128 = 1536 patch rows + the prompt:
    the long-sequence attention kernels (32 query rows per wave forward, chunked backward, XCD-aware launch) inside the full model, B = 1"""
    _check_full_model("llama2_7b", "semantic_segmentation", 1, 1024, 12, 1024, "interleave", "linear", True, d_model=32, d_ff=128, H=8, num_tokens=1024,
                      hf=LLAMA2_7B_2L, sd=_state("llama2", LLAMA2_7B_2L), prompting=SHIPPED_PROMPTS, grad_bar=LONGT_GRAD_FACTOR)


def test_llama3_8b_width_trainable_vocabulary_vs_oracle():
    """BASELINE.json configs[4] geometry: GQA 32 / 8 at hd 128, ffn 14336, vocabulary 128 256 -> the 100 000 linspace-sampled rows are a TRAINABLE
    parameter (R:models/medtsllm.py:220-222): their gradient and the [1024, 100 000] mapping gradient against the oracle's"""
    _check_full_model("llama3_8b", "reconstruction", 2, 1024, 12, 1024, "concat", "linear", False, d_model=32, d_ff=128, H=8, num_tokens=1024,
                      hf=LLAMA3_8B_2L, sd=_state("llama3", LLAMA3_8B_2L), prompting=dict(SHIPPED_PROMPTS, dataset=False, task=False))


def test_gpt2_small_full_depth_metric_windows_vs_oracle():
    """the metric workload's model at its full 12 layers (T = prompt + 128 patches), B = 2, forecasting pred 96"""
    _check_full_model("gpt2s", "forecasting", 2, 1024, 12, 96, "concat", "linear", True, d_model=32, d_ff=128, H=8, num_tokens=1024,
                      hf=GPT2_SMALL, sd=_state("gpt2s", GPT2_SMALL), prompting=SHIPPED_PROMPTS)
