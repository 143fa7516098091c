"""The REAL reference at the widths it is run at: tests/golden/rw_*.{npz,json} were captured by importing /root/reference
(tests/golden/make_realwidth_golden.py) with a GPT-2-small-width (two layers, and the full 12), a Llama-2-7B-width and a Llama-3-8B-width backbone (two
layers each), on the metric workload's window geometry ([L = 1024, C = 12], d_model 32, d_ff 128, 8 heads, 1024 prototypes, dataset + task prompt) plus
`independent` / `add` / `weighted-average` covariates, the `truncate` down-sample, input-statistics prompts and the segmentation / reconstruction heads. Every weight is formula-generated
(helpers.rw_backbone_state / rw_trainable_values — the generator imported the same functions), so the fixtures hold inputs, expected outputs,
sampled stage tensors and gradient summaries only.

Until round 4 the real-width comparisons (tests/test_gpu_realwidth.py) were HIP vs the ORACLE, and the oracle itself was pinned to the reference
only at d_llm 128: these tests close that gap.
  * CPU suite: the oracle against the reference on five of the seven fixtures (fp32, <= 2e-5 — the L1 rung of SURVEY.md 8c);
  * GPU suite: the HIP path against the reference on all seven (bar = 1.5 x the reference's own bf16-autocast deviation, exactly as
    tests/test_gpu_golden.py), and the oracle against the two largest fixtures on the GPU box's host cores (6 - 12 GB, minutes: too large for the CPU suite).
"""
import numpy as np
import pytest
import torch

from helpers import RW_CASES, RW_CASES_CPU, load_rw_case, oracle_mcfg, golden_loss, rel_err, abs_err, big_grad_summary, fixture_tokenizer

TOL = 2e-5         # fp32 reductions over K = 50 257 / 16 384 / 11 008 in two different summation orders


def _sample(meta, key, t):
    s = (meta.get("sampled") or {}).get(key)
    return t if s is None else torch.as_tensor(t).detach().float().flatten()[::s]


def _oracle_vs_reference(name):
    from oracle import medtsllm_oracle as O
    meta, data, bcfg, backbone, params = load_rw_case(name)
    m = oracle_mcfg(meta)
    p = {n: v.clone().requires_grad_(True) for n, v in params.items()}
    x = torch.from_numpy(data["x_enc"])
    mean, stdev = O.revin_stats(x)
    assert rel_err(mean, data["revin_mean"]) < 1e-6 and rel_err(stdev, data["revin_stdev"]) < 1e-6
    pe = O.patch_embed(O.revin_norm(x, mean, stdev), p["patch_embedding.value_embedding.tokenConv.weight"], meta["patch_len"], meta["stride"])
    assert rel_err(_sample(meta, "patch_embed_out", pe), data["patch_embed_out"]) < TOL
    we_kw = {"word_emb": p["word_embeddings"]} if "word_embeddings" in p else {}        # trainable table (vocabulary > 100 000)
    pred, inter = O.medtsllm_forward(x, p, backbone, bcfg, m, token_ids=meta["prompt_token_ids"], pad_token_id=meta["pad_token_id"],
                                     training=True, return_intermediates=True, **we_kw)
    assert inter["llm_inputs_embeds"].shape[1] == meta["T"]
    assert rel_err(_sample(meta, "llm_inputs_embeds", inter["llm_inputs_embeds"]), data["llm_inputs_embeds"]) < TOL
    with torch.no_grad():
        last = O.backbone_forward(inter["llm_inputs_embeds"], backbone, bcfg)[:, -meta["n_patches"]:, :]
        src = O.source_embeddings(p["word_embeddings"] if we_kw else O.word_embeddings_of(backbone, bcfg), p["mapping_layer.weight"], p["mapping_layer.bias"])
    assert rel_err(_sample(meta, "llm_last_hidden", last), data["llm_last_hidden"]) < TOL
    assert rel_err(_sample(meta, "source_embeddings", src), data["source_embeddings"]) < TOL
    assert rel_err(_sample(meta, "pred_train", pred), data["pred_train"]) < TOL
    loss = golden_loss(pred, data["target"], meta["task"])
    assert abs(loss.item() - float(data["loss"])) < 1e-5 * max(1.0, abs(float(data["loss"])))
    loss.backward()
    n_checked = 0
    for k, v in data.items():
        if k.startswith("grad."):
            n = k[len("grad."):]
            # (the key bias has an analytically-zero gradient — softmax shift invariance — hence the absolute floor)
            assert abs_err(p[n].grad, v) < 5e-5 * float(np.linalg.norm(v)) + 1e-7, (n, rel_err(p[n].grad, v))
            n_checked += 1
        elif k.startswith("gradnorm."):
            n = k[len("gradnorm."):]
            g = p[n].grad
            norm, prow, pcol, sample = big_grad_summary(g.reshape(g.shape[0], -1), meta["synth"]["stride"])
            assert abs(norm - float(v)) < 5e-5 * float(v), n
            for got, want in ((prow, data["gradproj_rows." + n]), (pcol, data["gradproj_cols." + n]), (sample, data["gradsample." + n])):
                assert abs_err(got, want) < 5e-5 * float(np.linalg.norm(want)) + 1e-7, (n, rel_err(got, want))
            n_checked += 1
    assert n_checked == len(p)
    with torch.no_grad():
        pe_eval = O.medtsllm_forward(x, p, backbone, bcfg, m, token_ids=meta["prompt_token_ids"], pad_token_id=meta["pad_token_id"], training=False, **we_kw)
    assert rel_err(pe_eval, data["pred_eval"]) < TOL


@pytest.mark.parametrize("name", RW_CASES_CPU)
def test_oracle_vs_reference_real_width(name):
    """GPT-2-small width: the metric model cut to two layers and at its full 12, `independent` covariates with input-statistics prompts, `add` with the
    `truncate` down-sample and the boundary-segmentation head; Llama-2-7B width (0.54 G backbone weights + a [1024, 32000] mapping layer: ~1 min, ~6 GB)"""
    _oracle_vs_reference(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in RW_CASES if n not in RW_CASES_CPU])
def test_oracle_vs_reference_real_width_large(name):
    """Llama-2-7B width with `weighted-average` covariates; Llama-3-8B width (GQA 32 / 8, ffn 14336, vocabulary 128 256 -> 100 000 TRAINABLE rows:
    1.1 G fp32 numbers with their gradients, ~3 min) — on the GPU box's host for their size only (no device code runs)"""
    _oracle_vs_reference(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RW_CASES)
def test_hip_vs_reference_real_width(name):
    from med_ts_llm_amd.models import model_lookup
    from test_gpu_golden import check_hip_vs_golden, _cfg_from_meta, _DS
    meta, data, bcfg, backbone, params = load_rw_case(name)
    model = model_lookup["medtsllm"](_cfg_from_meta(meta), _DS(meta), backbone_state=(bcfg, backbone))
    model.tokenizer = fixture_tokenizer()
    assert {n: tuple(q.shape) for n, q in model.named_parameters() if q.requires_grad} == {n: tuple(s) for n, s in meta["param_table"].items()}
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected and set(missing) <= {"word_embeddings"}, (missing, unexpected)
    del backbone, params
    check_hip_vs_golden(model.to("cuda"), meta, data, bcfg, name)
