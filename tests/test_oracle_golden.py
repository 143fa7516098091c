"""Pin the oracle (oracle/medtsllm_oracle.py) against golden vectors captured from the REAL reference.

L0 ints bit-exact; L1 fp32 forward / gradients <= 1e-5 norm-wise relative (SURVEY.md §8c ladder).
"""
import numpy as np
import pytest
import torch

from oracle import medtsllm_oracle as O
from helpers import CASES, load_case, oracle_mcfg, golden_loss, rel_err, abs_err, GOLDEN, prompt_parts_with_examples, big_grad_summary

TOL = 1e-5


@pytest.mark.parametrize("name", CASES)
def test_patch_index_map_bit_exact(name):
    meta, data, _, _ = load_case(name)
    idx = O.patch_index_map(meta["L"], meta["patch_len"], meta["stride"]).numpy()
    assert idx.dtype == np.int32
    assert idx.shape == data["patch_index_map"].shape
    assert np.array_equal(idx, data["patch_index_map"])
    # n_patches formula (R:models/medtsllm.py:52) agrees with the unfold count for these L
    assert O.n_patches_of(meta["L"], meta["patch_len"], meta["stride"]) == idx.shape[0]


@pytest.mark.parametrize("name", CASES)
def test_forward_backward_vs_reference(name):
    meta, data, bcfg, backbone = load_case(name)
    m = oracle_mcfg(meta)
    p = {k[len("param."):]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in data.items() if k.startswith("param.")}
    x = torch.from_numpy(data["x_enc"])
    tok = prompt_parts_with_examples(meta, data) if "examples" in data else meta["prompt_token_ids"]

    mean, stdev = O.revin_stats(x)
    assert rel_err(mean, data["revin_mean"]) < 1e-6
    assert rel_err(stdev, data["revin_stdev"]) < 1e-6

    pe = O.patch_embed(O.revin_norm(x, mean, stdev), p["patch_embedding.value_embedding.tokenConv.weight"],
                       meta["patch_len"], meta["stride"])
    assert rel_err(pe, data["patch_embed_out"]) < TOL

    we = O.word_embeddings_of(backbone, bcfg)
    src = O.source_embeddings(we, p["mapping_layer.weight"], p["mapping_layer.bias"])
    assert rel_err(src, data["source_embeddings"]) < TOL

    we_kw = {"word_emb": p["word_embeddings"]} if "word_embeddings" in p else {}      # trainable table (vocabulary > 100 000)
    pred, inter = O.medtsllm_forward(x, p, backbone, bcfg, m, token_ids=tok, pad_token_id=meta["pad_token_id"],
                                     training=True, return_intermediates=True, **we_kw)
    assert rel_err(inter["llm_inputs_embeds"], data["llm_inputs_embeds"]) < TOL
    assert rel_err(O.backbone_forward(inter["llm_inputs_embeds"], backbone, bcfg), data["llm_last_hidden"]) < TOL
    assert pred.shape == data["pred_train"].shape
    assert rel_err(pred, data["pred_train"]) < TOL

    loss = golden_loss(pred, data["target"], meta["task"])
    assert abs(loss.item() - float(data["loss"])) < 1e-5 * max(1.0, abs(float(data["loss"])))
    loss.backward()
    for k, v in data.items():
        if k.startswith("grad."):
            n = k[len("grad."):]
            g = p[n].grad
            assert g is not None, n
            # key_projection.bias has an analytically-zero gradient (softmax shift invariance): absolute floor
            assert abs_err(g, v) < 5e-5 * float(np.linalg.norm(v)) + 1e-7, (n, rel_err(g, v))

    # 100 000-wide gradients (vocabulary > 100 000 fixture): norm, projections along both axes and strided samples
    for k, v in data.items():
        if k.startswith("gradnorm."):
            n = k[len("gradnorm."):]
            assert p[n].grad is not None, n
            norm, prow, pcol, sample = big_grad_summary(p[n].grad, meta["synth"]["stride"])
            assert abs(norm - float(v)) < 5e-5 * float(v), n
            for got, want in ((prow, data["gradproj_rows." + n]), (pcol, data["gradproj_cols." + n]), (sample, data["gradsample." + n])):
                assert abs_err(got, want) < 5e-5 * float(np.linalg.norm(want)) + 1e-7, (n, rel_err(got, want))

    with torch.no_grad():
        pe_eval = O.medtsllm_forward(x, p, backbone, bcfg, m, token_ids=tok, pad_token_id=meta["pad_token_id"], training=False, **we_kw)
    assert rel_err(pe_eval, data["pred_eval"]) < TOL


def test_calc_lags_ints_exact():
    z = np.load(GOLDEN / "stats.npz")
    x = torch.from_numpy(z["x"])
    assert np.array_equal(O.calc_lags(x, 5).numpy(), z["lags_3d"])
    assert np.array_equal(O.calc_lags(x[:, :, 1], 5).numpy(), z["lags_2d"])
