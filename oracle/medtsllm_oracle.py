"""CPU restatement (plain torch, fp32) of the MedTsLLM forward hot path — the parity ORACLE.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg, and there only as the checker / reported CPU baseline. Never by the product.

Pinning: every function below is checked against golden vectors captured from the real
reference (flixpar/med-ts-llm @ 2024_10_08, imported on CPU with transformers 5.15.0,
torch 2.10) by tests/golden/make_golden.py -> tests/test_oracle_golden.py. The reference has no
tests of its own, so those goldens are the pin.

All arithmetic is differentiable torch, so `loss.backward()` on the oracle output gives the
oracle gradients (also pinned against the reference's autograd gradients in the goldens).

Citations are into /root/reference (R:) and HuggingFace transformers 5.15.0 (HF:).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- a1: RevIN
def revin_stats(x, eps=1e-5):
    """R:models/layers/RevIN.py:37-43 — per (b, c) mean and sqrt(biased var + eps) over time, detached."""
    mean = x.mean(dim=1, keepdim=True).detach()
    stdev = torch.sqrt(x.var(dim=1, keepdim=True, unbiased=False) + eps).detach()
    return mean, stdev


def revin_norm(x, mean, stdev):
    """R:models/layers/RevIN.py:45-56 (affine=False, R:models/medtsllm.py:91)."""
    return (x - mean) / stdev


def revin_denorm(y, mean, stdev):
    """R:models/layers/RevIN.py:58-69."""
    return y * stdev + mean


# ----------------------------------------------------------------------------- a2: patch index map
def n_patches_of(seq_len, patch_len, stride):
    """R:models/medtsllm.py:52."""
    return int((seq_len - patch_len) / stride + 2)


def patch_index_map(seq_len, patch_len, stride):
    """Integer map idx[p, j] = source time index of element j of patch p.

    R:models/layers/embed.py:160-163 (right replicate-pad by `stride` with the last value) and
    :190 (unfold size=patch_len step=stride) ==> idx = min(p*stride + j, L-1) for
    p < (L + stride - patch_len)//stride + 1.
    """
    P = (seq_len + stride - patch_len) // stride + 1
    p = torch.arange(P).unsqueeze(1)
    j = torch.arange(patch_len).unsqueeze(0)
    return torch.clamp(p * stride + j, max=seq_len - 1).to(torch.int32)


# ----------------------------------------------------------------------------- a3: token conv
def token_conv(patches, w):
    """R:models/layers/embed.py:44-46 — Conv1d(patch_len->d_model, k=3, circular pad over the PATCH axis, no bias).

    patches [N, P, patch_len], w [d_model, patch_len, 3]  ->  [N, P, d_model]
    out[n, p, :] = sum_k W[:, :, k] @ patches[n, (p + k - 1) mod P, :]
    """
    out = 0
    for k in range(3):
        shifted = torch.roll(patches, shifts=1 - k, dims=1)  # shifted[p] = patches[(p + k - 1) mod P]
        out = out + shifted @ w[:, :, k].t()
    return out


def patch_embed(x_norm, w, patch_len, stride):
    """R:models/medtsllm.py:272-273 + R:models/layers/embed.py:186-197 (dropout p=0).

    x_norm [B, L, C] -> [B*C, P, d_patch]
    """
    B, L, C = x_norm.shape
    idx = patch_index_map(L, patch_len, stride).long()          # [P, patch_len]
    xc = x_norm.permute(0, 2, 1).reshape(B * C, L)               # [B*C, L]
    patches = xc[:, idx]                                         # [B*C, P, patch_len]
    return token_conv(patches, w)


# ----------------------------------------------------------------------------- a5: mapping + reprogramming
def source_embeddings(word_emb, map_w, map_b):
    """R:models/medtsllm.py:281 — mapping_layer(word_embeddings^T)^T = map_w @ word_emb + map_b[:, None]."""
    return map_w @ word_emb + map_b.unsqueeze(1)


def reprogramming(x, source, p, n_heads, prefix="reprogramming_layer."):
    """R:models/medtsllm.py:566-591 (attention dropout p=0)."""
    B, Lq, _ = x.shape
    S = source.shape[0]
    H = n_heads
    q = F.linear(x, p[prefix + "query_projection.weight"], p[prefix + "query_projection.bias"]).view(B, Lq, H, -1)
    k = F.linear(source, p[prefix + "key_projection.weight"], p[prefix + "key_projection.bias"]).view(S, H, -1)
    v = F.linear(source, p[prefix + "value_projection.weight"], p[prefix + "value_projection.bias"]).view(S, H, -1)
    E = q.shape[-1]
    scores = torch.einsum("blhe,she->bhls", q, k)
    A = torch.softmax(scores / math.sqrt(E), dim=-1)
    out = torch.einsum("bhls,she->blhe", A, v).reshape(B, Lq, -1)
    return F.linear(out, p[prefix + "out_projection.weight"], p[prefix + "out_projection.bias"])


def encode_ts(x_enc, p, word_emb, m):
    """R:models/medtsllm.py:263-297. Returns (x_tokens [B', P', d_llm], (mean, stdev))."""
    B, L, C = x_enc.shape
    mean, stdev = revin_stats(x_enc)
    xn = revin_norm(x_enc, mean, stdev)
    enc = patch_embed(xn, p["patch_embedding.value_embedding.tokenConv.weight"], m["patch_len"], m["stride"])
    P = enc.shape[1]
    d_patch = enc.shape[2]
    cov = m["covariate_mode"]
    if cov == "concat":
        enc = enc.reshape(B, C, P, d_patch).permute(0, 2, 1, 3).reshape(B, P, C * d_patch)
    src = source_embeddings(word_emb, p["mapping_layer.weight"], p["mapping_layer.bias"])
    enc = reprogramming(enc, src, p, m["n_heads"])
    d_llm = enc.shape[-1]
    if cov == "add":
        enc = enc.reshape(B, C, P, d_llm).mean(dim=1)
    elif cov == "weighted-average":
        enc = enc.reshape(B, C, P, d_llm).permute(0, 2, 3, 1)
        enc = F.linear(enc, p["feature_weighting.weight"], p["feature_weighting.bias"]).squeeze(-1)
    elif cov == "interleave":
        enc = enc.reshape(B, C, -1, d_llm).permute(0, 2, 1, 3).reshape(B, -1, d_llm)
    return enc, (mean, stdev), src


# ----------------------------------------------------------------------------- a6: prompt embedding assembly
def prompt_embeddings(token_ids, embed_w, pad_token_id, encode_tensor=None):
    """R:models/medtsllm.py:299-319,331-337.

    token_ids: per sample, a list of parts, each tokenised separately: a list of ids, or ("examples" prompting,
    R:models/medtsllm.py:313-319) a tensor [1, L_ex, C] that goes through encode_ts (`encode_tensor`) and contributes
    its patch embeddings. Parts are concatenated, then LEFT-padded to the batch max with the pad(=eos) embedding; no
    attention mask.
    """
    def embed(part):
        if torch.is_tensor(part):
            return encode_tensor(part)[0]
        return embed_w[torch.tensor(list(part), dtype=torch.long)]
    seqs = [torch.cat([embed(part) for part in parts], dim=0) for parts in token_ids]
    max_len = max(s.shape[0] for s in seqs)
    pad = embed_w[pad_token_id]
    out = []
    for s in seqs:
        if s.shape[0] < max_len:
            s = torch.cat([pad.unsqueeze(0).expand(max_len - s.shape[0], -1), s], dim=0)
        out.append(s)
    return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------- a7: frozen backbones
def gelu_new(x):
    """HF:activations.py:65-66 (NewGELUActivation)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def causal_attention(q, k, v, scale, drop_mult=None):
    """Eager causal attention, softmax in fp32 — HF:models/gpt2/modeling_gpt2.py:54-72,
    HF:models/llama/modeling_llama.py:191-213. q,k,v [B, H, T, hd]. drop_mult: explicit attention-dropout multipliers
    [B, H, T, T] (keep / (1 - p), or 0), applied to the probabilities like nn.Dropout (HF gpt2 :65)."""
    T = q.shape[-2]
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.ones(T, T, dtype=torch.bool).tril()
    s = s.masked_fill(~mask, torch.finfo(s.dtype).min)
    a = torch.softmax(s.float(), dim=-1).to(q.dtype)
    if drop_mult is not None:
        a = a * drop_mult
    return a @ v


def gpt2_forward(h, w, cfg, masks=None):
    """HF:models/gpt2/modeling_gpt2.py:514-628 (GPT2Model.forward with inputs_embeds, eager attention).
    w: HF state-dict names ("wpe.weight", "h.0.ln_1.weight", ...). masks = None: all dropouts off (eval, or pdrop 0).
    masks = {"embd": [B,T,d], "attn": [L x [B,H,T,T]], "resid1": [L x [B,T,d]], "resid2": [L x [B,T,d]]} of explicit
    dropout multipliers (keep / (1-p) or 0) reproduces train mode deterministically: embd_pdrop after the position
    add (:579), attn_pdrop on the probabilities (:65), resid_pdrop after each c_proj (:243 / :330 and :397)."""
    B, T, d = h.shape
    H = cfg["n_head"]
    hd = d // H
    eps = cfg.get("layer_norm_epsilon", 1e-5)
    h = h + w["wpe.weight"][:T].unsqueeze(0)
    mk = masks or {}
    if "embd" in mk:
        h = h * mk["embd"]
    for i in range(cfg["n_layer"]):
        pre = f"h.{i}."
        x = F.layer_norm(h, (d,), w[pre + "ln_1.weight"], w[pre + "ln_1.bias"], eps)
        qkv = x @ w[pre + "attn.c_attn.weight"] + w[pre + "attn.c_attn.bias"]   # Conv1D: HF:pytorch_utils.py:117-121
        q, k, v = qkv.split(d, dim=-1)
        q = q.view(B, T, H, hd).transpose(1, 2)
        k = k.view(B, T, H, hd).transpose(1, 2)
        v = v.view(B, T, H, hd).transpose(1, 2)
        a = causal_attention(q, k, v, hd ** -0.5, mk["attn"][i] if "attn" in mk else None).transpose(1, 2).reshape(B, T, d)
        br = a @ w[pre + "attn.c_proj.weight"] + w[pre + "attn.c_proj.bias"]
        h = h + (br * mk["resid1"][i] if "resid1" in mk else br)
        x = F.layer_norm(h, (d,), w[pre + "ln_2.weight"], w[pre + "ln_2.bias"], eps)
        m = gelu_new(x @ w[pre + "mlp.c_fc.weight"] + w[pre + "mlp.c_fc.bias"])
        br = m @ w[pre + "mlp.c_proj.weight"] + w[pre + "mlp.c_proj.bias"]
        h = h + (br * mk["resid2"][i] if "resid2" in mk else br)
    return F.layer_norm(h, (d,), w["ln_f.weight"], w["ln_f.bias"], eps)


def rms_norm(x, weight, eps):
    """HF:models/llama/modeling_llama.py:62-67."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return weight * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_tables(T, hd, theta):
    """HF:models/llama/modeling_llama.py:113-127 — fp32 cos/sin [T, hd], emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    freqs = torch.arange(T).float().unsqueeze(1) * inv_freq.unsqueeze(0)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    """HF:models/llama/modeling_llama.py:130-134."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat([-x2, x1], dim=-1)


def llama_forward(h, w, cfg):
    """HF:models/llama/modeling_llama.py:367-417 (LlamaModel.forward with inputs_embeds, eager attention)."""
    B, T, d = h.shape
    H = cfg["num_attention_heads"]
    Hkv = cfg.get("num_key_value_heads", H)
    hd = cfg.get("head_dim") or d // H
    eps = cfg["rms_norm_eps"]
    cos, sin = rope_tables(T, hd, cfg["rope_theta"])
    cos, sin = cos.to(h.dtype), sin.to(h.dtype)          # HF:models/llama/modeling_llama.py:126-127 (tables in the hidden states' dtype)
    for i in range(cfg["num_hidden_layers"]):
        pre = f"layers.{i}."
        x = rms_norm(h, w[pre + "input_layernorm.weight"], eps)
        q = F.linear(x, w[pre + "self_attn.q_proj.weight"]).view(B, T, H, hd).transpose(1, 2)
        k = F.linear(x, w[pre + "self_attn.k_proj.weight"]).view(B, T, Hkv, hd).transpose(1, 2)
        v = F.linear(x, w[pre + "self_attn.v_proj.weight"]).view(B, T, Hkv, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        if Hkv != H:  # repeat_kv  HF:models/llama/modeling_llama.py:179-188
            k = k.repeat_interleave(H // Hkv, dim=1)
            v = v.repeat_interleave(H // Hkv, dim=1)
        a = causal_attention(q, k, v, hd ** -0.5).transpose(1, 2).reshape(B, T, H * hd)
        h = h + F.linear(a, w[pre + "self_attn.o_proj.weight"])
        x = rms_norm(h, w[pre + "post_attention_layernorm.weight"], eps)
        m = F.linear(F.silu(F.linear(x, w[pre + "mlp.gate_proj.weight"])) * F.linear(x, w[pre + "mlp.up_proj.weight"]),
                     w[pre + "mlp.down_proj.weight"])
        h = h + m
    return rms_norm(h, w["norm.weight"], eps)


def backbone_forward(h, w, cfg, masks=None):
    if cfg["model_type"] == "gpt2":
        return gpt2_forward(h, w, cfg, masks)
    if cfg["model_type"] == "llama":
        return llama_forward(h, w, cfg)
    raise ValueError(cfg["model_type"])


def backbone_embed_weight(w, cfg):
    return w["wte.weight"] if cfg["model_type"] == "gpt2" else w["embed_tokens.weight"]


def word_embeddings_of(w, cfg):
    """R:models/medtsllm.py:219-222 — alias of the input embedding table, or 100 000 linspace rows if V > 100 000."""
    we = backbone_embed_weight(w, cfg)
    if we.shape[0] > 100_000:
        inds = torch.linspace(0, we.shape[0] - 1, 100_000, dtype=torch.long)
        we = we[inds, :]
    return we


# ----------------------------------------------------------------------------- a8/a9: tail
def medtsllm_forward(x_enc, p, w, bcfg, m, token_ids=None, pad_token_id=0, training=True, word_emb=None,
                     return_intermediates=False):
    """R:models/medtsllm.py:248-261,321-384 — the full forward.

    p: trainable params (reference names); w: frozen backbone (HF names); bcfg: backbone config dict;
    m: dict(task, pred_len, patch_len, stride, n_heads, d_ff, covariate_mode, embedding_downsample_mode,
            n_outputs_per_step, [seg_mode]); token_ids: per-sample per-part prompt ids or None.
    """
    B, L, C = x_enc.shape
    if word_emb is None:
        word_emb = word_embeddings_of(w, bcfg)
    x_tok, (mean, stdev), src = encode_ts(x_enc, p, word_emb, m)
    cov = m["covariate_mode"]
    d_llm = x_tok.shape[-1]
    if token_ids is not None and len(token_ids[0]) > 0:
        enc_ex = lambda t: encode_ts(t.to(x_enc.dtype), p, word_emb, m)[0]       # [1, P_ex', d_llm] (R: encode_part -> encode_ts)
        prompt = prompt_embeddings(token_ids, backbone_embed_weight(w, bcfg), pad_token_id, enc_ex).to(x_tok.dtype)
    else:
        prompt = torch.zeros(B, 0, d_llm, dtype=x_enc.dtype)
    if cov in ("independent", "merge-end"):
        prompt = prompt.repeat_interleave(C, dim=0)
    enc = torch.cat([prompt, x_tok], dim=1)
    dec = backbone_forward(enc, w, bcfg)
    n_patches = x_tok.shape[1]
    dec = dec[:, -n_patches:, :]
    d_ff = m["d_ff"]
    down = m["embedding_downsample_mode"]
    if down == "truncate":
        dec = dec[:, :, :d_ff]
    elif down == "linear":
        dec = F.linear(dec, p["embedding_downsample_layer.weight"], p["embedding_downsample_layer.bias"])
    elif down == "average":
        dec = dec.reshape(dec.shape[0], n_patches, d_ff, -1).mean(dim=-1)
    else:
        raise ValueError(down)
    head_in = dec.permute(0, 2, 1).reshape(dec.shape[0], -1)          # feature index = f * P + p
    out = F.linear(head_in, p["output_projection.linear.weight"], p["output_projection.linear.bias"])
    pred_len, nops = m["pred_len"], m["n_outputs_per_step"]
    if cov == "independent":
        out = out.view(B, C, pred_len, nops).mean(dim=1)
    elif cov == "merge-end":
        out = out.view(B, C, pred_len, nops).permute(0, 2, 3, 1).reshape(B, pred_len, -1)
        out = F.linear(out, p["feature_weighting.weight"], p["feature_weighting.bias"])
    else:
        out = out.view(B, pred_len, nops)
    task = m["task"]
    if task in ("forecasting", "reconstruction", "anomaly_detection", "pretraining"):
        out = revin_denorm(out, mean, stdev)
    else:
        out = out.squeeze(-1)
    if not training:  # R:models/medtsllm.py:251-259
        if task == "semantic_segmentation":
            out = F.softmax(out, dim=-1) if m.get("n_classes", 0) > 2 else torch.sigmoid(out)
        elif task == "segmentation" and m.get("seg_mode") == "boundary-prediction":
            out = torch.sigmoid(out)
    if return_intermediates:
        return out, {"revin_mean": mean, "revin_stdev": stdev, "source_embeddings": src,
                     "reprog_tokens": x_tok, "llm_inputs_embeds": enc, "llm_last_hidden": None}
    return out


# ----------------------------------------------------------------------------- a6: input statistics
def calc_lags(x, n_lags=5):
    """R:models/medtsllm.py:530-538 — top-k of the channel-mean circular autocorrelation via rFFT."""
    x = x.permute(0, 2, 1).contiguous() if x.ndim == 3 else x.unsqueeze(1)
    f = torch.fft.rfft(x, dim=-1)
    corr = torch.fft.irfft(f * torch.conj(f), dim=-1)
    return torch.topk(corr.mean(dim=1), n_lags, dim=-1).indices
