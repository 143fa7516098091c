"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, through the C-ABI, against the oracle /
a plain torch fp32 statement of the same op on the SAME bf16-rounded inputs.

Tolerances (norm-wise relative; SURVEY.md 8c ladder L2, <= 1e-3-class per rounding): fp32 outputs of bf16-input GEMMs 2e-5
(accumulation order only); bf16 outputs of one fp32 computation 2e-3 (ONE bf16 rounding: at most 2^-9 = 1.95e-3 per element,
1.1-1.6e-3 norm-wise); attention forward 3e-3 (the probabilities are rounded to bf16 before P.V, then the output is); attention
backward 5e-3 (bf16 P, dS, dO operands and the bf16-rounded O inside delta: three to four roundings); integer maps bit-exact.
Every comparison's measured error is logged per source line to gpurun_out/kernel_parity_floor.json (largest value per line),
which is how the bars above were set (profiles/r02_kernel_parity_floor.json).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import json
import os
import sys

import helpers
from helpers import GOLDEN, load_case

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32
TOL_F32, TOL_BF16, TOL_ATTN_FWD, TOL_ATTN_BWD = 2e-5, 2e-3, 3e-3, 5e-3
# outputs that exist BEFORE the bf16 output rounding: fp32 GEMM C (TOL_F32) and the softmax statistics (measured 5e-8) meet
# north_star's 1e-3 with orders of margin. The fp32 attention OUTPUT does not quite: the probabilities are a bf16 MFMA operand of P.V (the design), and
# their 2^-9 roundings do not average out over the keys — measured 1.25e-3 (32 keys) .. 1.45e-3 (1024 keys); bar 1.6e-3, said here instead of hidden
# behind the output rounding
TOL_PRE, TOL_PRE_ATTN = 2e-5, 1.6e-3

_FLOOR = {}


def rel_err(a, b):
    """helpers.rel_err + a log of the largest error seen per calling line"""
    e = helpers.rel_err(a, b)
    f = sys._getframe(1)
    key = f"{f.f_code.co_name}:{f.f_lineno}"
    _FLOOR[key] = max(_FLOOR.get(key, 0.0), e)
    return e


@pytest.fixture(scope="module", autouse=True)
def _dump_floor():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "kernel_parity_floor.json"), "w") as fh:
            json.dump(dict(sorted(_FLOOR.items())), fh, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no ROCm device is visible")
    from med_ts_llm_amd.hip import ops as o
    return o


def rb(t):
    """round to bf16 and back (what the kernels see)"""
    return t.to(BF16).float()


def dev(t):
    return t.to("cuda")


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (100, 72, 192), (8192, 768, 768), (33, 50257 // 16, 64), (512, 130, 256)])
def test_gemm_store(ops, M, N, K):
    A = torch.randn(M, K, generator=g(1)).to(BF16)
    B = torch.randn(N, K, generator=g(2)).to(BF16)
    bias = torch.randn(N, generator=g(3))
    ref = A.float() @ B.float().t() + bias
    c32 = ops.gemm_nt(dev(A), dev(B), out_dtype=F32, bias=dev(bias))
    assert rel_err(c32, ref) < TOL_F32
    c16 = ops.gemm_nt(dev(A), dev(B), out_dtype=BF16, bias=dev(bias))
    assert rel_err(c16.float(), ref) < TOL_BF16
    # transpose detection: asymmetric operands, exact small-integer arithmetic
    Ai = torch.randint(-3, 4, (M, K), generator=g(4)).float()
    Bi = torch.randint(-3, 4, (N, K), generator=g(5)).float()
    ci = ops.gemm_nt(dev(Ai.to(BF16)), dev(Bi.to(BF16)), out_dtype=F32)
    assert torch.equal(ci.cpu(), Ai @ Bi.t())


def test_gemm_epilogues(ops):
    from med_ts_llm_amd.hip import _native as Nn
    from oracle.medtsllm_oracle import gelu_new
    M, N, K = 200, 192, 128
    A = torch.randn(M, K, generator=g(1)).to(BF16)
    B = (0.1 * torch.randn(N, K, generator=g(2))).to(BF16)
    bias = torch.randn(N, generator=g(3))
    v = A.float() @ B.float().t() + bias
    # GELU: aux_out = pre-activation (bf16), C = gelu_new(bf16(pre))
    aux = torch.empty(M, N, dtype=BF16, device="cuda")
    c = ops.gemm_nt(dev(A), dev(B), bias=dev(bias), epilogue=Nn.EPI_GELU, aux_out=aux)
    assert rel_err(aux.float(), v) < TOL_BF16
    assert rel_err(c.float(), gelu_new(rb(v))) < TOL_BF16
    # RESID: C(f32) = resid + bf16(v)
    resid = torch.randn(M, N, generator=g(4))
    c = ops.gemm_nt(dev(A), dev(B), out_dtype=F32, bias=dev(bias), epilogue=Nn.EPI_RESID, aux_in=dev(resid))
    assert rel_err(c, resid + v) < 2e-3
    # DGELU: C = v * gelu_new'(h)
    h = torch.randn(M, N, generator=g(5)).to(BF16)
    hf = h.float().requires_grad_(True)
    gelu_new(hf).sum().backward()
    c = ops.gemm_nt(dev(A), dev(B), epilogue=Nn.EPI_DGELU, aux_in=dev(h))
    assert rel_err(c.float(), (A.float() @ B.float().t()) * hf.grad) < TOL_BF16
    # ACCUM
    base = torch.randn(M, N, generator=g(6))
    out = dev(base).clone()
    ops.gemm_nt(dev(A), dev(B), out=out, epilogue=Nn.EPI_ACCUM)
    assert rel_err(out, base + A.float() @ B.float().t()) < TOL_F32


def test_gemm_splitk_and_row_gather(ops):
    M, N, K = 256, 128, 64 * 37
    A = torch.randn(M, K, generator=g(1)).to(BF16)
    B = (0.05 * torch.randn(N, K, generator=g(2))).to(BF16)
    ref = A.float() @ B.float().t()
    for sk in (2, 5, 8):
        c = ops.gemm_nt(dev(A), dev(B), out_dtype=F32, split_k=sk)
        assert rel_err(c, ref) < TOL_F32
    # "last P tokens of every sample" gather on A rows: [Bt, T, K] -> rows (b, T-P+p)
    Bt, T, Pp, K2 = 3, 10, 4, 64
    X = torch.randn(Bt * T, K2, generator=g(3)).to(BF16)
    W = torch.randn(96, K2, generator=g(4)).to(BF16)
    c = ops.gemm_nt(dev(X), dev(W), out_dtype=F32, M=Bt * Pp, a_rows=(Pp, T, T - Pp))
    ref = X.float().view(Bt, T, K2)[:, -Pp:, :].reshape(Bt * Pp, K2) @ W.float().t()
    assert rel_err(c, ref) < TOL_F32


# ------------------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, Hq, Hkv, D, scale, causal):
    """q [B,Tq,Hq*D], k/v [B,Tk,Hkv*D] fp32 -> (o, grads fn)"""
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    qh = q.view(B, Tq, Hq, D).transpose(1, 2)
    kh = k.view(B, Tk, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    vh = v.view(B, Tk, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool).tril(), float("-inf"))
    a = torch.softmax(s, dim=-1)
    return (a @ vh).transpose(1, 2).reshape(B, Tq, Hq * D)


@pytest.mark.parametrize("B,T,Hq,Hkv,D,causal", [(2, 64, 2, 2, 64, True), (2, 173, 4, 4, 64, True), (3, 256, 4, 2, 128, True),
                                                 (2, 100, 2, 2, 32, True), (2, 96, 4, 1, 64, False),
                                                 # hd 64, T = 512: chunked forward (the resident one's wider row padding no longer fits) + resident two-launch backward
                                                 (2, 512, 2, 2, 64, True)])
def test_attention_self(ops, B, T, Hq, Hkv, D, causal):
    q = torch.randn(B, T, Hq * D, generator=g(1)).to(BF16)
    k = torch.randn(B, T, Hkv * D, generator=g(2)).to(BF16)
    v = torch.randn(B, T, Hkv * D, generator=g(3)).to(BF16)
    do = torch.randn(B, T, Hq * D, generator=g(4)).to(BF16)
    scale = 1.0 / math.sqrt(D)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(qf, kf, vf, Hq, Hkv, D, scale, causal)
    ref.backward(do.float())
    o, lse = ops.attention_fwd(dev(q), dev(k), dev(v), Hq, Hkv, D, scale, causal)
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD
    # the softmax statistic is an fp32 output with no bf16 rounding on its path (fp32-accumulated scores of the bf16 inputs): north_star's
    # "1e-3 before output rounding" holds there with two orders of margin (exp2-domain arithmetic: ~1e-6)
    with torch.no_grad():
        sc = torch.einsum("bqhd,bkhd->bhqk", qf.view(B, T, Hq, D), kf.view(B, T, Hkv, D).repeat_interleave(Hq // Hkv, dim=2)) * scale
        if causal:
            sc = sc.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf"))
        assert rel_err(lse, torch.logsumexp(sc.double(), dim=-1)) < TOL_PRE
    dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), o, lse, dev(do), Hq, Hkv, D, scale, causal)
    # backward consumes the bf16-rounded O (for delta) and bf16 P/dS operands
    assert rel_err(dq.float(), qf.grad) < TOL_ATTN_BWD
    assert rel_err(dk.float(), kf.grad) < TOL_ATTN_BWD
    assert rel_err(dv.float(), vf.grad) < TOL_ATTN_BWD


@pytest.mark.parametrize("B,T,Hq,Hkv,D", [(2, 173, 4, 4, 64), (2, 256, 4, 2, 128), (1, 40, 2, 1, 64)])
def test_attention_chunked_equals_resident(ops, B, T, Hq, Hkv, D):
    """the chunked (64-key LDS tiles) and the resident (whole head in LDS) causal kernels are two schedules of one algorithm"""
    from med_ts_llm_amd.hip import _native
    q = torch.randn(B, T, Hq * D, generator=g(1)).to(BF16).cuda()
    k = torch.randn(B, T, Hkv * D, generator=g(2)).to(BF16).cuda()
    v = torch.randn(B, T, Hkv * D, generator=g(3)).to(BF16).cuda()
    do = torch.randn(B, T, Hq * D, generator=g(4)).to(BF16).cuda()
    scale = 1.0 / math.sqrt(D)
    res = {}
    for mode in (1, 0):
        with ops.attention_tune(resident=bool(mode)):        # per-call field of the argument struct: the library keeps no such switch
            o, lse = ops.attention_fwd(q, k, v, Hq, Hkv, D, scale, True)
            res[mode] = (o, lse) + ops.attention_bwd(q, k, v, o, lse, do, Hq, Hkv, D, scale, True)
    for a, b in zip(res[0], res[1]):
        assert rel_err(a.float(), b.float()) < 5e-3
    ref = ref_attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), Hq, Hkv, D, scale, True)
    assert rel_err(res[0][0].float(), ref) < TOL_ATTN_FWD and rel_err(res[1][0].float(), ref) < TOL_ATTN_FWD


@pytest.mark.parametrize("B,T,H,Hkv,D,resident", [(2, 96, 2, 2, 64, 1), (2, 200, 4, 2, 128, 1), (2, 96, 2, 1, 64, 0), (1, 256, 2, 2, 128, 0)])
def test_attention_backward_fused_inverse_rope(ops, B, T, H, Hkv, D, resident):
    """the inverse rotary embedding of dq / dk in the attention backward's store epilogues == mtl_rope_inplace(inverse) on the stored
    gradients, bit for bit (resident and chunked kernels, GQA, hd 64 / 128); dv is untouched"""
    q, k, v = (torch.randn(B, T, h_ * D, generator=g(i)).to(BF16).cuda() for i, h_ in ((1, H), (2, Hkv), (3, Hkv)))
    do = torch.randn(B, T, H * D, generator=g(4)).to(BF16).cuda()
    pos = torch.arange(T).float()[:, None]
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.cat([pos * inv, pos * inv], dim=1)
    cos, sin = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda()
    scale = 1.0 / math.sqrt(D)
    with ops.attention_tune(resident=bool(resident)):
        o, lse = ops.attention_fwd(q, k, v, H, Hkv, D, scale, True)
        dq0, dk0, dv0 = ops.attention_bwd(q, k, v, o, lse, do, H, Hkv, D, scale, True)
        dq1, dk1, dv1 = ops.attention_bwd(q, k, v, o, lse, do, H, Hkv, D, scale, True, rope=(cos, sin))
    want_q = ops.rope_inplace(dq0.reshape(B * T, H * D).clone(), cos, sin, T, H, D, inverse=True).view_as(dq0)
    want_k = ops.rope_inplace(dk0.reshape(B * T, Hkv * D).clone(), cos, sin, T, Hkv, D, inverse=True).view_as(dk0)
    assert torch.equal(dq1, want_q) and torch.equal(dk1, want_k) and torch.equal(dv1, dv0)
    assert not torch.equal(dq1, dq0)


@pytest.mark.parametrize("B,T,Tq,Hq,Hkv,D", [(1, 700, 700, 2, 2, 128), (2, 1000, 870, 4, 2, 64), (1, 1664, 1536, 2, 1, 128), (1, 515, 515, 2, 2, 64),
                                             # round 5: hd 128 takes the 32-row forward / chunked dQ (GQA or > 128 key rows: also dK/dV) from 128 query rows on, also
                                             # where K / V would fit the LDS — the cached Llama shape, ragged sizes below 256 rows, MHA with <= 128 key rows (resident dK/dV)
                                             (2, 256, 128, 4, 4, 128), (2, 200, 130, 4, 2, 128), (1, 150, 150, 2, 1, 128), (1, 300, 129, 2, 2, 128)])
def test_attention_long_causal_ragged(ops, B, T, Tq, Hq, Hkv, D):
    """the long-sequence kernels (32 query rows per wave on 32x32x16 MFMAs for the forward and dQ, chunked dK / dV, XCD-aware 1-D launch) on shapes that
    are NOT multiples of their 128-row workgroups / 64-key chunks, with the queries being the LAST Tq rows of a T-key sequence (prompt-row cache,
    pruned backward: causal_off = kv_row0 = T - Tq), MHA and GQA, hd 64 / 128 — against fp32 math; and the inverse rotary embedding fused into the dQ / dK
    store epilogues == mtl_rope_inplace(inverse) on the plain gradients, bit for bit"""
    off = T - Tq
    q = torch.randn(B, Tq, Hq * D, generator=g(1)).to(BF16)
    k = torch.randn(B, T, Hkv * D, generator=g(2)).to(BF16)
    v = torch.randn(B, T, Hkv * D, generator=g(3)).to(BF16)
    do = torch.randn(B, Tq, Hq * D, generator=g(4)).to(BF16)
    scale = 1.0 / math.sqrt(D)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    qh = qf.view(B, Tq, Hq, D).transpose(1, 2)
    kh = kf.view(B, T, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    vh = vf.view(B, T, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    sc = (qh @ kh.transpose(-1, -2)) * scale
    vis = torch.arange(T)[None, :] <= (torch.arange(Tq)[:, None] + off)          # key k visible to query i iff k <= i + off
    ref = (torch.softmax(sc.masked_fill(~vis, float("-inf")), dim=-1) @ vh).transpose(1, 2).reshape(B, Tq, Hq * D)
    ref.backward(do.float())
    o, lse = ops.attention_fwd(dev(q), dev(k), dev(v), Hq, Hkv, D, scale, True, causal_off=off)
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD
    lse_ref = torch.logsumexp(sc.masked_fill(~vis, float("-inf")), dim=-1)
    assert rel_err(lse, lse_ref) < 1e-4
    dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), o, lse, dev(do), Hq, Hkv, D, scale, True, causal_off=off, kv_row0=off)
    assert rel_err(dq.float(), qf.grad) < TOL_ATTN_BWD
    assert rel_err(dk.float()[:, off:], kf.grad[:, off:]) < TOL_ATTN_BWD and rel_err(dv.float()[:, off:], vf.grad[:, off:]) < TOL_ATTN_BWD
    # fused inverse RoPE (query row i sits at position i + off, key row j at position j)
    pos = torch.arange(T).float()[:, None]
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.cat([pos * inv, pos * inv], dim=1)
    cos, sin = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda()
    dq1, dk1, dv1 = ops.attention_bwd(dev(q), dev(k), dev(v), o, lse, dev(do), Hq, Hkv, D, scale, True, causal_off=off, kv_row0=off, rope=(cos, sin))
    want_q = ops.rope_inplace(dq.reshape(B * Tq, Hq * D).clone(), cos[off:].contiguous(), sin[off:].contiguous(), Tq, Hq, D, inverse=True).view_as(dq)
    want_k = ops.rope_inplace(dk.reshape(B * T, Hkv * D).clone(), cos, sin, T, Hkv, D, inverse=True).view_as(dk)
    assert torch.equal(dq1, want_q)
    assert torch.equal(dk1[:, off:], want_k[:, off:]) and torch.equal(dv1[:, off:], dv[:, off:])


def test_attention_rejects_negative_causal_offset(ops):
    """causal_off < 0 would leave some query rows without any visible key in their first chunk (the 32-row kernel then averages masked keys in:
    ADVICE r04): the C-ABI refuses it instead, forward and backward"""
    from med_ts_llm_amd.hip._native import MtlError
    B, T, H, D = 1, 320, 2, 64
    q, k, v = (dev(torch.randn(B, T, H * D, generator=g(i)).to(BF16)) for i in (1, 2, 3))
    with pytest.raises(MtlError):
        ops.attention_fwd(q, k, v, H, H, D, 0.125, True, causal_off=-64)
    o, lse = ops.attention_fwd(q, k, v, H, H, D, 0.125, True)
    with pytest.raises(MtlError):
        ops.attention_bwd(q, k, v, o, lse, q, H, H, D, 0.125, True, causal_off=-64)


def test_attention_fused_qkv_views(ops):
    """strided q/k/v views into one fused [B,T,(Hq+2Hkv)*D] buffer, as the backbone uses them"""
    B, T, Hq, Hkv, D = 2, 80, 4, 2, 64
    qkv = torch.randn(B, T, (Hq + 2 * Hkv) * D, generator=g(1)).to(BF16)
    dq_ = dev(qkv)
    q, k, v = dq_[..., : Hq * D], dq_[..., Hq * D: (Hq + Hkv) * D], dq_[..., (Hq + Hkv) * D:]
    scale = 1.0 / math.sqrt(D)
    o, _ = ops.attention_fwd(q, k, v, Hq, Hkv, D, scale, True)
    ref = ref_attention(qkv[..., : Hq * D].float(), qkv[..., Hq * D: (Hq + Hkv) * D].float(), qkv[..., (Hq + Hkv) * D:].float(),
                        Hq, Hkv, D, scale, True)
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD


@pytest.mark.parametrize("B,L,S,H,E", [(2, 8, 32, 2, 32), (3, 20, 1024, 8, 128), (2, 128, 200, 2, 64)])
def test_attention_cross_shared_kv(ops, B, L, S, H, E):
    """reprogramming attention: K/V [S, H*E] shared by every sample; dK/dV are summed over the batch"""
    q = torch.randn(B, L, H * E, generator=g(1)).to(BF16)
    k = torch.randn(S, H * E, generator=g(2)).to(BF16)
    v = torch.randn(S, H * E, generator=g(3)).to(BF16)
    do = torch.randn(B, L, H * E, generator=g(4)).to(BF16)
    scale = 1.0 / math.sqrt(E)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    scores = torch.einsum("blhe,she->bhls", qf.view(B, L, H, E), kf.view(S, H, E))
    A = torch.softmax(scale * scores, dim=-1)
    ref = torch.einsum("bhls,she->blhe", A, vf.view(S, H, E)).reshape(B, L, H * E)
    ref.backward(do.float())
    o, lse = ops.attention_fwd(dev(q), dev(k), dev(v), H, H, E, scale, False, shared_kv=True)
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD
    dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), o, lse, dev(do), H, H, E, scale, False, shared_kv=True)
    assert rel_err(dq.float(), qf.grad) < TOL_ATTN_BWD
    assert rel_err(dk.float(), kf.grad) < TOL_ATTN_BWD
    assert rel_err(dv.float(), vf.grad) < TOL_ATTN_BWD
    # delta from the fp32 forward output (the route the model takes: no second pass over K / V in the dQ kernel)
    o2, lse2, o32 = ops.attention_fwd(dev(q), dev(k), dev(v), H, H, E, scale, False, shared_kv=True, want_o32=True)
    assert torch.equal(o2, o) and torch.equal(o32.to(BF16), o)
    # the PRE-rounding output: what is left in it is the bf16 rounding of the probabilities in front of P.V; the statistic has no rounding at all
    assert rel_err(o32, ref.detach()) < TOL_PRE_ATTN
    assert rel_err(lse2, torch.logsumexp((scale * scores.detach()).double(), dim=-1)) < TOL_PRE
    dq2, dk2, dv2 = ops.attention_bwd(dev(q), dev(k), dev(v), o2, lse2, dev(do), H, H, E, scale, False, shared_kv=True, o32=o32)
    assert rel_err(dq2.float(), qf.grad) < TOL_ATTN_BWD
    assert rel_err(dk2.float(), kf.grad) < TOL_ATTN_BWD and rel_err(dv2.float(), vf.grad) < TOL_ATTN_BWD
    # sum_s dS_qs = 0: the violation (per query, relative to sum_s |dS_qs| ~ |dQ| scale) must not be worse than the two-pass route's
    assert rel_err(dq2.float(), qf.grad) < 1.5 * rel_err(dq.float(), qf.grad) + 5e-4


@pytest.mark.parametrize("pdrop", [0.1, 0.5])
def test_attention_cross_dropout(ops, pdrop):
    """A = dropout(softmax(.)) (R:models/medtsllm.py:588): the kernel's counter-based keep mask is replicated on the host,
    so forward and all three gradients are compared exactly like the no-dropout case; keep rate ~ 1 - p."""
    B, L, S, H, E, seed = 2, 40, 200, 2, 64, 12345
    q = torch.randn(B, L, H * E, generator=g(1)).to(BF16)
    k = torch.randn(S, H * E, generator=g(2)).to(BF16)
    v = torch.randn(S, H * E, generator=g(3)).to(BF16)
    do = torch.randn(B, L, H * E, generator=g(4)).to(BF16)
    scale = 1.0 / math.sqrt(E)
    keep = helpers.drop_mult_attention(seed, pdrop, B, H, L, S)            # host replica of the library's mask, keep * scale
    assert abs(float((keep > 0).float().mean()) - (1 - pdrop)) < 0.02
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    scores = torch.einsum("blhe,she->bhls", qf.view(B, L, H, E), kf.view(S, H, E))
    A = torch.softmax(scale * scores, dim=-1) * keep
    ref = torch.einsum("bhls,she->blhe", A, vf.view(S, H, E)).reshape(B, L, H * E)
    ref.backward(do.float())
    o, lse = ops.attention_fwd(dev(q), dev(k), dev(v), H, H, E, scale, False, shared_kv=True, dropout=(pdrop, seed))
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD
    dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), o, lse, dev(do), H, H, E, scale, False, shared_kv=True, dropout=(pdrop, seed))
    assert rel_err(dq.float(), qf.grad) < TOL_ATTN_BWD
    assert rel_err(dk.float(), kf.grad) < TOL_ATTN_BWD
    assert rel_err(dv.float(), vf.grad) < TOL_ATTN_BWD


def test_attention_online_softmax_rescale(ops):
    """force the running-max rescale branch: one key far above the rest, placed in the LAST chunk"""
    B, T, H, D = 1, 192, 1, 64
    q = torch.randn(B, T, D, generator=g(1))
    k = torch.randn(B, T, D, generator=g(2))
    v = torch.randn(B, T, D, generator=g(3))
    k[0, 180] = 6.0 * q[0, 190] / q[0, 190].norm() * math.sqrt(D)
    q, k, v = q.to(BF16), k.to(BF16), v.to(BF16)
    ref = ref_attention(q.double(), k.double(), v.double(), H, H, D, 1 / math.sqrt(D), False)
    o, _ = ops.attention_fwd(dev(q), dev(k), dev(v), H, H, D, 1 / math.sqrt(D), False)
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD


# ------------------------------------------------------------------------------------------------ transposed-operand GEMM
@pytest.mark.parametrize("M,Nn,K,split", [(1024, 384, 4096, 0), (136, 264, 515, 1), (768, 1024, 4099, 4), (1152, 16384, 32, 0), (8, 8, 40, 1)])
def test_gemm_xt_weight_gradient_form(ops, M, Nn, K, split):
    """dW = dY^T . X with BOTH operands K-major (the contraction runs over their rows), fp32 output + the column sums of dY (bias
    gradient); ragged M / N / K, one-pass and split contraction"""
    dy = torch.randn(K, M, generator=g(1)).to(BF16).cuda()
    x = torch.randn(K, Nn, generator=g(2)).to(BF16).cuda()
    out, cs = ops.gemm_xt(dy, x, a_trans=True, b_trans=True, out_dtype=F32, want_colsum=True, split_k=split)
    ref = dy.double().t() @ x.double()
    assert rel_err(out.double(), ref) < TOL_F32
    assert rel_err(cs.double(), dy.double().sum(0)) < TOL_F32
    out_b = ops.gemm_xt(dy, x, a_trans=True, b_trans=True, out_dtype=BF16, alpha=0.5, split_k=split)
    assert rel_err(out_b.double(), 0.5 * ref) < TOL_BF16


@pytest.mark.parametrize("M,Nn,K,split", [(4096, 768, 1024, 0), (300, 136, 264, 1), (32, 16384, 1152, 0), (515, 72, 2048, 4)])
def test_gemm_xt_input_gradient_form(ops, M, Nn, K, split):
    """dX = dY . W with W K-major ([K = out features, N = in features], as the forward holds it); and the mirrored a_trans form"""
    dy = torch.randn(M, K, generator=g(3)).to(BF16).cuda()
    w = torch.randn(K, Nn, generator=g(4)).to(BF16).cuda()
    ref = dy.double() @ w.double()
    out = ops.gemm_xt(dy, w, b_trans=True, out_dtype=BF16, split_k=split)
    assert rel_err(out.double(), ref) < TOL_BF16
    out32 = ops.gemm_xt(dy, w, b_trans=True, out_dtype=F32, split_k=split)
    assert rel_err(out32.double(), ref) < TOL_F32
    wn = w.t().contiguous()                                                     # [Nn, K] K-contiguous
    dyk = dy.t().contiguous() if M % 8 == 0 else None                           # [K, M] K-major A
    if dyk is not None:
        out_t = ops.gemm_xt(dyk, wn, a_trans=True, out_dtype=F32, split_k=split)
        assert rel_err(out_t.double(), ref) < TOL_F32


def test_gemm_xt_strided_views_and_errors(ops):
    """column-sliced K-major B (x2[:, :Kin] of a K-padded activation) and the argument checks"""
    x = torch.randn(512, 448, generator=g(5)).to(BF16).cuda()
    dy = torch.randn(512, 128, generator=g(6)).to(BF16).cuda()
    out = ops.gemm_xt(dy, x[:, :384], a_trans=True, b_trans=True, out_dtype=F32)
    assert rel_err(out.double(), dy.double().t() @ x[:, :384].double()) < TOL_F32
    with pytest.raises(Exception):
        ops.gemm_xt(dy, x)                                                      # no K-major operand
    with pytest.raises(Exception):
        ops.gemm_xt(dy[:, :100], x, a_trans=True, b_trans=True)                 # fine: row stride 128 — but K mismatch below
        ops.gemm_xt(dy, x[:100], a_trans=True, b_trans=True)


def test_linear_pair_backward_equals_two_linears(ops):
    """the key / value projections share their input: [Wk; Wv] treated as ONE weight in the backward (one dX GEMM over K = 2 N, one
    d[Wk; Wv] GEMM) == the two projections differentiated separately, for adjacent incoming gradients (what the reprogramming
    attention's backward hands over) and for unrelated ones (concatenated)"""
    from med_ts_llm_amd.hip.optim import Bf16Shadow
    M, K, N1 = 1024, 768, 256
    x = torch.randn(M, K, generator=g(1)).to(BF16).cuda()
    W = [torch.nn.Parameter((torch.randn(N1, K, generator=g(2 + i)) * 0.05).cuda()) for i in range(2)]
    b = [torch.nn.Parameter((torch.randn(N1, generator=g(4 + i)) * 0.05).cuda()) for i in range(2)]
    base = torch.zeros(2 * N1, K, dtype=BF16, device="cuda")
    sh = [Bf16Shadow(W[0], base[:N1]), Bf16Shadow(W[1], base[N1:])]
    dyy = torch.randn(M, 2 * N1, generator=g(6)).to(BF16).cuda()
    for adjacent in (True, False):
        xs = [x.clone().requires_grad_(True) for _ in range(2)]
        y1, y2 = ops.LinearPairFn.apply(xs[0], W[0], b[0], W[1], b[1], sh[0], sh[1], base)
        d1, d2 = (dyy[:, :N1], dyy[:, N1:]) if adjacent else (dyy[:, :N1].contiguous(), dyy[:, N1:].contiguous())
        torch.autograd.backward([y1, y2], [d1, d2])
        got = [xs[0].grad.clone()] + [p.grad.clone() for p in W + b]
        for p in W + b:
            p.grad = None
        r1 = ops.LinearFn.apply(xs[1], W[0], b[0])
        r2 = ops.LinearFn.apply(xs[1], W[1], b[1])
        assert torch.equal(r1, y1) and torch.equal(r2, y2)
        torch.autograd.backward([r1, r2], [d1.contiguous(), d2.contiguous()])
        want = [xs[1].grad.clone()] + [p.grad.clone() for p in W + b]
        for p in W + b:
            p.grad = None
        assert rel_err(got[0].float(), want[0].float()) < 2 * TOL_BF16      # (one bf16 rounding of the sum here; two roundings + a rounded add there)
        for a_, w_ in zip(got[1:], want[1:]):
            assert rel_err(a_, w_) < TOL_F32


@pytest.mark.parametrize("B,C,R,J,O,mean,out_dtype", [(3, 5, 700, 1, 1, True, BF16), (2, 12, 1000, 1, 1, False, BF16), (4, 3, 96, 1, 1, True, F32),
                                                      (2, 7, 33, 4, 4, False, F32)])
def test_channel_mix_vs_reference(ops, B, C, R, J, O, mean, out_dtype):
    """channel mean ("add" / "independent") and the feature_weighting Linear on the channel-last view ("weighted-average" / "merge-end"):
    forward and all three gradients vs the ATen formulation in fp64"""
    x = torch.randn(B, C, R, J, generator=g(1)).to(BF16)
    W = None if mean else torch.randn(O, J * C, generator=g(2)) * 0.3
    b = None if mean else torch.randn(O, generator=g(3)) * 0.1
    dy = torch.randn(B, R, O, generator=g(4)).to(out_dtype)
    xf = x.double().requires_grad_(True)
    if mean:
        ref = xf.mean(dim=1).reshape(B, R, 1)
    else:
        Wd, bd = W.double().requires_grad_(True), b.double().requires_grad_(True)
        ref = F.linear(xf.permute(0, 2, 3, 1).reshape(B, R, J * C), Wd, bd)          # k = j * C + c
    ref.backward(dy.double())
    xd = x.cuda().requires_grad_(True)
    Wg = None if mean else W.cuda().requires_grad_(True)
    bg = None if mean else b.cuda().requires_grad_(True)
    y = ops.ChannelMixFn.apply(xd, Wg, bg, out_dtype)
    assert y.dtype == out_dtype and rel_err(y.double(), ref) < (TOL_BF16 if out_dtype == BF16 else TOL_F32)
    y.backward(dy.cuda())
    assert rel_err(xd.grad.double(), xf.grad) < TOL_BF16
    if not mean:
        assert rel_err(Wg.grad.double(), Wd.grad) < TOL_F32 and rel_err(bg.grad.double(), bd.grad) < TOL_F32


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,d", [(7, 64), (300, 768), (64, 4096), (10, 1000)])
def test_layernorm(ops, M, d):
    x = torch.randn(M, d, generator=g(1)) * 2 + 0.5
    w, b = 1 + 0.1 * torch.randn(d, generator=g(2)), 0.1 * torch.randn(d, generator=g(3))
    dy = torch.randn(M, d, generator=g(4)).to(BF16)
    dres = torch.randn(M, d, generator=g(5))
    xf = x.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (d,), w, b, 1e-5)
    ref.backward(dy.float())
    y, stats = ops.norm_fwd(dev(x), dev(w), dev(b), 1e-5)
    assert rel_err(y.float(), ref) < TOL_BF16
    assert rel_err(stats[:, 0], x.mean(1)) < 1e-5
    dx, dxb = ops.norm_bwd(dev(dy), dev(x), dev(w), stats, dres_in=dev(dres), want_bf16=True)
    assert rel_err(dx, dres + xf.grad) < 1e-5
    assert rel_err(dxb.float(), dres + xf.grad) < TOL_BF16


@pytest.mark.parametrize("M,d", [(9, 128), (130, 4096)])
def test_rmsnorm(ops, M, d):
    from oracle.medtsllm_oracle import rms_norm
    x = torch.randn(M, d, generator=g(1)) * 3
    w = 1 + 0.1 * torch.randn(d, generator=g(2))
    dy = torch.randn(M, d, generator=g(4)).to(BF16)
    xf = x.clone().requires_grad_(True)
    ref = rms_norm(xf, w, 1e-5)
    ref.backward(dy.float())
    y, stats = ops.norm_fwd(dev(x), dev(w), None, 1e-5, rms=True)
    assert rel_err(y.float(), ref) < TOL_BF16
    dx = ops.norm_bwd(dev(dy), dev(x), dev(w), stats, rms=True)
    assert rel_err(dx, xf.grad) < 1e-5


def test_norm_row_gather(ops):
    B, T, Pn, d = 3, 12, 5, 256
    x = torch.randn(B * T, d, generator=g(1))
    w, b = torch.randn(d, generator=g(2)), torch.randn(d, generator=g(3))
    y, stats = ops.norm_fwd(dev(x), dev(w), dev(b), 1e-5, rows=(Pn, T, T - Pn), M=B * Pn)
    sel = x.view(B, T, d)[:, -Pn:, :].reshape(B * Pn, d)
    assert rel_err(y.float(), F.layer_norm(sel, (d,), w, b, 1e-5)) < TOL_BF16
    dy = torch.randn(B * Pn, d, generator=g(4)).to(BF16)
    sf = sel.clone().requires_grad_(True)
    F.layer_norm(sf, (d,), w, b, 1e-5).backward(dy.float())
    dx = ops.norm_bwd(dev(dy), dev(x), dev(w), stats, rows=(Pn, T, T - Pn))
    full = torch.zeros(B, T, d)
    full[:, -Pn:, :] = sf.grad.view(B, Pn, d)
    assert rel_err(dx, full.view(B * T, d)) < 1e-5


# ------------------------------------------------------------------------------------------------ tokenizer
@pytest.mark.parametrize("name", ["gpt2_concat_fc", "llama_concat_semseg", "gpt2_indep_recon"])
def test_patch_index_map_bit_exact(ops, name):
    meta, data, _, _ = load_case(name)
    idx = ops.patch_index_map(meta["L"], meta["patch_len"], meta["stride"], "cuda").cpu().numpy()
    assert idx.dtype == np.int32 and np.array_equal(idx, data["patch_index_map"])


@pytest.mark.parametrize("B,L,C,pl,st,dm,concat", [(2, 64, 3, 16, 8, 8, True), (2, 100, 3, 16, 8, 8, False), (4, 1024, 12, 16, 8, 32, True),
                                                    (3, 512, 1, 16, 8, 32, False)])
def test_patch_tokenizer(ops, B, L, C, pl, st, dm, concat):
    from oracle import medtsllm_oracle as O
    x = torch.randn(B, L, C, generator=g(1)) * torch.linspace(0.3, 3, C) + torch.linspace(-2, 2, C)
    x[0, :, 0] = 1.25  # constant channel: stdev = sqrt(eps)
    w = torch.randn(dm, pl, 3, generator=g(2)) * 0.3
    wf = w.clone().requires_grad_(True)
    mean, stdev = O.revin_stats(x)
    ref = O.patch_embed(O.revin_norm(x, mean, stdev), wf, pl, st)          # [B*C, P, dm]
    P = ref.shape[1]
    if concat:
        ref = ref.reshape(B, C, P, dm).permute(0, 2, 1, 3).reshape(B, P, C * dm)
    out, m, s = ops.patch_tokenize_fwd(dev(x), dev(w), pl, st, concat)
    width = ref.shape[-1]
    assert out.shape[-1] % 64 == 0 and torch.all(out[..., width:] == 0)
    assert rel_err(m, mean.view(B, C)) < 1e-6 and rel_err(s, stdev.view(B, C)) < 1e-6
    assert rel_err(out[..., :width].float(), ref) < TOL_BF16
    dout = torch.randn(ref.shape, generator=g(3)).to(BF16)
    ref.backward(dout.float())
    dpad = torch.zeros(out.shape, dtype=BF16)
    dpad[..., :width] = dout
    dw = ops.patch_tokenize_bwd(dev(x), m, s, dev(dpad), tuple(w.shape), pl, st, concat)
    assert rel_err(dw, wf.grad) < 1e-5
    # PatchEmbedding's train-mode dropout fused into both kernels: the counter mask of (seed, row of out, column of out),
    # replicated on the host and handed to the reference as an explicit multiplier
    pd, seed = 0.25, 424242
    rows = out.shape[0] * out.shape[1]
    mult = helpers.drop_mult_matrix(seed, pd, rows, out.shape[-1]).view(out.shape)[..., :width]
    assert abs(float((mult > 0).float().mean()) - (1 - pd)) < 0.05
    wf2 = w.clone().requires_grad_(True)
    ref2 = O.patch_embed(O.revin_norm(x, mean, stdev), wf2, pl, st)
    if concat:
        ref2 = ref2.reshape(B, C, P, dm).permute(0, 2, 1, 3).reshape(B, P, C * dm)
    ref2 = ref2 * mult
    out2, _, _ = ops.patch_tokenize_fwd(dev(x), dev(w), pl, st, concat, drop=(pd, seed))
    assert rel_err(out2[..., :width].float(), ref2) < TOL_BF16 and torch.all(out2[..., width:] == 0)
    ref2.backward(dout.float())
    dw2 = ops.patch_tokenize_bwd(dev(x), m, s, dev(dpad), tuple(w.shape), pl, st, concat, drop=(pd, seed))
    assert rel_err(dw2, wf2.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ prompt input statistics (a6 / f2)
def _twin(lags, L):
    """lags -> twin-pair ids min(k, n - k), n = 2 * (L // 2) = the length of irfft's default output: the autocorrelation sequence is
    symmetric, the order inside a pair is round-off noise"""
    n = 2 * (L // 2)
    return np.minimum(np.asarray(lags), n - np.asarray(lags))


def test_input_stats_vs_reference_goldens(ops):
    """device-computed statistics against the REFERENCE's golden lags (tests/golden/stats.npz: calcute_lags outputs of the real
    reference) and against torch's min / max / median / trend on the same tensor: values exact, lags equal as twin-pair ids"""
    z = np.load(GOLDEN / "stats.npz")
    x = torch.from_numpy(z["x"])                       # [2, 96, 3]
    B, L, C = x.shape
    st, lg, _ = ops.input_stats(dev(x), -1, 5)         # all channels: lags of the channel mean (reference: calcute_lags on [B, L, C])
    assert np.array_equal(_twin(lg.cpu().numpy().astype(np.int64), L), _twin(z["lags_3d"], L))
    st1, lg1, _ = ops.input_stats(dev(x), 1, 5)        # one channel (reference: calcute_lags on x[:, :, 1])
    assert np.array_equal(_twin(lg1.cpu().numpy().astype(np.int64), L), _twin(z["lags_2d"], L))
    for stats, xs in ((st, x), (st1, x[:, :, 1:2])):
        s = stats.cpu()
        assert torch.equal(s[:, :, 0], xs.min(dim=1).values) and torch.equal(s[:, :, 1], xs.max(dim=1).values)
        assert torch.equal(s[:, :, 2], xs.median(dim=1).values)                      # LOWER median, like torch.median
        assert torch.equal(s[:, :, 3] > 0, xs.diff(dim=1).sum(dim=1) > 0)
    # sizes of the metric workload, odd lengths, ties in the data (repeated values -> the median's rank window is wide)
    # ... and windows beyond the first version's 5460-point limit: 13 904 points (the series kernel drops its twiddle tables: 3 L floats no longer
    # fit the LDS) and 20 000 (both kernels table-free) — models/prompt.py has no torch stand-in on device tensors any more
    for (B, L, C) in ((4, 1024, 12), (3, 101, 2), (2, 64, 1), (1, 13904, 2), (1, 20000, 1)):
        x = torch.randn(B, L, C, generator=g(L)) + torch.linspace(-1, 1, L)[None, :, None] * torch.tensor([1.0, -1.0] * 6)[:C]
        x[:, 7::7, 0] = x[:, 3:4, 0]            # repeated values (not at t = 0: the trend is the sign of x[L-1] - x[0] up to round-off)
        st, lg, _ = ops.input_stats(dev(x), -1, 5)
        s = st.cpu()
        assert torch.equal(s[:, :, 0], x.min(dim=1).values) and torch.equal(s[:, :, 1], x.max(dim=1).values)
        assert torch.equal(s[:, :, 2], x.median(dim=1).values)
        assert torch.equal(s[:, :, 3] > 0, x.diff(dim=1).sum(dim=1) > 0)
        from med_ts_llm_amd.models.prompt import calc_lags
        ref = calc_lags(x, 5).numpy()
        # compare as SETS of twin ids per sample against the FFT-based reference: values one ulp apart may swap neighbouring ranks
        # of DIFFERENT pairs only if the pairs' correlations tie to ~1e-6, which random data does not produce
        assert np.array_equal(_twin(lg.cpu().numpy().astype(np.int64), L), _twin(ref, L)), (lg, ref)
        assert lg[:, 0].eq(0).all()                                               # lag 0 always ranks first


# ------------------------------------------------------------------------------------------------ elementwise / layout
def test_cast_transpose_colsum(ops):
    src = torch.randn(70, 50, generator=g(1))
    d, dt = ops.cast_pad(dev(src), ld_dst=64, want_t=True, ld_dst_t=128)
    assert torch.equal(d[:, :50].cpu(), src.to(BF16)) and torch.all(d[:, 50:] == 0)
    assert torch.equal(dt[:, :70].cpu(), src.to(BF16).t()) and torch.all(dt[:, 70:] == 0)
    xb = torch.randn(130, 70, generator=g(2)).to(BF16)
    t = ops.transpose_bf16(dev(xb), 192)
    assert torch.equal(t[:, :130].cpu(), xb.t()) and torch.all(t[:, 130:] == 0)
    assert rel_err(ops.colsum(dev(xb)), xb.float().sum(0)) < 1e-5
    big = torch.randn(1001, generator=g(3))
    assert torch.equal(ops.to_bf16(dev(big)).cpu(), big.to(BF16))
    assert torch.equal(ops.to_f32(dev(big.to(BF16))).cpu(), big.to(BF16).float())


def test_rope_and_swiglu(ops):
    from oracle import medtsllm_oracle as O
    B, T, H, Hkv, D = 2, 37, 4, 2, 64
    M = B * T
    qkv = torch.randn(M, (H + 2 * Hkv) * D, generator=g(1)).to(BF16)
    cos, sin = O.rope_tables(T, D, 10000.0)
    x = qkv.float().view(B, T, H + 2 * Hkv, D)
    rot = x[:, :, : H + Hkv]
    ref = rot * cos[None, :, None, :] + O.rotate_half(rot) * sin[None, :, None, :]
    out = ops.rope_inplace(dev(qkv).clone(), dev(cos), dev(sin), T, H + Hkv, D)
    o4 = out.float().cpu().view(B, T, H + 2 * Hkv, D)
    assert rel_err(o4[:, :, : H + Hkv], ref) < TOL_BF16
    assert torch.equal(o4[:, :, H + Hkv:], x[:, :, H + Hkv:])          # v heads untouched
    back = ops.rope_inplace(out.clone(), dev(cos), dev(sin), T, H + Hkv, D, inverse=True)
    assert rel_err(back.float().cpu(), qkv.float()) < 2 * TOL_BF16      # R^T R = I
    Fd = 96
    gu = torch.randn(M, 2 * Fd, generator=g(2)).to(BF16)
    dh = torch.randn(M, Fd, generator=g(3)).to(BF16)
    gf = gu.float().requires_grad_(True)
    ref = F.silu(gf[:, :Fd]) * gf[:, Fd:]
    ref.backward(dh.float())
    h = ops.swiglu_fwd(dev(gu))
    assert rel_err(h.float(), ref) < 2 * TOL_BF16
    assert rel_err(ops.swiglu_bwd(dev(gu), dev(dh)).float(), gf.grad) < TOL_BF16


def test_assemble_and_denorm(ops):
    B, n_tok, Pn, d, V = 3, 5, 4, 128, 50
    emb = torch.randn(V, d, generator=g(1))
    wpe = torch.randn(n_tok + Pn, d, generator=g(2))
    xt = torch.randn(B, Pn, d, generator=g(3)).to(BF16)
    ids = torch.randint(0, V, (B, n_tok), generator=g(4), dtype=torch.int32)
    ref = torch.cat([emb[ids.long()], xt.float()], dim=1) + wpe
    assert rel_err(ops.assemble_llm_input(dev(ids), dev(emb), dev(xt), dev(wpe)), ref) < 1e-7
    ref1 = torch.cat([emb[ids[:1].long()].expand(B, -1, -1), xt.float()], dim=1)
    assert rel_err(ops.assemble_llm_input(dev(ids[:1]), dev(emb), dev(xt), None), ref1) < 1e-7
    assert rel_err(ops.assemble_llm_input(None, None, dev(xt), None), xt.float()) < 1e-7
    # embd dropout fused into the assembly == assembly, then mtl_dropout_f32 (bit-identical); its backward = masked token rows in bf16
    from helpers import drop_mult_matrix
    p_, seed = 0.1, 4242
    h_plain = ops.assemble_llm_input(dev(ids), dev(emb), dev(xt), dev(wpe))
    h_drop = ops.assemble_llm_input(dev(ids), dev(emb), dev(xt), dev(wpe), drop=(p_, seed))
    assert torch.equal(h_drop, ops.dropout_f32(h_plain, p_, seed))
    mult = drop_mult_matrix(seed, p_, B * (n_tok + Pn), d).view(B, n_tok + Pn, d)
    assert torch.equal(h_drop.cpu(), h_plain.cpu() * mult)
    dh = torch.randn(B, n_tok + Pn, d, generator=g(8))
    assert torch.equal(ops.assemble_bwd(dev(dh), n_tok, drop=(p_, seed)).cpu(), (dh * mult)[:, n_tok:].to(BF16))
    assert torch.equal(ops.assemble_bwd(dev(dh), n_tok).cpu(), dh[:, n_tok:].to(BF16))
    m = torch.randn(300, 770, generator=g(9)).to(BF16)
    assert rel_err(ops.rowsum(dev(m)), m.double().sum(1)) < 1e-6
    y = torch.randn(B, 7, 3, generator=g(5))
    mean, sd = torch.randn(B, 3, generator=g(6)), torch.rand(B, 3, generator=g(7)) + 0.1
    assert rel_err(ops.revin_denorm(dev(y), dev(mean), dev(sd)), y * sd[:, None] + mean[:, None]) < 1e-6
    assert rel_err(ops.revin_denorm(dev(y), None, dev(sd)), y * sd[:, None]) < 1e-6
    yb = y.to(BF16)                                  # bf16 in (the model's head output), bf16 out (the gradient handed back to it)
    assert rel_err(ops.revin_denorm(dev(yb), dev(mean), dev(sd)), yb.float() * sd[:, None] + mean[:, None]) < 1e-6
    assert torch.equal(ops.revin_denorm(dev(y), None, dev(sd), out_dtype=BF16).cpu(), (y * sd[:, None]).to(BF16))


@pytest.mark.gpu
@pytest.mark.parametrize("wd,decoupled", [(0.0, False), (0.01, False), (0.01, True)])
def test_hip_adam_matches_torch(ops, wd, decoupled):
    """mtl_adam_step == torch.optim.Adam / AdamW (R:tasks/base.py:97,99) over 5 steps, incl. odd sizes, an unaligned
    view and the bf16 shadow of the updated weight."""
    from med_ts_llm_amd.hip.optim import HipAdam, Bf16Shadow
    g = torch.Generator().manual_seed(3)
    shapes = [(37, 1001), (5,), (64, 64), (3, 7, 5), (1,), (130, 259)]
    ref_p = [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]
    our_p = [p.detach().clone().requires_grad_() for p in ref_p]
    ref = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref_p, lr=1e-2, weight_decay=wd)
    ours = HipAdam(our_p, lr=1e-2, weight_decay=wd, decoupled_weight_decay=decoupled)
    shadow = Bf16Shadow(our_p[0], torch.zeros((37, 1024), dtype=torch.bfloat16, device="cuda"))
    ours.register_shadow(shadow)
    for step in range(5):
        for a, b in zip(ref_p, our_p):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (step - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        v0 = our_p[0]._version
        ref.step(), ours.step()
        assert our_p[0]._version > v0 and shadow.fresh()
        for a, b in zip(ref_p, our_p):
            torch.testing.assert_close(b, a, rtol=2e-6, atol=2e-7)
    torch.testing.assert_close(shadow.tensor[:, :1001], our_p[0].detach().to(torch.bfloat16), rtol=0, atol=0)
    assert float(shadow.tensor[:, 1001:].abs().max()) == 0.0
    sd = ours.state_dict()["state"][0]
    assert set(sd) == {"step", "exp_avg", "exp_avg_sq"} and sd["step"] == 5
    torch.testing.assert_close(sd["exp_avg"], ref.state_dict()["state"][0]["exp_avg"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_adam_many_tensors_with_empty_ones(ops):
    """more tensors than one kernel table holds (24), with zero-element parameters in between: every tensor gets exactly ONE
    update per step (the table loop used to restart 24 entries after the previous table's FIRST tensor, so an empty tensor
    inside a table made the tensors after it step twice)"""
    from med_ts_llm_amd.hip.optim import HipAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 3)] * 10 + [(0,)] + [(5,)] * 12 + [(0, 4)] + [(9, 2)] * 9 + [(33,)] * 20
    ref_p = [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]
    our_p = [p.detach().clone().requires_grad_() for p in ref_p]
    ref, ours = torch.optim.Adam(ref_p, lr=1e-2), HipAdam(our_p, lr=1e-2)
    for step in range(3):
        for a, b in zip(ref_p, our_p):
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step(), ours.step()
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            torch.testing.assert_close(b, a, rtol=2e-6, atol=2e-7, msg=lambda m, i=i: f"tensor {i} {shapes[i]}: {m}")


@pytest.mark.gpu
@pytest.mark.parametrize("M,Nn,K,S", [(1024, 768, 64 * 32, 16), (200, 132, 64 * 8, 4), (128, 64, 64 * 6, 4)])
def test_gemm_split_k_paths(ops, M, Nn, K, S):
    """split-K through the persistent work-item kernel (k-steps divisible by S) and through the classic one (not
    divisible): both equal the unsplit GEMM up to fp32 summation order, bf16 output identical after rounding almost everywhere"""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.1).to(BF16).cuda()
    bias = torch.randn(Nn, generator=g).cuda()
    ref = (A.float() @ B.float().t() + bias)
    one = ops.gemm_nt(A, B, bias=bias, out_dtype=F32)
    split = ops.gemm_nt(A, B, bias=bias, out_dtype=F32, split_k=S)
    assert rel_err(one, ref) < 2e-5 and rel_err(split, ref) < 2e-5
    splitb = ops.gemm_nt(A, B, bias=bias, split_k=S)
    assert splitb.dtype == BF16 and rel_err(splitb.float(), ref) < TOL_BF16


@pytest.mark.gpu
@pytest.mark.parametrize("M,Nn,K", [(1024, 50257, 128), (300, 1001, 64), (128, 7, 64)])
def test_gemm_fp32_out_unaligned_rows(ops, M, Nn, K):
    """N % 4 != 0 with fp32 output (the mapping weight gradient [num_tokens, 50257]): rows are only 4-B aligned, the
    persistent kernel stores dwords; exact on small-integer operands, padding beyond N untouched"""
    g = torch.Generator().manual_seed(Nn)
    A = torch.randint(-3, 4, (M, K), generator=g).to(BF16).cuda()
    B = torch.randint(-3, 4, (Nn, K), generator=g).to(BF16).cuda()
    buf = torch.full((M * Nn + 8,), 7.0, dtype=F32, device="cuda")
    out = buf[:M * Nn].view(M, Nn)
    ops.gemm_nt(A, B, out=out)
    assert torch.equal(out, A.float() @ B.float().t())
    assert torch.all(buf[M * Nn:] == 7.0)


# ----------------------------------------------------------------------------- GPT-2 train-mode dropout building blocks
@pytest.mark.gpu
def test_dropout_sites_match_host_hash(ops):
    """resid_pdrop in the RESID GEMM epilogue (both kernels), its backward twin in norm_bwd's bf16 copy, and the plain
    f32 dropout (embd_pdrop): all three use the (seed, 0, row, col) counter hash replicated in tests/helpers.py."""
    from helpers import drop_mult_matrix
    p, seed = 0.1, 0xBEEF1234
    for M, Nn, K in [(256, 192, 128), (100, 36, 64)]:          # wave-level epilogue / classic kernel (unaligned N)
        A = torch.randn(M, K, generator=g(1)).to(BF16).cuda()
        B = (torch.randn(Nn, K, generator=g(2)) * 0.1).to(BF16).cuda()
        bias = torch.randn(Nn, generator=g(3)).cuda()
        res = torch.randn(M, Nn, generator=g(4)).cuda()
        out = ops.gemm_nt(A, B, bias=bias, epilogue=ops.N.EPI_RESID, aux_in=res, out_dtype=F32, drop=(p, seed))
        lin = rb(A.float().cpu() @ B.float().cpu().t() + bias.cpu())
        mult = drop_mult_matrix(seed, p, M, Nn)
        assert abs(float((mult > 0).float().mean()) - (1 - p)) < 0.03
        assert rel_err(out, res.cpu() + lin * mult) < 1e-5
    # norm backward: fp32 stream unmasked, bf16 copy masked
    M, d = 64, 256
    x = torch.randn(M, d, generator=g(5)).cuda()
    gamma, beta = torch.randn(d, generator=g(6)).cuda(), torch.randn(d, generator=g(7)).cuda()
    y, stats = ops.norm_fwd(x, gamma, beta, 1e-5)
    dy = torch.randn(M, d, generator=g(8)).to(BF16).cuda()
    plain, plain_b = ops.norm_bwd(dy, x, gamma, stats, want_bf16=True)
    masked, masked_b = ops.norm_bwd(dy, x, gamma, stats, want_bf16=True, bf16_drop=(p, seed))
    assert torch.equal(plain, masked)
    assert rel_err(masked_b.float(), rb(plain.cpu() * drop_mult_matrix(seed, p, M, d))) < 1e-6
    # plain dropout and its backward
    h = torch.randn(3, 50, d, generator=g(9)).cuda()
    assert torch.equal(ops.dropout_f32(h, p, seed).cpu(), h.cpu() * drop_mult_matrix(seed, p, 150, d).view(3, 50, d))


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [1, 0])
@pytest.mark.parametrize("pdrop", [0.1, 0.5])
def test_attention_causal_dropout(ops, pdrop, resident):
    """attn_pdrop on the causal backbone attention (HF gpt2 :65): forward and all three gradients vs the host-mask reference,
    including the pruned backward (queries = last n_grad rows: the mask must be indexed by ABSOLUTE query position)"""
    from helpers import drop_mult_attention
    B, T, H, D, seed = 2, 100, 2, 64, 777
    q, k, v = (torch.randn(B, T, H * D, generator=g(i)).to(BF16) for i in (1, 2, 3))
    do = torch.randn(B, T, H * D, generator=g(4)).to(BF16)
    scale = 1.0 / math.sqrt(D)
    mult = drop_mult_attention(seed, pdrop, B, H, T, T)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qf.view(B, T, H, D), kf.view(B, T, H, D)) * scale
    s = s.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf"))
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1) * mult, vf.view(B, T, H, D)).reshape(B, T, H * D)
    ref.backward(do.float())
    with ops.attention_tune(resident=bool(resident)):        # K/V-resident kernels, or the chunked ones
        o, lse = ops.attention_fwd(dev(q), dev(k), dev(v), H, H, D, scale, True, dropout=(pdrop, seed))
        dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), o, lse, dev(do), H, H, D, scale, True, dropout=(pdrop, seed))
    assert rel_err(o.float(), ref) < TOL_ATTN_FWD
    assert rel_err(dq.float(), qf.grad) < TOL_ATTN_BWD and rel_err(dk.float(), kf.grad) < TOL_ATTN_BWD and rel_err(dv.float(), vf.grad) < TOL_ATTN_BWD


@pytest.mark.gpu
def test_transpose_with_column_sums(ops):
    for R, Cc, ld in [(300, 130, 320), (64, 64, 64), (4096, 1152, 4096), (37, 5, 64)]:
        x = torch.randn(R, Cc, generator=g(R)).to(BF16).cuda()
        t, sums = ops.transpose_colsum_bf16(x, ld)
        assert torch.equal(t, ops.transpose_bf16(x, ld))
        assert rel_err(sums, x.double().sum(0).cpu()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(128, 64, 2, 4), (128, 96, 2, 4), (128, 96, 3, 4), (128, 128, 2, 8), (128, 128, 3, 8), (128, 128, 2, 4),
                                 (128, 192, 2, 8), (256, 128, 3, 16), (256, 128, 2, 16), (256, 192, 2, 8), (256, 96, 3, 8), (256, 96, 2, 8)])
def test_gemm_every_tile_configuration(ops, cfg):
    """each persistent-kernel tile configuration, forced through mtl_gemm_args.tune_* (ops.gemm_tune), on ragged shapes (edge tiles in M and N)
    with the plain, residual and accumulate epilogues; the launch heuristic only ever picks among these"""
    lib = ops.lib()
    for (M, Nn, K) in [(300, 260, 128), (37, 100, 64), (515, 388, 320)]:
        A = torch.randn(M, K, generator=g(M)).to(BF16).cuda()
        B = (torch.randn(Nn, K, generator=g(Nn)) * 0.2).to(BF16).cuda()
        bias = torch.randn(Nn, generator=g(K)).cuda()
        res = torch.randn(M, Nn, generator=g(7)).cuda()
        lin = A.double().cpu() @ B.double().cpu().t()
        with ops.gemm_tune(*cfg):
            plain = ops.gemm_nt(A, B, bias=bias)
            resid = ops.gemm_nt(A, B, bias=bias, epilogue=ops.N.EPI_RESID, aux_in=res, out_dtype=F32)
            accum = ops.gemm_nt(A, B, epilogue=ops.N.EPI_ACCUM, out=res.clone())
        assert rel_err(plain.float(), lin + bias.double().cpu()) < TOL_BF16
        assert rel_err(resid, res.double().cpu() + rb((lin + bias.double().cpu()).float()).double()) < 1e-4   # a few bf16 rounding ties
        assert rel_err(accum, res.double().cpu() + lin) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(128, 64, 2, 4), (128, 96, 2, 4), (128, 128, 2, 8), (128, 192, 2, 8), (256, 128, 3, 16), (256, 192, 2, 8), (256, 256, 2, 8)])
def test_gemm_bf16_epilogues_on_every_tile_configuration(ops, cfg):
    """the bf16-output epilogues (plain, GELU + saved pre-activation, dGELU, SwiGLU, dSwiGLU) on ragged shapes for every tile
    configuration: waves with an even number of column tiles store 8-column pairs (16-byte stores), odd ones 4-column quads,
    edge tiles fall back to predicated narrow stores — all must equal the automatic configuration's result up to summation order"""
    lib = ops.lib()
    for (M, Nn, K) in [(300, 264, 128), (517, 392, 192), (130, 1032, 64)]:
        A = torch.randn(M, K, generator=g(M)).to(BF16).cuda()
        B = (torch.randn(Nn, K, generator=g(Nn)) * 0.2).to(BF16).cuda()
        bias = torch.randn(Nn, generator=g(K)).cuda()
        pre_in = torch.randn(M, Nn, generator=g(9)).to(BF16).cuda()
        gu = torch.randn(M, 2 * Nn, generator=g(10)).to(BF16).cuda()

        def run():
            pre = torch.zeros(M, Nn, dtype=BF16, device="cuda")
            act = ops.gemm_nt(A, B, bias=bias, epilogue=ops.N.EPI_GELU, aux_out=pre)
            sact = torch.zeros(M, Nn // 2, dtype=BF16, device="cuda")
            sgu = ops.gemm_nt(A, B, epilogue=ops.N.EPI_SWIGLU, aux_out=sact)
            dgu = torch.zeros(M, 2 * Nn, dtype=BF16, device="cuda")
            ops.gemm_nt(A, B, out=dgu, epilogue=ops.N.EPI_DSWIGLU, aux_in=gu)
            return (ops.gemm_nt(A, B, bias=bias), act, pre, ops.gemm_nt(A, B, epilogue=ops.N.EPI_DGELU, aux_in=pre_in), sgu, sact, dgu)

        ref = run()
        with ops.gemm_tune(*cfg):
            out = run()
        lin = A.double().cpu() @ B.double().cpu().t() + bias.double().cpu()
        assert rel_err(out[0].float(), lin) < TOL_BF16
        for a, b in zip(out, ref):
            assert rel_err(a.float(), b.float().double().cpu()) < 2e-3, cfg


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(128, 64, 2, 4), (128, 96, 2, 4), (128, 96, 2, 8), (128, 128, 2, 8), (128, 192, 2, 8), (256, 96, 3, 8), (256, 192, 2, 8),
                                 (256, 256, 2, 8)])
@pytest.mark.parametrize("tiles_m", [8, 16, 24])
def test_gemm_tile_order_whole_rows_per_xcd(ops, cfg, tiles_m):
    """grids with a multiple of 8 tile rows take the whole-rows-per-XCD order with the per-XCD column rotation (and, with two
    k-groups, the k rotation): every tile must still be computed exactly once, for 1 .. 9 tile columns incl. a ragged last one"""
    lib = ops.lib()
    bm, bn = cfg[0], cfg[1]
    M = tiles_m * bm - 5                                    # ragged last tile row, same number of tile rows
    for tiles_n, K in ((1, 128), (3, 256), (5, 128), (9, 192)):
        Nn = tiles_n * bn - (8 if tiles_n > 1 else 0)
        if cfg == (128, 96, 2, 8):
            if tiles_m * tiles_n > 256:
                continue                                    # two k-groups: at most one tile per CU ...
            K = 256                                         # ... and an even number (>= 4) of 64-wide k-tiles
        A = torch.randn(M, K, generator=g(M + tiles_n)).to(BF16).cuda()
        B = (torch.randn(Nn, K, generator=g(Nn)) * 0.2).to(BF16).cuda()
        with ops.gemm_tune(*cfg):
            out = ops.gemm_nt(A, B, out_dtype=F32)
        ref = A.double().cpu() @ B.double().cpu().t()
        assert rel_err(out, ref) < 1e-5, (cfg, tiles_m, tiles_n)


@pytest.mark.gpu
@pytest.mark.parametrize("M,Nn,K", [(300, 260, 256), (515, 388, 384), (4096, 768, 768), (128, 96, 3072)])
def test_gemm_two_k_groups(ops, M, Nn, K):
    """128x96 tile with two 4-wave k-groups (the configuration picked for grids of at most one tile per CU): same results as the
    single-group kernel up to fp32 summation order, for the plain / residual / accumulate / dgelu epilogues"""
    lib = ops.lib()
    A = torch.randn(M, K, generator=g(M)).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g(Nn)) * 0.2).to(BF16).cuda()
    bias = torch.randn(Nn, generator=g(K)).cuda()
    res = torch.randn(M, Nn, generator=g(7)).cuda()
    pre = torch.randn(M, Nn, generator=g(8)).to(BF16).cuda()
    outs = {}
    for cfg in [(128, 96, 2, 4), (128, 96, 2, 8)]:
        with ops.gemm_tune(*cfg):
            outs[cfg] = (ops.gemm_nt(A, B, bias=bias), ops.gemm_nt(A, B, bias=bias, epilogue=ops.N.EPI_RESID, aux_in=res, out_dtype=F32),
                         ops.gemm_nt(A, B, epilogue=ops.N.EPI_ACCUM, out=res.clone()), ops.gemm_nt(A, B, epilogue=ops.N.EPI_DGELU, aux_in=pre))
    lin = A.double().cpu() @ B.double().cpu().t()
    assert rel_err(outs[(128, 96, 2, 8)][0].float(), lin + bias.double().cpu()) < TOL_BF16
    for a, b in zip(outs[(128, 96, 2, 4)], outs[(128, 96, 2, 8)]):
        assert rel_err(b.float(), a.float().double().cpu()) < 2e-3          # bf16 outputs: a few one-ulp flips from the summation order
    assert rel_err(outs[(128, 96, 2, 8)][2], res.double().cpu() + lin) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("M,Nn,K", [(256, 256, 64), (256, 512, 128), (512, 768, 320), (1024, 256, 4096), (4096, 4096, 704),
                                    (4096, 8192, 256), (4096, 12288, 128), (4096, 8192, 320)])
def test_gemm_w4_hand_placed_kernel(ops, M, Nn, K):
    """gemm_nt_w4_kernel (256 x 256 tile, 4 waves, the generated hand-placed k-loop: tools/gen_gemm_w4_loop.py) against fp64 math on the same bf16 inputs,
    every epilogue it is dispatched for: fp32 / bf16 store with bias, residual (+ dropout, against the 8-wave kernel's identical mask), SwiGLU, dSwiGLU;
    odd and even k-tile counts (the loop is unrolled twice), one k-tile; 2 and 3 tiles per workgroup on a 256-CU device with an even k-tile count
    (the CHAINED path: a tile's trailing iterations stage the next tile's first k-tiles — cold entry, chained entry with and without a successor) and
    with an odd one (several tiles, no chaining); repeated launches are bit-identical (race screen)"""
    A = torch.randn(M, K, generator=g(M + K)).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g(Nn)) * 0.1).to(BF16).cuda()
    bias = torch.randn(Nn, generator=g(3)).cuda()
    res = torch.randn(M, Nn, generator=g(4)).cuda()
    gu = torch.randn(M, 2 * Nn, generator=g(5)).to(BF16).cuda()
    lin = A.double().cpu() @ B.double().cpu().t()

    def run(waves):
        with ops.gemm_tune(bm=256, bn=256, stages=2, waves=waves):
            act = torch.empty(M, Nn // 2, dtype=BF16, device="cuda")
            return dict(f32=ops.gemm_nt(A, B, out_dtype=F32), bf16=ops.gemm_nt(A, B, bias=bias),
                        resid=ops.gemm_nt(A, B, bias=bias, epilogue=ops.N.EPI_RESID, aux_in=res, out_dtype=F32, drop=(0.1, 77)),
                        swiglu=ops.gemm_nt(A, B, epilogue=ops.N.EPI_SWIGLU, aux_out=act), act=act,
                        dswiglu=ops.gemm_nt(A, B, out=torch.empty(M, 2 * Nn, dtype=BF16, device="cuda"), epilogue=ops.N.EPI_DSWIGLU, aux_in=gu))
    w4, w8 = run(4), run(8)
    assert rel_err(w4["f32"], lin) < TOL_F32                          # pre-rounding output: accumulation order only
    assert rel_err(w4["bf16"].float(), lin + bias.double().cpu()) < TOL_BF16
    for k in ("resid", "swiglu", "act", "dswiglu"):                   # same arithmetic as the 8-wave kernel up to the fp32 summation order of a tile
        assert rel_err(w4[k].float(), w8[k].float().double().cpu()) < (1e-4 if k == "resid" else 2e-3), k
    again = run(4)
    for k in w4:
        assert torch.equal(w4[k], again[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("M,Nn,K,S", [(512, 768, 64 * 75, 7), (256, 256, 64 * 16, 2), (1024, 768, 51200, 21)])
def test_gemm_w4_split_k_uneven_slabs(ops, M, Nn, K, S):
    """split-K on the 4-wave kernel: (tile, k-slab) work items whose slabs need not divide the k-steps (75 = 7 x 10 + 5; the mapping GEMM's 800 = 21 x 38 + 2),
    raw partial sums through the workspace, epilogue (bias, bf16 rounding) in the reduce kernel; against fp64 math and the unsplit kernel"""
    A = torch.randn(M, K, generator=g(M + S)).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g(Nn)) * 0.05).to(BF16).cuda()
    bias = torch.randn(Nn, generator=g(3)).cuda()
    lin = A.double().cpu() @ B.double().cpu().t()
    with ops.gemm_tune(bm=256, bn=256, stages=2, waves=4):              # (automatic only where S does not divide the k-steps: the first case)
        f32 = ops.gemm_nt(A, B, out_dtype=F32, split_k=S)
        b16 = ops.gemm_nt(A, B, bias=bias, split_k=S)
        again = ops.gemm_nt(A, B, out_dtype=F32, split_k=S)
    assert rel_err(f32, lin) < TOL_F32
    assert rel_err(b16.float(), lin + bias.double().cpu()) < TOL_BF16
    assert rel_err(f32, ops.gemm_nt(A, B, out_dtype=F32, split_k=1).double().cpu()) < 1e-5
    assert torch.equal(f32, again)                                     # repeated launches bit-identical
    if (K // 64) % S:
        assert torch.equal(f32, ops.gemm_nt(A, B, out_dtype=F32, split_k=S))      # the automatic dispatch takes the same kernel


@pytest.mark.gpu
def test_gemm_w4_row_mapped_operand_and_dispatch(ops):
    """the pruned backward's row-mapped A operand through the 4-wave kernel's per-instruction row offsets; shapes the kernel does not take
    (ragged M / N, row groups that are not multiples of 8) are refused when forced and fall to the 8-wave kernels when automatic"""
    K, Nn = 512, 1024
    Abig = torch.randn(8 * 384, K, generator=g(1)).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g(2)) * 0.1).to(BF16).cuda()
    idx = torch.cat([torch.arange(256, 384) + 384 * i for i in range(8)])
    with ops.gemm_tune(bm=256, bn=256, stages=2, waves=4):
        out = ops.gemm_nt(Abig, B, M=1024, a_rows=(128, 384, 256), out_dtype=F32)
        with pytest.raises(Exception):
            ops.gemm_nt(Abig[:300], B, out_dtype=F32)                  # M % 256 != 0
        with pytest.raises(Exception):
            ops.gemm_nt(Abig, B, M=1024, a_rows=(4, 12, 8), out_dtype=F32)      # 8-row staging pieces would straddle row groups
    assert rel_err(out, Abig[idx].double().cpu() @ B.double().cpu().t()) < TOL_F32
    auto = ops.gemm_nt(Abig[:300], B, out_dtype=F32)                   # automatic dispatch still serves it
    assert rel_err(auto, Abig[:300].double().cpu() @ B.double().cpu().t()) < TOL_F32
    # plain fp32 store into rows that are only 4-byte aligned (odd row length, as the mapping layer's [1024, 50257] weight gradient): dword stores
    wide = torch.zeros(1024, Nn + 1, dtype=F32, device="cuda")
    with ops.gemm_tune(bm=256, bn=256, stages=2, waves=4):
        ops.gemm_nt(Abig[:1024], B, out=wide[:, :Nn])
    assert rel_err(wide[:, :Nn], Abig[:1024].double().cpu() @ B.double().cpu().t()) < TOL_F32 and float(wide[:, Nn].abs().max()) == 0.0
    # several tiles per workgroup (512 tiles: chained entries) with the row map: the NEXT tile's row offsets go through the same map
    K2, N2 = 256, 8192
    A2 = torch.randn(32 * 384, K2, generator=g(5)).to(BF16).cuda()
    B2 = (torch.randn(N2, K2, generator=g(6)) * 0.1).to(BF16).cuda()
    idx2 = torch.cat([torch.arange(256, 384) + 384 * i for i in range(32)])
    with ops.gemm_tune(bm=256, bn=256, stages=2, waves=4):
        out2 = ops.gemm_nt(A2, B2, M=4096, a_rows=(128, 384, 256), out_dtype=F32)
    assert rel_err(out2, A2[idx2].double().cpu() @ B2.double().cpu().t()) < TOL_F32


@pytest.mark.gpu
@pytest.mark.parametrize("M,F,K", [(300, 96, 128), (8192, 1024, 256)])
def test_gemm_swiglu_epilogue_and_interleaved_backward(ops, M, F, K):
    """gate|up GEMM with the SwiGLU fused into the epilogue (row-interleaved weights: column 2j = gate_j, 2j+1 = up_j) ==
    plain GEMM + mtl_swiglu_fwd on the [gate | up] layout; the interleaved backward == the plain one, re-interleaved"""
    x = torch.randn(M, K, generator=g(1)).to(BF16).cuda()
    wg = (torch.randn(F, K, generator=g(2)) * 0.2).to(BF16).cuda()
    wu = (torch.randn(F, K, generator=g(3)) * 0.2).to(BF16).cuda()
    gu_ref = ops.gemm_nt(x, torch.cat([wg, wu], 0))                       # [M, 2F] = [gate | up]
    h_ref = ops.swiglu_fwd(gu_ref)
    w_il = torch.stack([wg, wu], dim=1).reshape(2 * F, K).contiguous()
    act = torch.empty(M, F, dtype=BF16, device="cuda")
    gu_il = ops.gemm_nt(x, w_il, epilogue=ops.N.EPI_SWIGLU, aux_out=act)
    # same products, but the fp32 summation order of a tile depends on the XCD it runs on (per-XCD k rotation) and the two launches
    # map columns to tiles differently: equal up to a few one-ulp bf16 flips
    assert rel_err(gu_il.view(M, F, 2)[:, :, 0].float(), gu_ref[:, :F].float()) < 1e-3
    assert rel_err(gu_il.view(M, F, 2)[:, :, 1].float(), gu_ref[:, F:].float()) < 1e-3
    assert rel_err(act.float(), ops.swiglu_fwd(torch.cat([gu_il.view(M, F, 2)[:, :, 0], gu_il.view(M, F, 2)[:, :, 1]], 1).contiguous()).float()) < 1e-3
    assert rel_err(act.float(), h_ref.float()) < 2e-3                     # + __expf vs expf in the sigmoid
    dh = torch.randn(M, F, generator=g(4)).to(BF16).cuda()
    gu_ref = torch.cat([gu_il.view(M, F, 2)[:, :, 0], gu_il.view(M, F, 2)[:, :, 1]], 1).contiguous()    # identical pre-activations for the layout check
    d_ref = ops.swiglu_bwd(gu_ref, dh)
    d_il = ops.swiglu_bwd(gu_il, dh, interleaved=True)
    assert torch.equal(d_il.view(M, F, 2)[:, :, 0], d_ref[:, :F]) and torch.equal(d_il.view(M, F, 2)[:, :, 1], d_ref[:, F:])


@pytest.mark.gpu
@pytest.mark.parametrize("epi", ["gelu", "swiglu"])
def test_gemm_backward_only_output_is_row_pruned(ops, epi):
    """bwd_rows=(group, first): the output only a backward reads (GELU: saved pre-activation, SWIGLU: saved gate|up) is written for
    rows with m % group >= first and left untouched elsewhere; the forward output is complete either way"""
    M, Nn, K, grp, first = 600, 192, 128, 50, 30
    A = torch.randn(M, K, generator=g(1)).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g(2)) * 0.2).to(BF16).cuda()
    rows = torch.arange(M)
    keep = (rows % grp >= first).cuda()
    if epi == "gelu":
        full_pre, pruned_pre = torch.empty(M, Nn, dtype=BF16, device="cuda"), torch.full((M, Nn), 7.0, dtype=BF16, device="cuda")
        act_full = ops.gemm_nt(A, B, epilogue=ops.N.EPI_GELU, aux_out=full_pre)
        act_pruned = ops.gemm_nt(A, B, epilogue=ops.N.EPI_GELU, aux_out=pruned_pre, bwd_rows=(grp, first))
        none_pre = torch.full((M, Nn), 7.0, dtype=BF16, device="cuda")
        ops.gemm_nt(A, B, epilogue=ops.N.EPI_GELU, aux_out=none_pre, bwd_rows=(grp, grp))
        assert bool((none_pre == 7.0).all())
    else:
        full_pre, pruned_pre = torch.empty(M, Nn, dtype=BF16, device="cuda"), torch.full((M, Nn), 7.0, dtype=BF16, device="cuda")
        act_full, act_pruned = torch.empty(M, Nn // 2, dtype=BF16, device="cuda"), torch.empty(M, Nn // 2, dtype=BF16, device="cuda")
        ops.gemm_nt(A, B, out=full_pre, epilogue=ops.N.EPI_SWIGLU, aux_out=act_full)
        ops.gemm_nt(A, B, out=pruned_pre, epilogue=ops.N.EPI_SWIGLU, aux_out=act_pruned, bwd_rows=(grp, first))
    assert torch.equal(act_full, act_pruned)
    assert torch.equal(pruned_pre[keep], full_pre[keep]) and bool((pruned_pre[~keep] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("M,F,K,tune", [(300, 96, 128, None), (2048, 1024, 256, None), (1024, 512, 128, (256, 256, 2, 8)),
                                        (1000, 320, 64, (128, 192, 2, 8))])
def test_gemm_dswiglu_epilogue(ops, M, F, K, tune):
    """d(act) GEMM with the SwiGLU backward fused into the epilogue (MTL_EPI_DSWIGLU) == plain GEMM + mtl_swiglu_bwd_rows on the
    interleaved layout, incl. gathered physical rows (same row map for the saved pairs and the output)"""
    lib = ops.lib()
    dy = torch.randn(M, K, generator=g(1)).to(BF16).cuda()
    wp = (torch.randn(F, K, generator=g(2)) * 0.2).to(BF16).cuda()
    gu = torch.randn(M, 2 * F, generator=g(3)).to(BF16).cuda()
    dh = ops.gemm_nt(dy, wp)
    ref = ops.swiglu_bwd(gu, dh, interleaved=True)
    out = torch.zeros(M, 2 * F, dtype=BF16, device="cuda")
    with ops.gemm_tune(*(tune or ())):
        ops.gemm_nt(dy, wp, out=out, epilogue=ops.N.EPI_DSWIGLU, aux_in=gu)
    assert rel_err(out.float(), ref.float()) < 1e-3                        # same roundings; __expf vs expf in the sigmoid
    # pruned backward: logical row m lives at physical row (m // 60) * 100 + 40 + m % 60 of the saved pairs and of the output
    if M == 300:
        rows = (60, 100, 40)
        phys = torch.arange(M).div(60, rounding_mode="floor") * 100 + 40 + torch.arange(M) % 60
        gu_p = torch.zeros(500, 2 * F, dtype=BF16, device="cuda"); gu_p[phys.cuda()] = gu
        out_p = torch.zeros(500, 2 * F, dtype=BF16, device="cuda")
        ops.gemm_nt(dy, wp, out=out_p, epilogue=ops.N.EPI_DSWIGLU, aux_in=gu_p, c_rows=rows)
        assert torch.equal(out_p[phys.cuda()], out)
        mask = torch.ones(500, dtype=torch.bool); mask[phys] = False
        assert not out_p[mask.cuda()].any()


@pytest.mark.gpu
@pytest.mark.parametrize("Tq,Tk,kv0,pdrop", [(128, 256, 128, 0.1), (128, 256, 128, 0.0), (128, 128, 0, 0.1), (100, 100, 0, 0.0), (48, 112, 64, 0.5), (128, 256, 0, 0.1)])
def test_attention_backward_one_launch_equals_two(ops, Tq, Tk, kv0, pdrop):
    """The resident backward of MHA heads of width 64 as ONE launch (attn_bwd_res_merged_kernel: K, V, Q, dO, O of the head staged once; waves 0-7
    take one query tile each for dQ while waves 8-15 take one key tile each for dK / dV) against the two launches it replaces (dQ, then dK / dV): bit-identical dq, dk, dv on the
    rows they write — the pruned shape of the metric workload's backward (128 query rows at offset 128 of 256 keys, dK / dV for keys >= 128),
    full squares, ragged tiles, heavy dropout; and the last case (16 key tiles) must keep taking the two-launch path."""
    B, H, D, seed = 3, 4, 64, 4242
    coff = Tk - Tq
    q = dev(torch.randn(B, Tq, H * D, generator=g(1)).to(BF16))
    k, v = (dev(torch.randn(B, Tk, H * D, generator=g(i)).to(BF16)) for i in (2, 3))
    do = dev(torch.randn(B, Tq, H * D, generator=g(4)).to(BF16))
    scale = 1.0 / math.sqrt(D)
    o, lse = ops.attention_fwd(q, k, v, H, H, D, scale, True, dropout=(pdrop, seed), causal_off=coff)
    outs = []
    for merged in (1, 0):
        with ops.attention_tune(merged=bool(merged)):
            outs.append(ops.attention_bwd(q, k, v, o, lse, do, H, H, D, scale, True, dropout=(pdrop, seed), causal_off=coff, kv_row0=kv0))
    (dq1, dk1, dv1), (dq0, dk0, dv0) = outs
    assert torch.equal(dq1, dq0)
    assert torch.equal(dk1[:, kv0:], dk0[:, kv0:]) and torch.equal(dv1[:, kv0:], dv0[:, kv0:])
    assert float(dq1.float().abs().max()) > 0 and float(dk1[:, kv0:].float().abs().max()) > 0
    # and against the fp32 reference of the same (possibly pruned) problem
    qf, kf, vf = (t.float().cpu().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qf.view(B, Tq, H, D), kf.view(B, Tk, H, D)) * scale
    mask = torch.arange(Tk)[None, :] > (torch.arange(Tq)[:, None] + coff)
    p = torch.softmax(s.masked_fill(mask, float("-inf")), -1)
    if pdrop > 0:
        from helpers import drop_mult_attention
        p = p * drop_mult_attention(seed, pdrop, B, H, Tk, Tk)[:, :, coff:, :]
    ref = torch.einsum("bhqk,bkhd->bqhd", p, vf.view(B, Tk, H, D)).reshape(B, Tq, H * D)
    ref.backward(do.float().cpu())
    assert rel_err(dq1.float().cpu(), qf.grad) < TOL_ATTN_BWD
    assert rel_err(dk1[:, kv0:].float().cpu(), kf.grad[:, kv0:]) < TOL_ATTN_BWD and rel_err(dv1[:, kv0:].float().cpu(), vf.grad[:, kv0:]) < TOL_ATTN_BWD
