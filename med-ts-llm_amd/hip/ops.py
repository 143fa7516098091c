"""Raw wrappers over the C-ABI (tensors in, tensors out) and the torch.autograd.Function ops built on them.

PyTorch here is plumbing only: device memory (caching allocator), the current stream and autograd bookkeeping.
All device arithmetic of the hot path runs in libmedtsllm_hip.so; nothing in this file falls back to ATen math for
a kernel that failed to load — `_native.lib()` raises instead.
"""
import contextlib
import ctypes as C
import math
import os

import torch

from . import _native as N
from ._native import ptr, stream, check, lib

BF16, F32 = torch.bfloat16, torch.float32


def pad64(n):
    return (int(n) + 63) // 64 * 64


def pad_vocab(n):
    """K padding of the mapping GEMM operands: a multiple of 64 * 16, so that every power-of-two split-K up to 16 divides
    the k-steps evenly (persistent split-K path of mtl_gemm_nt); +1.9 % zero columns for GPT-2's 50 257 + 1."""
    return (int(n) + 1023) // 1024 * 1024


def mapping_split_k(rows, d, kp):
    """power-of-two split of the mapping GEMM's K = kp so that ~2 work items per CU exist"""
    tiles = ((rows + 255) // 256) * ((d + 191) // 192)            # the split path prefers 256x192 work items
    want = max(1, min(16, (512 + tiles - 1) // tiles, kp // 64 // 16))
    s = 1
    while s * 2 <= want and (kp // 64) % (s * 2) == 0:
        s *= 2
    return s


def _dt(t):
    return N.MTL_BF16 if t.dtype == BF16 else N.MTL_F32


def _req(cond, msg):
    if not cond:
        raise ValueError(msg)


# =============================================================================================== raw wrappers
# ---- per-call kernel selection for A/B runs and tests. The LIBRARY keeps no such state (mtl_gemm_args.tune_*, mtl_attn_fwd_args.tune travel
# with every call); these context managers only decide what this module puts into the calls it builds while they are active.
_TUNE = {"gemm": (0, 0, 0, 0, 0), "attn": 0}


@contextlib.contextmanager
def gemm_tune(bm=0, bn=0, stages=0, waves=0, one_tile=False):
    """force one tile configuration for the gemm_nt calls inside the block: bm x bn tile, LDS ring depth, waves per workgroup (0 = automatic);
    one_tile = the one-output-tile-per-workgroup kernel instead of the persistent one"""
    old = _TUNE["gemm"]
    _TUNE["gemm"] = (1 if one_tile else (2 if (bm or bn or stages or waves) else 0), bm, bn, stages, waves)
    try:
        yield
    finally:
        _TUNE["gemm"] = old


@contextlib.contextmanager
def attention_tune(resident=True, merged=True):
    """resident=False: the chunked attention kernels even where K / V fit the LDS; merged=False: the resident backward as two launches"""
    old = _TUNE["attn"]
    _TUNE["attn"] = (0 if resident else 1) | (0 if merged else 2)
    try:
        yield
    finally:
        _TUNE["attn"] = old


def gemm_nt(A, B, out=None, out_dtype=BF16, bias=None, epilogue=N.EPI_STORE, aux_in=None, aux_out=None, alpha=1.0,
            split_k=0, M=None, a_rows=None, c_rows=None, drop=(0.0, 0), bwd_rows=None):
    """C[M,N] = epi(alpha * A[M,K] @ B[N,K]^T + bias). A, B bf16 with unit inner stride, K % 64 == 0.
    split_k = 0: the library's own choice (mtl_gemm_auto_split_k: > 1 only for few output tiles with a long K)."""
    _req(A.dtype == BF16 and B.dtype == BF16 and A.dim() == 2 and B.dim() == 2, "gemm_nt: bf16 2-D operands")
    _req(A.stride(1) == 1 and B.stride(1) == 1 and A.shape[1] == B.shape[1], "gemm_nt: K mismatch / inner stride")
    K, Nn = A.shape[1], B.shape[0]
    M = A.shape[0] if M is None else M
    if out is None:
        out = torch.empty((M, Nn), dtype=out_dtype, device=A.device)
    g = N.GemmArgs()
    g.A, g.lda, g.B, g.ldb = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0)
    g.C, g.ldc, g.c_dtype = out.data_ptr(), out.stride(0), _dt(out)
    g.M, g.N, g.K = M, Nn, K
    if a_rows is not None:
        g.a_group_rows, g.a_group_stride, g.a_row_offset = a_rows
    if c_rows is not None:
        g.c_group_rows, g.c_group_stride, g.c_row_offset = c_rows
    g.bias = bias.data_ptr() if bias is not None else None
    g.epilogue = epilogue
    if aux_in is not None:
        g.aux_in, g.ld_aux_in = aux_in.data_ptr(), aux_in.stride(0)
    if aux_out is not None:
        g.aux_out, g.ld_aux_out = aux_out.data_ptr(), aux_out.stride(0)
    if split_k == 0:
        split_k = lib().mtl_gemm_auto_split_k(M, Nn, K, epilogue)
    g.alpha, g.split_k = alpha, split_k
    g.tune_mode, g.tune_bm, g.tune_bn, g.tune_stages, g.tune_waves = _TUNE["gemm"]
    g.drop_p, g.drop_seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF      # MTL_EPI_RESID only (resid_pdrop)
    if bwd_rows is not None:       # GELU / SWIGLU: (group_rows, first_row) of the rows whose backward-only output is stored
        g.bwd_group_rows, g.bwd_first_row = bwd_rows
    ws = None
    if split_k > 1:
        nbytes = lib().mtl_gemm_workspace_bytes(M, Nn, split_k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), nbytes
    check(lib().mtl_gemm_nt(C.byref(g), stream()), "mtl_gemm_nt")
    return out


def gemm_xt(A, B, a_trans=False, b_trans=False, out_dtype=BF16, alpha=1.0, want_colsum=False, split_k=0):
    """C[M, N] = alpha * sum_k A(m, k) B(n, k) with K-MAJOR operands where flagged: A is [M, K] (a_trans False) or [K, M] (True),
    B is [N, K] or [K, N]; rows contiguous, row strides % 8 == 0. want_colsum (a_trans only): also returns sum_k A(m, k) as f32 [M].
    The backward GEMMs of the trainable Linear layers (no transposed copies): see mtl_gemm_xt in include/medtsllm_hip.h."""
    _req(A.dtype == BF16 and B.dtype == BF16 and A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1, "gemm_xt: bf16 2-D row-major operands")
    _req(a_trans or b_trans, "gemm_xt: at least one K-major operand (else gemm_nt)")
    K, M = (A.shape[0], A.shape[1]) if a_trans else (A.shape[1], A.shape[0])
    Kb, Nn = (B.shape[0], B.shape[1]) if b_trans else (B.shape[1], B.shape[0])
    _req(K == Kb, "gemm_xt: K mismatch")
    out = torch.empty((M, Nn), dtype=out_dtype, device=A.device)
    g = N.GemmXtArgs()
    g.A, g.lda, g.a_trans = A.data_ptr(), A.stride(0), int(a_trans)
    g.B, g.ldb, g.b_trans = B.data_ptr(), B.stride(0), int(b_trans)
    g.C, g.ldc, g.c_dtype = out.data_ptr(), out.stride(0), _dt(out)
    g.M, g.N, g.K, g.alpha = M, Nn, K, alpha
    cs = torch.empty((M,), dtype=F32, device=A.device) if want_colsum else None
    g.a_colsum = cs.data_ptr() if cs is not None else None
    if split_k == 0:
        split_k = lib().mtl_gemm_xt_auto_split_k(M, Nn, K)
    g.split_k = split_k
    ws = None
    if split_k > 1:
        nbytes = lib().mtl_gemm_xt_workspace_bytes(M, Nn, split_k)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), nbytes
    check(lib().mtl_gemm_xt(C.byref(g), stream()), "mtl_gemm_xt")
    return (out, cs) if want_colsum else out


def cast_pad(src, ld_dst=None, want_t=False, ld_dst_t=None, dst=None, dst_t=None):
    """f32 [R, Cc] -> bf16 [R, ld_dst] (zero padded) and optionally the transpose bf16 [Cc, ld_dst_t]."""
    _req(src.dtype == F32 and src.dim() == 2 and src.stride(1) == 1, "cast_pad: f32 2-D")
    R, Cc = src.shape
    ld_dst = Cc if ld_dst is None else ld_dst
    if dst is None:
        dst = torch.empty((R, ld_dst), dtype=BF16, device=src.device)
    if want_t and dst_t is None:
        ld_dst_t = R if ld_dst_t is None else ld_dst_t
        dst_t = torch.empty((Cc, ld_dst_t), dtype=BF16, device=src.device)
    check(lib().mtl_cast_pad_f32_bf16(ptr(src), src.stride(0), ptr(dst), dst.stride(0), ptr(dst_t),
                                      dst_t.stride(0) if dst_t is not None else 0, R, Cc, stream()), "mtl_cast_pad_f32_bf16")
    return (dst, dst_t) if (want_t or dst_t is not None) else dst


def transpose_bf16(src, ld_dst=None):
    """bf16 [R, Cc] -> bf16 [Cc, ld_dst] with zero padding of columns >= R."""
    _req(src.dtype == BF16 and src.dim() == 2 and src.stride(1) == 1, "transpose_bf16: bf16 2-D")
    R, Cc = src.shape
    ld_dst = R if ld_dst is None else ld_dst
    dst = torch.empty((Cc, ld_dst), dtype=BF16, device=src.device)
    check(lib().mtl_transpose_bf16(ptr(src), src.stride(0), ptr(dst), ld_dst, R, Cc, stream()), "mtl_transpose_bf16")
    return dst


def transpose_colsum_bf16(src, ld_dst=None):
    """transpose_bf16 plus the fp32 column sums of src, one pass over src (dY^T for the dW GEMM + the bias gradient)"""
    _req(src.dtype == BF16 and src.dim() == 2 and src.stride(1) == 1, "transpose_colsum_bf16: bf16 2-D")
    R, Cc = src.shape
    ld_dst = R if ld_dst is None else ld_dst
    dst = torch.empty((Cc, ld_dst), dtype=BF16, device=src.device)
    sums = torch.empty(Cc, dtype=F32, device=src.device)
    check(lib().mtl_transpose_colsum_bf16(ptr(src), src.stride(0), ptr(dst), ld_dst, ptr(sums), R, Cc, stream()), "mtl_transpose_colsum_bf16")
    return dst, sums


def to_bf16(x):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib().mtl_cast_f32_to_bf16(ptr(x), ptr(out), x.numel(), stream()), "mtl_cast_f32_to_bf16")
    return out


def to_f32(x):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=F32, device=x.device)
    check(lib().mtl_cast_bf16_to_f32(ptr(x), ptr(out), x.numel(), stream()), "mtl_cast_bf16_to_f32")
    return out


def colsum(x):
    _req(x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1, "colsum: bf16 2-D")
    out = torch.empty(x.shape[1], dtype=F32, device=x.device)
    check(lib().mtl_colsum_bf16(ptr(x), x.stride(0), ptr(out), x.shape[0], x.shape[1], stream()), "mtl_colsum_bf16")
    return out


def rowsum(x):
    _req(x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1, "rowsum: bf16 2-D")
    out = torch.empty(x.shape[0], dtype=F32, device=x.device)
    check(lib().mtl_rowsum_bf16(ptr(x), x.stride(0), ptr(out), x.shape[0], x.shape[1], stream()), "mtl_rowsum_bf16")
    return out


def patch_index_map(L, patch_len, stride, device):
    P = (L + stride - patch_len) // stride + 1
    idx = torch.empty((P, patch_len), dtype=torch.int32, device=device)
    check(lib().mtl_patch_index_map(ptr(idx), L, patch_len, stride, stream()), "mtl_patch_index_map")
    return idx


def patch_tokenize_fwd(x, conv_w, patch_len, stride, concat, eps=1e-5, drop=(0.0, 0)):
    B, L, Cc = x.shape
    d_patch = conv_w.shape[0]
    P = (L + stride - patch_len) // stride + 1
    width = Cc * d_patch if concat else d_patch
    ld = pad64(width)
    rows = B if concat else B * Cc
    out = torch.empty((rows, P, ld), dtype=BF16, device=x.device)
    mean = torch.empty((B, Cc), dtype=F32, device=x.device)
    stdev = torch.empty((B, Cc), dtype=F32, device=x.device)
    check(lib().mtl_patch_tokenize_fwd(ptr(x), ptr(conv_w), ptr(out), ptr(mean), ptr(stdev), B, L, Cc, patch_len, stride,
                                       d_patch, ld, 1 if concat else 0, eps, float(drop[0]), int(drop[1]) & 0xFFFFFFFF, stream()), "mtl_patch_tokenize_fwd")
    return out, mean, stdev


def patch_tokenize_bwd(x, mean, stdev, dout, conv_w_shape, patch_len, stride, concat, drop=(0.0, 0)):
    B, L, Cc = x.shape
    d_patch = conv_w_shape[0]
    nw = d_patch * patch_len * 3
    partial = torch.empty((B * Cc, nw), dtype=F32, device=x.device)
    dw = torch.empty(conv_w_shape, dtype=F32, device=x.device)
    check(lib().mtl_patch_tokenize_bwd(ptr(x), ptr(mean), ptr(stdev), ptr(dout), ptr(partial), ptr(dw), B, L, Cc, patch_len,
                                       stride, d_patch, dout.shape[-1], 1 if concat else 0, float(drop[0]), int(drop[1]) & 0xFFFFFFFF, stream()), "mtl_patch_tokenize_bwd")
    return dw


def revin_denorm(y, mean, stdev, out_dtype=F32):
    """y f32 or bf16 [B,T,C] -> y * stdev (+ mean) as f32 or bf16; mean None -> multiply by stdev only (the backward)."""
    _req(y.dtype in (F32, BF16) and out_dtype in (F32, BF16), "revin_denorm: f32 / bf16")
    y = y.contiguous()
    B, T, Cc = y.shape
    out = torch.empty(y.shape, dtype=out_dtype, device=y.device)
    check(lib().mtl_revin_denorm(ptr(y), _dt(y), ptr(mean), ptr(stdev), ptr(out), _dt(out), B, T, Cc, stream()), "mtl_revin_denorm")
    return out


def input_stats(x, channel, n_lags):
    """x f32 [B, L, C] on the device -> ONE packed f32 [B, n_ch * 4 + n_lags] tensor: per selected channel (min, max, median, trend),
    then the top-n_lags autocorrelation lags; channel = -1 selects every channel (mtl_input_stats)."""
    x = x.contiguous().float()
    B, L, Cc = x.shape
    n_ch = Cc if channel < 0 else 1
    packed = torch.empty((2, B, max(n_ch * 4, n_lags)), dtype=F32, device=x.device)       # [0]: stats rows, [1]: lag rows
    nbytes = lib().mtl_input_stats_workspace_bytes(B, L, n_ch)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    stats, lags = packed[0].reshape(-1)[: B * n_ch * 4], packed[1].reshape(-1)[: B * n_lags]
    check(lib().mtl_input_stats(ptr(x), ptr(stats), ptr(lags), ptr(ws), nbytes, B, L, Cc, channel, n_lags, stream()), "mtl_input_stats")
    return stats.view(B, n_ch, 4), lags.view(B, n_lags), packed


def _attn_fwd_args(q, k, v, o, lse, B, Hq, Hkv, Tq, Tk, D, scale, causal, qs, ks, vs, os_, dropout=(0.0, 0)):
    a = N.AttnFwdArgs()
    a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF
    a.q, (a.q_bs, a.q_ts, a.q_hs) = q.data_ptr(), qs
    a.k, (a.k_bs, a.k_ts, a.k_hs) = k.data_ptr(), ks
    a.v, (a.v_bs, a.v_ts, a.v_hs) = v.data_ptr(), vs
    a.o, (a.o_bs, a.o_ts, a.o_hs) = o.data_ptr(), os_
    a.lse = lse.data_ptr()
    a.B, a.Hq, a.Hkv, a.Tq, a.Tk, a.D = B, Hq, Hkv, Tq, Tk, D
    a.scale, a.causal = scale, 1 if causal else 0
    a.tune = _TUNE["attn"]
    return a


def attention_fwd(q, k, v, Hq, Hkv, D, scale, causal, shared_kv=False, dropout=(0.0, 0), want_o32=False, causal_off=0):
    """q [B,Tq,Hq*D] bf16; k, v [B,Tk,Hkv*D] (or [Tk,Hkv*D] when shared_kv). Views with unit inner stride allowed.
    want_o32 (non-causal): also returns the fp32 output, from which the backward takes delta (pass it as attention_bwd's o32)."""
    B, Tq = q.shape[0], q.shape[1]
    Tk = k.shape[-2]
    o = torch.empty((B, Tq, Hq * D), dtype=BF16, device=q.device)
    o32 = torch.empty((B, Tq, Hq * D), dtype=F32, device=q.device) if (want_o32 and not causal) else None
    lse = torch.empty((B, Hq, Tq), dtype=F32, device=q.device)
    ks = (0, k.stride(-2), D) if shared_kv else (k.stride(0), k.stride(1), D)
    vs = (0, v.stride(-2), D) if shared_kv else (v.stride(0), v.stride(1), D)
    a = _attn_fwd_args(q, k, v, o, lse, B, Hq, Hkv, Tq, Tk, D, scale, causal, (q.stride(0), q.stride(1), D), ks, vs,
                       (o.stride(0), o.stride(1), D), dropout)
    a.o_f32 = o32.data_ptr() if o32 is not None else None
    a.causal_off = int(causal_off)          # query row i sits at absolute position i + causal_off (the last Tq rows of a longer sequence)
    check(lib().mtl_attention_fwd(C.byref(a), stream()), "mtl_attention_fwd")
    return (o, lse, o32) if want_o32 else (o, lse)


def attention_bwd(q, k, v, o, lse, dout, Hq, Hkv, D, scale, causal, shared_kv=False, dropout=(0.0, 0), dkv_out=None, o32=None, rope=None,
                  causal_off=0, kv_row0=0):
    """dkv_out = (dk, dv): write the key / value gradients there (row-strided views of one buffer are fine: the paired K/V
    projection backward then reads both as ONE operand)"""
    B, Tq = q.shape[0], q.shape[1]
    Tk = k.shape[-2]
    dout = dout.contiguous()
    dq = torch.empty((B, Tq, Hq * D), dtype=BF16, device=q.device)
    if dkv_out is not None:
        dk, dv = dkv_out
        _req(dk.shape == k.shape and dv.shape == v.shape and dk.stride(-1) == 1 and dv.stride(-1) == 1 and dk.stride(-2) == dv.stride(-2), "attention_bwd: dkv_out layout")
    else:
        dk = torch.empty(k.shape, dtype=BF16, device=q.device)
        dv = torch.empty(v.shape, dtype=BF16, device=q.device)
    delta = torch.empty((B, Hq, Tq), dtype=F32, device=q.device)
    ks = (0, k.stride(-2), D) if shared_kv else (k.stride(0), k.stride(1), D)
    vs = (0, v.stride(-2), D) if shared_kv else (v.stride(0), v.stride(1), D)
    b = N.AttnBwdArgs()
    b.f = _attn_fwd_args(q, k, v, o, lse, B, Hq, Hkv, Tq, Tk, D, scale, causal, (q.stride(0), q.stride(1), D), ks, vs,
                         (o.stride(0), o.stride(1), D), dropout)
    if o32 is not None:
        _req(o32.dtype == F32 and o32.shape == o.shape and o32.stride() == o.stride(), "attention_bwd: o32 layout")
        b.f.o_f32 = o32.data_ptr()
    b.dout, (b.do_bs, b.do_ts, b.do_hs) = dout.data_ptr(), (dout.stride(0), dout.stride(1), D)
    b.dq, (b.dq_bs, b.dq_ts, b.dq_hs) = dq.data_ptr(), (dq.stride(0), dq.stride(1), D)
    dks = (0, dk.stride(-2), D) if shared_kv else (dk.stride(0), dk.stride(1), D)
    b.dk, (b.dk_bs, b.dk_ts, b.dk_hs) = dk.data_ptr(), dks
    b.dv, (b.dv_bs, b.dv_ts, b.dv_hs) = dv.data_ptr(), dks
    b.delta = delta.data_ptr()
    b.f.causal_off, b.kv_row0 = int(causal_off), int(kv_row0)     # (pruned backward: dk / dv rows below kv_row0 are not written)
    if rope is not None:           # (cos, sin) f32 [positions, D]: inverse rotary embedding of dq / dk in the store epilogues
        b.rope_cos, b.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
    ws = None
    if shared_kv and B > 1:
        splits = max(1, min(B, 512 // max(1, ((Tk + 63) // 64) * Hkv)))      # ~2 workgroups per CU (measured: 512 / 1024 / 2048 workgroups -> 83 / 86 / 97 us)
        ws = torch.empty((splits, 2, Tk, Hkv * D), dtype=F32, device=q.device)
        b.dkv_ws, b.kv_splits = ws.data_ptr(), splits
    check(lib().mtl_attention_bwd(C.byref(b), stream()), "mtl_attention_bwd")
    return dq, dk, dv


def norm_fwd(x, gamma, beta, eps, rms=False, rows=None, M=None):
    """x f32 [Mx, d] -> y bf16 [M, d], stats f32 [M, 2]. rows=(group_rows, group_stride, row_offset) gathers."""
    d = x.shape[-1]
    M = x.shape[0] if M is None else M
    y = torch.empty((M, d), dtype=BF16, device=x.device)
    stats = torch.empty((M, 2), dtype=F32, device=x.device)
    gr, gs, ro = rows if rows is not None else (0, 0, 0)
    check(lib().mtl_norm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), d, ptr(stats), M, d, eps, 1 if rms else 0, gr, gs, ro, 0,
                             stream()), "mtl_norm_fwd")
    return y, stats


def norm_bwd(dy, x, gamma, stats, dres_in=None, rms=False, rows=None, want_bf16=False, dres_out=None, bf16_drop=(0.0, 0)):
    d = x.shape[-1]
    M = dy.shape[0]
    if dres_out is None:
        dres_out = torch.zeros_like(x) if rows is not None else torch.empty_like(x)
    dres_b = (torch.zeros(x.shape, dtype=BF16, device=x.device) if rows is not None
              else torch.empty(x.shape, dtype=BF16, device=x.device)) if want_bf16 else None
    gr, gs, ro = rows if rows is not None else (0, 0, 0)
    check(lib().mtl_norm_bwd(ptr(dy), dy.stride(0), ptr(x), ptr(gamma), ptr(stats), ptr(dres_in), ptr(dres_out), ptr(dres_b),
                             M, d, 1 if rms else 0, gr, gs, ro, 0, float(bf16_drop[0]), int(bf16_drop[1]) & 0xFFFFFFFF, stream()), "mtl_norm_bwd")
    return (dres_out, dres_b) if want_bf16 else dres_out


def rope_inplace(qkv, cos, sin, T, n_rot_heads, D, inverse=False):
    M = qkv.shape[0]
    check(lib().mtl_rope_inplace(ptr(qkv), qkv.stride(0), ptr(cos), ptr(sin), M, T, n_rot_heads, D, 1 if inverse else 0,
                                 stream()), "mtl_rope_inplace")
    return qkv


def swiglu_fwd(gu):
    M, F2 = gu.shape
    h = torch.empty((M, F2 // 2), dtype=BF16, device=gu.device)
    check(lib().mtl_swiglu_fwd(ptr(gu), ptr(h), M, F2 // 2, stream()), "mtl_swiglu_fwd")
    return h


def swiglu_bwd(gu, dh, interleaved=False):
    """interleaved: gu / dgu columns (2j, 2j+1) = (gate_j, up_j), the layout the MTL_EPI_SWIGLU GEMM epilogue writes"""
    M, F2 = gu.shape
    dgu = torch.empty_like(gu)
    check(lib().mtl_swiglu_bwd_rows(ptr(gu), ptr(dh), ptr(dgu), M, F2 // 2, 0, 0, 0, 1 if interleaved else 0, stream()), "mtl_swiglu_bwd_rows")
    return dgu


def assemble_llm_input(ids, embed, x_tok, wpe, drop=(0.0, 0), out_dtype=F32):
    """out_dtype: the residual stream's dtype (fp32; bf16 = the reference's setup.dtype "bf16")"""
    B, P, d = x_tok.shape
    n_tok = 0 if ids is None else ids.shape[1]
    h0 = torch.empty((B, n_tok + P, d), dtype=out_dtype, device=x_tok.device)
    check(lib().mtl_assemble_llm_input_t(ptr(ids), 0 if ids is None else ids.shape[0], ptr(embed), ptr(x_tok), ptr(wpe), ptr(h0), _dt(h0),
                                         B, n_tok, P, d, float(drop[0]), int(drop[1]) & 0xFFFFFFFF, stream()), "mtl_assemble_llm_input")
    return h0


def assemble_bwd(dh0, n_tok, drop=(0.0, 0)):
    """bf16 [B, P, d] = mask * dh0[:, n_tok:, :]"""
    dh0 = dh0.contiguous()
    B, T, d = dh0.shape
    out = torch.empty((B, T - n_tok, d), dtype=BF16, device=dh0.device)
    check(lib().mtl_assemble_bwd_t(ptr(dh0), _dt(dh0), ptr(out), B, n_tok, T - n_tok, d, float(drop[0]), int(drop[1]) & 0xFFFFFFFF, stream()), "mtl_assemble_bwd")
    return out


# =============================================================================================== autograd ops
class PatchTokenizeFn(torch.autograd.Function):
    """(x_enc f32 [B,L,C], conv_w) -> (tokens bf16 [B or B*C, P, K64], mean [B,C], stdev [B,C]).  a1-a4."""

    @staticmethod
    def forward(ctx, x, conv_w, patch_len, stride, concat, drop_p=0.0, drop_seed=0):
        """drop_p > 0: PatchEmbedding's train-mode dropout fused into the kernel (counter mask of (seed, row, column))"""
        x = x.contiguous().float()
        w = conv_w.detach().contiguous().float()
        out, mean, stdev = patch_tokenize_fwd(x, w, patch_len, stride, concat, drop=(drop_p, drop_seed))
        ctx.save_for_backward(x, mean, stdev)
        ctx.meta = (tuple(conv_w.shape), patch_len, stride, concat, (drop_p, drop_seed))
        ctx.mark_non_differentiable(mean, stdev)
        ctx.set_materialize_grads(False)         # (no zero-filled gradients for the two statistics outputs: two fill launches per step)
        return out, mean, stdev

    @staticmethod
    def backward(ctx, dout, _dm, _ds):
        if dout is None:
            return None, None, None, None, None, None, None
        x, mean, stdev = ctx.saved_tensors
        shape, patch_len, stride, concat, drop = ctx.meta
        dw = patch_tokenize_bwd(x, mean, stdev, dout.contiguous(), shape, patch_len, stride, concat, drop=drop)
        return None, dw, None, None, None, None, None


_LINEAR_XT = os.environ.get("MTL_LINEAR_XT", "1") != "0"      # A/B knob: 0 = backward through explicit transposes + NT GEMMs


def _check_shadow_unchanged(at):
    """`at` = (Bf16Shadow, its version at forward time) or None. The backward reads W from the shadow: an optimizer.step() between a
    forward and its backward (retain_graph, interleaved models, closure-style steps) would make dX use the UPDATED weights."""
    if at is not None and at[0].version != at[1]:
        raise RuntimeError("a trainable Linear's bf16 weight copy was rewritten (optimizer.step()) between this layer's forward and its "
                           "backward: the input gradient would be computed with the updated weights. Run backward before the optimiser step.")


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b with bf16 MFMA GEMMs. x bf16 [..., Kx] (Kx % 64 == 0, Kx >= W.shape[1], extra cols zero);
    W f32 [N, Kin] master weight (cast to bf16 per call, as autocast does); y bf16 [..., N].
    Backward on what the forward holds (mtl_gemm_xt): dX = dY . W with W read K-major, dW = dY^T . X with both operands K-major and
    the bias gradient as the column sums of the dY tiles that GEMM loads — no transposed copy of W, dY or X is written."""

    @staticmethod
    def forward(ctx, x, W, b, shadow=None):
        """shadow: hip.optim.Bf16Shadow of W with Kx columns (zero beyond Kin) that HipAdam keeps current — the per-call weight cast
        then only runs when the master changed behind the optimiser's back"""
        Kx = x.shape[-1]
        Nn, Kin = W.shape
        _req(x.dtype == BF16 and Kx % 64 == 0 and Kx >= Kin, "LinearFn: x must be bf16 with K padded to 64")
        x2 = x.reshape(-1, Kx)
        if x2.stride(1) != 1 or x2.stride(0) % 8 != 0:
            x2 = x2.contiguous()
        direct = Nn % 8 == 0 and _LINEAR_XT            # (row strides of dY must be whole 16-byte chunks for the K-major reads)
        use_sh = direct and shadow is not None and shadow.param is W and tuple(shadow.tensor.shape) == (Nn, Kx) and shadow.tensor.device == x.device
        def Wd():           # (only when a cast is due: with bf16 parameters — setup.dtype = "bf16" — .float() is a real pass over W)
            return W.detach().contiguous().float()
        wb = shadow.tensor if use_sh else torch.empty((Nn, Kx), dtype=BF16, device=x.device)
        wt = None
        if use_sh:
            if not shadow.fresh():
                cast_pad(Wd(), dst=wb)
                shadow.version = W._version
        elif direct:
            cast_pad(Wd(), dst=wb)
        else:
            Np = pad64(Nn)
            wt = torch.zeros((Kx, Np), dtype=BF16, device=x.device) if (Kx > Kin) else torch.empty((Kx, Np), dtype=BF16, device=x.device)
            cast_pad(Wd(), dst=wb, dst_t=wt)
        y = gemm_nt(x2, wb, bias=None if b is None else b.detach().float().contiguous())
        ctx.save_for_backward(x2, wb if direct else wt)
        ctx.meta = (tuple(x.shape), Nn, Kin, b is not None, direct)
        ctx.w_dtype = W.dtype if W.dtype in (F32, BF16) else F32      # the weight gradient leaves in the parameter's dtype (no autograd cast pass)
        # the shadow is a PERSISTENT tensor that HipAdam rewrites through raw pointers (autograd's version counter never sees it):
        # remember which weight version this forward multiplied by, so that a backward after an optimiser step fails loudly
        ctx.shadow_at = (shadow, shadow.version) if use_sh else None
        return y.reshape(*x.shape[:-1], Nn)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        xshape, Nn, Kin, has_b, direct = ctx.meta
        _check_shadow_unchanged(ctx.shadow_at)
        M, Kx = x2.shape
        dy2 = dy.reshape(M, Nn)
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dy2 = dy2.contiguous()
        dx = dW = db = None
        want_b = has_b and ctx.needs_input_grad[2]
        if direct:
            if ctx.needs_input_grad[0]:
                dx = gemm_xt(dy2, w, b_trans=True).reshape(xshape)                     # [M, Kx] = dY [M, N] . W [N, Kx]
            if ctx.needs_input_grad[1]:
                xk = x2[:, :Kin] if Kin != Kx else x2
                res = gemm_xt(dy2, xk, a_trans=True, b_trans=True, out_dtype=ctx.w_dtype, want_colsum=want_b)   # [N, Kin] = dY^T . X
                dW, db = res if want_b else (res, None)
            elif want_b:
                db = colsum(dy2)
            return dx, dW, db, None
        Np = w.shape[1]
        dyp = dy2 if Np == Nn else torch.nn.functional.pad(dy2, (0, Np - Nn))
        dx = gemm_nt(dyp, w).reshape(xshape) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1]:
            Mp = pad64(M)
            if want_b:
                dyT, db = transpose_colsum_bf16(dy2, Mp)   # [N, Mp] and the bias gradient from the same pass over dy
            else:
                dyT = transpose_bf16(dy2, Mp)        # [N, Mp]
            xT = transpose_bf16(x2[:, :Kin] if Kin != Kx else x2, Mp)  # [Kin, Mp]
            dW = gemm_nt(dyT, xT, out_dtype=F32)     # [N, Kin] fp32
        elif want_b:
            db = colsum(dy2)
        return dx, dW, db, None


class LinearPairFn(torch.autograd.Function):
    """(y1, y2) = (x W1^T + b1, x W2^T + b2): the key and value projections of the reprogramming layer (R:models/medtsllm.py:572-573),
    which share their input. Forward = two GEMMs; the BACKWARD treats [W1; W2] as one weight — dX = [dY1 | dY2] . [W1; W2] and
    d[W1; W2] = [dY1 | dY2]^T . X are one mtl_gemm_xt launch each (instead of two + an add) — when the two weight shadows are halves of
    one buffer (`pair` = the [N1 + N2, Kx] base tensor) and the incoming gradients are halves of one buffer (CrossAttnFn's backward
    makes them so); anything else falls back to LinearFn's arithmetic per projection."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, sh1, sh2, pair):
        Kx = x.shape[-1]
        x2 = x.reshape(-1, Kx)
        if x2.stride(1) != 1 or x2.stride(0) % 8 != 0:
            x2 = x2.contiguous()
        N1, N2, Kin = W1.shape[0], W2.shape[0], W1.shape[1]
        _req(W2.shape[1] == Kin and x.dtype == BF16 and Kx % 64 == 0 and Kx >= Kin and N1 % 8 == 0 and N2 % 8 == 0, "LinearPairFn: shapes")
        _req(pair is not None and tuple(pair.shape) == (N1 + N2, Kx) and sh1.tensor.data_ptr() == pair.data_ptr()
             and sh2.tensor.data_ptr() == pair[N1:].data_ptr(), "LinearPairFn: the shadows must be the halves of `pair`")
        ys = []
        for W, b, sh in ((W1, b1, sh1), (W2, b2, sh2)):
            if not sh.fresh():
                cast_pad(W.detach().contiguous().float(), dst=sh.tensor)
                sh.version = W._version
            ys.append(gemm_nt(x2, sh.tensor, bias=None if b is None else b.detach().float().contiguous()))
        ctx.save_for_backward(x2, pair)
        ctx.meta = (tuple(x.shape), N1, N2, Kin, b1 is not None, b2 is not None)
        ctx.shadows_at = ((sh1, sh1.version), (sh2, sh2.version))
        return ys[0].reshape(*x.shape[:-1], N1), ys[1].reshape(*x.shape[:-1], N2)

    @staticmethod
    def backward(ctx, dy1, dy2):
        x2, pair = ctx.saved_tensors
        xshape, N1, N2, Kin, has_b1, has_b2 = ctx.meta
        for at in ctx.shadows_at:
            _check_shadow_unchanged(at)
        M, Kx = x2.shape
        dy1, dy2 = dy1.reshape(M, N1), dy2.reshape(M, N2)
        adjacent = (dy1.dtype == BF16 and dy2.dtype == BF16 and dy1.stride(1) == 1 and dy2.stride(1) == 1 and dy1.stride(0) == dy2.stride(0) == N1 + N2
                    and dy2.data_ptr() == dy1.data_ptr() + 2 * N1)
        if adjacent:
            dy = torch.as_strided(dy1, (M, N1 + N2), (N1 + N2, 1))
        else:
            dy = torch.cat([dy1.to(BF16), dy2.to(BF16)], dim=1)
        dx = gemm_xt(dy, pair, b_trans=True).reshape(xshape) if ctx.needs_input_grad[0] else None
        dW1 = dW2 = db1 = db2 = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            xk = x2[:, :Kin] if Kin != Kx else x2
            dW, db = gemm_xt(dy, xk, a_trans=True, b_trans=True, out_dtype=F32, want_colsum=True)
            dW1, dW2 = dW[:N1], dW[N1:]
            db1, db2 = (db[:N1] if has_b1 else None), (db[N1:] if has_b2 else None)
        return dx, dW1, db1, dW2, db2, None, None, None


def _whole_rounds_columns(S, V, device):
    """columns of an [S, V] output (256 x 256 tiles, one per CU at a time) that make up whole rounds of the device's CUs, when the tail
    round would be less than 35 % full; 0 = do not split"""
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    tiles_m, tiles_n = (S + 255) // 256, (V + 255) // 256
    rounds = tiles_m * tiles_n / ncu
    whole = int(rounds)
    if whole < 1 or not (0.0 < rounds - whole < 0.35):
        return 0
    n_main = (whole * ncu // tiles_m) * 256
    return n_main if 0 < n_main < V else 0


class MappingFn(torch.autograd.Function):
    """source[S, d] = Wmap[S, V] @ Wemb[V, d] + b[:, None]   (R:models/medtsllm.py:281), batch independent.

    The bias rides in the K padding: column V of the bf16 Wmap copy holds b and row V of Wemb^T holds ones.
    `emb` = FrozenEmbedding-like object with .wT bf16 [d, Vp] (ones at column V) and .w bf16 [V, d]."""

    @staticmethod
    def forward(ctx, Wmap, b, wT, w, split_k, shadow=None):
        S, V = Wmap.shape
        Vp = wT.shape[1]
        if shadow is not None:                      # hip.optim.Bf16Shadow kept current by HipAdam (columns > V stay 0)
            wm = shadow.tensor
            if not shadow.fresh():
                cast_pad(Wmap.detach().contiguous().float(), dst=wm)
                shadow.version = Wmap._version
        else:
            wm = torch.empty((S, Vp), dtype=BF16, device=Wmap.device)
            cast_pad(Wmap.detach().contiguous().float(), dst=wm)
        wm[:, V] = b.detach().to(BF16)
        src = gemm_nt(wm, wT, split_k=split_k)
        ctx.save_for_backward(w)
        ctx.meta = (S, V)
        ctx.w_dtype = Wmap.dtype if Wmap.dtype in (F32, BF16) else F32
        return src

    @staticmethod
    def backward(ctx, dsrc):
        (w,) = ctx.saved_tensors
        S, V = ctx.meta
        dsrc = dsrc.contiguous()
        dW = None
        if ctx.needs_input_grad[0]:                                                   # [S, V] = dsrc[S,d] @ Wemb[V,d]^T
            # (bf16 parameters: the gradient is a bf16 tensor — written once by the GEMM; 16-byte row alignment of a [S, 50257] bf16 matrix does
            #  not hold, so that case keeps the fp32 output + autograd's cast)
            odt = ctx.w_dtype if (ctx.w_dtype == F32 or V % 8 == 0) else F32
            dW = torch.empty((S, V), dtype=odt, device=dsrc.device)
            n_main = _whole_rounds_columns(S, V, dsrc.device)
            if n_main:
                # 788 tiles of 256 x 256 on 256 CUs are 3.08 rounds = the time of 4: the columns that fill whole rounds go in one launch,
                # the remainder in a second, short one with small tiles
                gemm_nt(dsrc, w[:n_main], out=dW[:, :n_main])
                gemm_nt(dsrc, w[n_main:], out=dW[:, n_main:])
            else:
                gemm_nt(dsrc, w, out=dW)
        db = rowsum(dsrc) if ctx.needs_input_grad[1] else None       # row sums of dsrc
        return dW, db, None, None, None, None


class MappingTrainableFn(torch.autograd.Function):
    """MappingFn for vocabularies > 100 000, where the reference turns the (linspace sub-sampled) word-embedding table
    into a TRAINABLE parameter (R:models/medtsllm.py:220-222): source = Wmap @ Wemb + b with gradients for all three.
    dWemb[V, d] = Wmap^T[V, S] @ dsource[S, d] — one more NT GEMM with fp32 output."""

    @staticmethod
    def forward(ctx, Wmap, b, Wemb, split_k, sh_map=None, sh_emb=None):
        """sh_map / sh_emb: hip.optim.Bf16Shadow of Wmap ([S, pad64(V + 1)], zero padding) and of Wemb ([V, d]) that HipAdam keeps current while
        it updates the masters (MedTsLLM.bf16_shadows). With fresh shadows the per-call fp32 -> bf16 casts of both tables (Llama-3: 2 GB read) become
        bf16 transposes of the shadows, and under a row-sharded optimiser step (parallel.ShardedUpdate) the shadows are what the ranks exchange:
        2 B per element on the wire instead of 4, and no rank needs the other ranks' fp32 rows."""
        S, V = Wmap.shape
        d = Wemb.shape[1]
        Vp, Sp = pad64(V + 1), pad64(S)
        dev = Wmap.device
        use_m = sh_map is not None and sh_map.param is Wmap and tuple(sh_map.tensor.shape) == (S, Vp) and sh_map.tensor.device == dev
        use_e = sh_emb is not None and sh_emb.param is Wemb and tuple(sh_emb.tensor.shape) == (V, d) and sh_emb.tensor.device == dev
        if use_m and sh_map.fresh():
            wm = sh_map.tensor
            wmT = transpose_bf16(wm[:, :V], Sp)
        else:
            wm = sh_map.tensor if use_m else torch.empty((S, Vp), dtype=BF16, device=dev)
            wmT = torch.empty((V, Sp), dtype=BF16, device=dev)
            cast_pad(Wmap.detach().contiguous().float(), dst=wm, dst_t=wmT)
            if use_m:
                sh_map.version = Wmap._version
        wm[:, V] = b.detach().to(BF16)
        if use_e and sh_emb.fresh():
            we = sh_emb.tensor
            weT = transpose_bf16(we, Vp)                    # [d, Vp], columns >= V zero
        else:
            we = sh_emb.tensor if use_e else torch.empty((V, d), dtype=BF16, device=dev)
            weT = torch.zeros((d, Vp), dtype=BF16, device=dev)
            cast_pad(Wemb.detach().contiguous().float(), dst=we, dst_t=weT)
            if use_e:
                sh_emb.version = Wemb._version
        weT[:, V] = 1.0
        src = gemm_nt(wm, weT, split_k=split_k)
        ctx.save_for_backward(we, wmT)
        ctx.meta = (S, V, d)
        # `we` may be the PERSISTENT shadow HipAdam rewrites: a backward after an optimiser step must fail loudly (see LinearFn)
        ctx.shadow_at = (sh_emb, sh_emb.version) if use_e else None
        return src

    @staticmethod
    def backward(ctx, dsrc):
        _check_shadow_unchanged(ctx.shadow_at)
        we, wmT = ctx.saved_tensors
        S, V, d = ctx.meta
        dsrc = dsrc.contiguous()
        dW = gemm_nt(dsrc, we, out_dtype=F32) if ctx.needs_input_grad[0] else None          # [S, V]
        dsT = transpose_bf16(dsrc, wmT.shape[1])                                               # [d, Sp]
        db = rowsum(dsrc) if ctx.needs_input_grad[1] else None              # row sums of dsrc
        dE = gemm_nt(wmT, dsT, out_dtype=F32) if ctx.needs_input_grad[2] else None            # [V, d]
        return dW, db, dE, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """Reprogramming attention (R:models/medtsllm.py:581-591): q [B,L,H*E], k/v [S,H*E] shared by every sample."""

    @staticmethod
    def forward(ctx, q, k, v, H, E, dropout_p=0.0, dropout_seed=0):
        """dropout_p > 0: A = dropout(softmax(.)) with a counter-based keep mask (seed chosen by the caller per step)."""
        scale = 1.0 / math.sqrt(E)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        drop = (float(dropout_p), int(dropout_seed))
        o, lse, o32 = attention_fwd(q, k, v, H, H, E, scale, causal=False, shared_kv=True, dropout=drop, want_o32=True)
        ctx.save_for_backward(q, k, v, o, lse, o32)
        ctx.meta = (H, E, scale, drop)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, o32 = ctx.saved_tensors
        H, E, scale, drop = ctx.meta
        # dK | dV side by side in one [S, 2 H E] buffer: LinearPairFn's backward reads them as one GEMM operand
        dkv = torch.empty((k.shape[0], 2 * H * E), dtype=BF16, device=q.device)
        dq, dk, dv = attention_bwd(q, k, v, o, lse, do, H, H, E, scale, causal=False, shared_kv=True, dropout=drop,
                                   dkv_out=(dkv[:, :H * E], dkv[:, H * E:]), o32=o32)
        return dq, dk, dv, None, None, None, None


class ChannelMixFn(torch.autograd.Function):
    """y[b, r, o] = bias[o] + sum_{j, c} W[o, j*C + c] x[b, c, r, j] for x bf16 [B, C, R, J] -> y [B, R, O] (bf16 or f32): the channel
    mean of the "add" / "independent" covariate modes (W None) and the feature_weighting Linear of "weighted-average" / "merge-end"
    on the channel-last view (R:models/medtsllm.py:286-291,371-375), without the fp32 copies and permutes."""

    @staticmethod
    def forward(ctx, x, W, bias, out_dtype):
        _req(x.dtype == BF16 and x.dim() == 4, "ChannelMixFn: x bf16 [B, C, R, J]")
        x = x.contiguous()
        B, Cc, R, J = x.shape
        O = 1 if W is None else W.shape[0]
        _req(W is None or (W.shape[1] == J * Cc), "ChannelMixFn: W [O, J*C]")
        Wf = None if W is None else W.detach().float().contiguous()
        bf = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty((B, R, O), dtype=out_dtype, device=x.device)
        check(lib().mtl_channel_mix_fwd(ptr(x), ptr(Wf), ptr(bf), ptr(y), _dt(y), B, Cc, R, J, O, stream()), "mtl_channel_mix_fwd")
        ctx.save_for_backward(x, Wf)
        ctx.meta = (W is not None, bias is not None, O)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wf = ctx.saved_tensors
        has_w, has_b, O = ctx.meta
        B, Cc, R, J = x.shape
        if dy.dtype not in (BF16, F32):
            dy = dy.float()
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        want_w, want_b = has_w and ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]
        dW = torch.empty((O, J * Cc), dtype=F32, device=x.device) if want_w else None
        db = torch.empty((O,), dtype=F32, device=x.device) if want_b else None
        ws = torch.empty(lib().mtl_channel_mix_workspace_bytes(Cc, J, O), dtype=torch.uint8, device=x.device) if (want_w or want_b) else None
        check(lib().mtl_channel_mix_bwd(ptr(x), ptr(Wf), ptr(dy), _dt(dy), ptr(dx), ptr(dW), ptr(db), ptr(ws), B, Cc, R, J, O, stream()), "mtl_channel_mix_bwd")
        return dx, dW, db, None


class AssembleFn(torch.autograd.Function):
    """h0 = cat[embed[ids], x_tok] (+ wpe) in fp32 (R:models/medtsllm.py:331-337,349; HF gpt2 :576-577)."""

    @staticmethod
    def forward(ctx, x_tok, ids, embed, wpe, drop_p=0.0, drop_seed=0, out_dtype=F32):
        """drop_p > 0: GPT-2's embd_pdrop fused into the assembly (same mask as EmbdDropoutFn on the assembled tensor)"""
        ctx.drop = (float(drop_p), int(drop_seed))
        h0 = assemble_llm_input(ids, embed, x_tok.contiguous(), wpe, ctx.drop, out_dtype)
        ctx.n_tok = 0 if ids is None else ids.shape[1]
        return h0

    @staticmethod
    def backward(ctx, dh0):
        return assemble_bwd(dh0, ctx.n_tok, ctx.drop), None, None, None, None, None, None


def dropout_f32(x, p, seed, out=None):
    """x f32 [..., d] -> dropout(x) with the library's counter-hash mask of (seed, flattened row, column)"""
    x = x.contiguous()
    out = torch.empty_like(x) if out is None else out
    check(lib().mtl_dropout_f32(ptr(x), ptr(out), x.numel() // x.shape[-1], x.shape[-1], float(p), int(seed) & 0xFFFFFFFF, stream()), "mtl_dropout_f32")
    return out


class EmbdDropoutFn(torch.autograd.Function):
    """GPT-2 embd_pdrop on inputs_embeds + wpe (HF:models/gpt2/modeling_gpt2.py:579); backward = the same mask on the gradient"""

    @staticmethod
    def forward(ctx, h0, p, seed):
        ctx.meta = (p, seed)
        return dropout_f32(h0, p, seed) if h0.dtype == F32 else dropout_f32(h0.float(), p, seed).to(h0.dtype)     # (bf16 stream: rare path, "examples")

    @staticmethod
    def backward(ctx, dh):
        p, seed = ctx.meta
        return (dropout_f32(dh, p, seed) if dh.dtype == F32 else dropout_f32(dh.float(), p, seed).to(dh.dtype)), None, None


class BackboneFn(torch.autograd.Function):
    """Frozen decoder stack (R:models/medtsllm.py:350) fwd + activation-gradient-only bwd, one C call each."""

    @staticmethod
    def forward(ctx, h0, backbone, n_last, n_grad=None, drop=None, prefix=None):
        """n_grad: number of trailing tokens per sample whose input gradient is consumed (the patch tokens). The text
        prompt rows before them never depend on a trainable parameter (causal attention), so their gradient is dead and
        the backward runs on B*n_grad rows only; dh0 is zero there. None -> full backward.
        prefix = (cache, n_prefix) from FrozenBackbone.prefix_cache: the first n_prefix rows are a constant prompt whose per-layer keys /
        values are cached; the forward runs on the remaining rows only (mtl_backbone_fwd)."""
        h0 = h0.contiguous()
        T = h0.shape[1]
        n_save = (T if n_grad is None else max(int(n_grad), n_last)) if ctx.needs_input_grad[0] else 0
        if prefix is not None and ctx.needs_input_grad[0] and (n_grad is None or n_save > T - prefix[1]):
            prefix = None                      # a backward over rows the cached forward would not compute: run the full forward
        out, saved = backbone.run_forward(h0, n_last, keep=ctx.needs_input_grad[0], drop=drop, n_save=n_save, prefix=prefix)
        ctx.backbone, ctx.n_last, ctx.saved, ctx.n_grad, ctx.drop = backbone, n_last, saved, n_grad, drop
        ctx.save_for_backward(h0)
        return out

    @staticmethod
    def backward(ctx, dout):
        (h0,) = ctx.saved_tensors
        dh0 = ctx.backbone.run_backward(h0, dout.contiguous(), ctx.saved, ctx.n_last, ctx.n_grad, drop=ctx.drop)
        ctx.saved = None
        return dh0, None, None, None, None, None


class RevinDenormFn(torch.autograd.Function):
    """y * stdev + mean (R:models/layers/RevIN.py:58-69); statistics are detached constants."""

    @staticmethod
    def forward(ctx, y, mean, stdev):
        ctx.save_for_backward(stdev)
        ctx.in_dtype = y.dtype
        return revin_denorm(y if y.dtype in (F32, BF16) else y.float(), mean, stdev)

    @staticmethod
    def backward(ctx, dout):
        (stdev,) = ctx.saved_tensors
        od = ctx.in_dtype if ctx.in_dtype in (F32, BF16) else F32
        g = revin_denorm(dout if dout.dtype in (F32, BF16) else dout.float(), None, stdev, out_dtype=od)
        return (g if od == ctx.in_dtype else g.to(ctx.in_dtype)), None, None
