"""ctypes binding of libmedtsllm_hip.so (C-ABI declared in include/medtsllm_hip.h).

The product path has NO CPU / PyTorch fallback: if the library is missing or a call fails we raise.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first so our kernels share torch's HIP runtime instance)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTL_LIB_PATH") or os.path.join(_HERE, "libmedtsllm_hip.so")   # override: diagnostic builds only

MTL_F32, MTL_BF16 = 0, 1
EPI_STORE, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_ACCUM, EPI_SWIGLU, EPI_DSWIGLU = 0, 1, 2, 3, 4, 5, 6
ARCH_GPT2, ARCH_LLAMA = 0, 1
ABI_VERSION = 15

i64, vp, f32, i32 = C.c_int64, C.c_void_p, C.c_float, C.c_int


class MtlError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64), ("c_dtype", i32),
                ("M", i64), ("N", i64), ("K", i64),
                ("a_group_rows", i64), ("a_group_stride", i64), ("a_row_offset", i64),
                ("c_group_rows", i64), ("c_group_stride", i64), ("c_row_offset", i64),
                ("bias", vp), ("epilogue", i32), ("aux_in", vp), ("ld_aux_in", i64), ("aux_out", vp), ("ld_aux_out", i64),
                ("alpha", f32), ("split_k", i32), ("workspace", vp), ("workspace_bytes", C.c_size_t),
                ("drop_p", f32), ("drop_seed", C.c_uint32), ("bwd_group_rows", i64), ("bwd_first_row", i64),
                ("tune_mode", i32), ("tune_bm", i32), ("tune_bn", i32), ("tune_stages", i32), ("tune_waves", i32)]


class GemmXtArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("a_trans", i32), ("B", vp), ("ldb", i64), ("b_trans", i32), ("C", vp), ("ldc", i64), ("c_dtype", i32),
                ("M", i64), ("N", i64), ("K", i64), ("alpha", f32), ("a_colsum", vp), ("split_k", i32), ("workspace", vp),
                ("workspace_bytes", C.c_size_t)]


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("kind", C.c_int32), ("launches", i64), ("total_ms", C.c_double), ("min_ms", C.c_double),
                ("max_ms", C.c_double), ("total_work", C.c_double)]


class BackboneDropout(C.Structure):
    _fields_ = [("attn_p", f32), ("resid_p", f32), ("seed", C.c_uint32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("shadow", vp), ("cols", i64), ("ld_shadow", i64), ("param_dtype", i64)]


ADAM_MAX_TENSORS = 24


class AttnFwdArgs(C.Structure):
    _fields_ = [("q", vp), ("q_bs", i64), ("q_ts", i64), ("q_hs", i64),
                ("k", vp), ("k_bs", i64), ("k_ts", i64), ("k_hs", i64),
                ("v", vp), ("v_bs", i64), ("v_ts", i64), ("v_hs", i64),
                ("o", vp), ("o_bs", i64), ("o_ts", i64), ("o_hs", i64),
                ("lse", vp), ("B", i64), ("Hq", i64), ("Hkv", i64), ("Tq", i64), ("Tk", i64), ("D", i64),
                ("scale", f32), ("causal", i32), ("causal_off", i64), ("stat_stride", i64),
                ("dropout_p", f32), ("dropout_seed", C.c_uint32), ("o_f32", vp), ("tune", i32)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("f", AttnFwdArgs),
                ("dout", vp), ("do_bs", i64), ("do_ts", i64), ("do_hs", i64),
                ("dq", vp), ("dq_bs", i64), ("dq_ts", i64), ("dq_hs", i64),
                ("dk", vp), ("dk_bs", i64), ("dk_ts", i64), ("dk_hs", i64),
                ("dv", vp), ("dv_bs", i64), ("dv_ts", i64), ("dv_hs", i64),
                ("delta", vp), ("kv_row0", i64), ("dkv_ws", vp), ("kv_splits", i64), ("rope_cos", vp), ("rope_sin", vp)]


PP = C.POINTER(vp)


class BackboneWeights(C.Structure):
    _fields_ = [("arch", i32), ("n_layers", i32),
                ("d", i64), ("n_heads", i64), ("n_kv_heads", i64), ("head_dim", i64), ("ffn", i64), ("eps", f32),
                ("w_qkv", PP), ("w_qkv_t", PP), ("b_qkv", PP),
                ("w_o", PP), ("w_o_t", PP), ("b_o", PP),
                ("w_fc", PP), ("w_fc_t", PP), ("b_fc", PP),
                ("w_proj", PP), ("w_proj_t", PP), ("b_proj", PP),
                ("ln1_w", PP), ("ln1_b", PP), ("ln2_w", PP), ("ln2_b", PP),
                ("lnf_w", vp), ("lnf_b", vp), ("rope_cos", vp), ("rope_sin", vp), ("stream_dtype", i32)]


# name -> (restype, argtypes); every symbol include/medtsllm_hip.h declares
SIGNATURES = {
    "mtl_abi_version": (i32, []),
    "mtl_build_flags": (i32, []),
    "mtl_strerror": (C.c_char_p, [i32]),
    "mtl_patch_tokenize_fwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, f32, f32, C.c_uint32, vp]),
    "mtl_patch_tokenize_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, f32, C.c_uint32, vp]),
    "mtl_patch_index_map": (i32, [vp, i64, i64, i64, vp]),
    "mtl_revin_denorm": (i32, [vp, i32, vp, vp, vp, i32, i64, i64, i64, vp]),
    "mtl_gemm_workspace_bytes": (C.c_size_t, [i64, i64, i32]),
    "mtl_gemm_auto_split_k": (i32, [i64, i64, i64, i32]),
    "mtl_gemm_xt_workspace_bytes": (C.c_size_t, [i64, i64, i32]),
    "mtl_gemm_xt_auto_split_k": (i32, [i64, i64, i64]),
    "mtl_gemm_xt": (i32, [C.POINTER(GemmXtArgs), vp]),
    "mtl_gemm_nt": (i32, [C.POINTER(GemmArgs), vp]),
    "mtl_prof_enable": (i32, [i32]),
    "mtl_gemm_tile_order": (i32, [i32, i32, i32, i32, i32, i64, i32]),
    "mtl_prof_read": (i32, [C.POINTER(ProfRow), i32]),
    "mtl_cast_pad_f32_bf16": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, vp]),
    "mtl_transpose_bf16": (i32, [vp, i64, vp, i64, i64, i64, vp]),
    "mtl_transpose_colsum_bf16": (i32, [vp, i64, vp, i64, vp, i64, i64, vp]),
    "mtl_cast_f32_to_bf16": (i32, [vp, vp, i64, vp]),
    "mtl_cast_bf16_to_f32": (i32, [vp, vp, i64, vp]),
    "mtl_colsum_bf16": (i32, [vp, i64, vp, i64, i64, vp]),
    "mtl_rowsum_bf16": (i32, [vp, i64, vp, i64, i64, vp]),
    "mtl_channel_mix_fwd": (i32, [vp, vp, vp, vp, i32, i64, i64, i64, i64, i64, vp]),
    "mtl_channel_mix_workspace_bytes": (C.c_size_t, [i64, i64, i64]),
    "mtl_channel_mix_bwd": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "mtl_adam_step": (i32, [C.POINTER(AdamTensor), i32, f32, f32, f32, f32, f32, i32, i64, vp]),
    "mtl_attention_fwd": (i32, [C.POINTER(AttnFwdArgs), vp]),
    "mtl_attention_bwd": (i32, [C.POINTER(AttnBwdArgs), vp]),
    "mtl_norm_fwd": (i32, [vp, vp, vp, vp, i64, vp, i64, i64, f32, i32, i64, i64, i64, i32, vp]),
    "mtl_norm_bwd": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, i64, i64, i32, i64, i64, i64, i32, f32, C.c_uint32, vp]),
    "mtl_norm_fwd_t": (i32, [vp, i32, vp, vp, vp, i64, vp, i64, i64, f32, i32, i64, i64, i64, i32, vp]),
    "mtl_norm_bwd_t": (i32, [vp, i64, vp, i32, vp, vp, vp, vp, vp, i64, i64, i32, i64, i64, i64, i32, f32, C.c_uint32, vp]),
    "mtl_dropout_f32": (i32, [vp, vp, i64, i64, f32, C.c_uint32, vp]),
    "mtl_rope_inplace": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "mtl_rope_inplace_rows": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i32, i64, i64, i64, vp]),
    "mtl_swiglu_bwd_rows": (i32, [vp, vp, vp, i64, i64, i64, i64, i64, i32, vp]),
    "mtl_swiglu_fwd": (i32, [vp, vp, i64, i64, vp]),
    "mtl_swiglu_bwd": (i32, [vp, vp, vp, i64, i64, vp]),
    "mtl_assemble_llm_input": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, f32, C.c_uint32, vp]),
    "mtl_assemble_bwd": (i32, [vp, vp, i64, i64, i64, i64, f32, C.c_uint32, vp]),
    "mtl_assemble_llm_input_t": (i32, [vp, i64, vp, vp, vp, vp, i32, i64, i64, i64, i64, f32, C.c_uint32, vp]),
    "mtl_assemble_bwd_t": (i32, [vp, i32, vp, i64, i64, i64, i64, f32, C.c_uint32, vp]),
    "mtl_input_stats_workspace_bytes": (C.c_size_t, [i64, i64, i64]),
    "mtl_input_stats": (i32, [vp, vp, vp, vp, C.c_size_t, i64, i64, i64, i64, i64, vp]),
    "mtl_backbone_saved_bytes": (C.c_size_t, [C.POINTER(BackboneWeights), i64, i64]),
    "mtl_backbone_saved_hidden_offset": (C.c_size_t, [C.POINTER(BackboneWeights), i64, i64, i32]),
    "mtl_backbone_work_bytes": (C.c_size_t, [C.POINTER(BackboneWeights), i64, i64]),
    "mtl_backbone_fwd": (i32, [C.POINTER(BackboneWeights), vp, vp, vp, vp, i64, i64, i64, i64, C.POINTER(BackboneDropout), vp, i64, vp]),
    "mtl_backbone_prefix_bytes": (C.c_size_t, [C.POINTER(BackboneWeights), i64]),
    "mtl_backbone_prefix_build": (i32, [C.POINTER(BackboneWeights), vp, vp, vp, vp, i64, vp]),
    "mtl_backbone_bwd": (i32, [C.POINTER(BackboneWeights), vp, vp, vp, vp, vp, i64, i64, i64, i64, C.POINTER(BackboneDropout), vp]),
}

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Load (once) and return the ctypes library; raises MtlError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MtlError(f"{LIB_PATH} not found: the HIP extension is required (run `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` or `make -C med-ts-llm_amd/csrc`). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        if l.mtl_abi_version() != ABI_VERSION:
            raise MtlError(f"ABI mismatch: library {l.mtl_abi_version()} vs binding {ABI_VERSION}")
        flags = l.mtl_build_flags()
        if flags and os.environ.get("MTL_ALLOW_DIAG_LIB") != "1":
            # a diagnostic build (environment switches compiled in, or timing ablations that compute wrong results on purpose) must never
            # stand in for the product library by accident: tools/ set MTL_ALLOW_DIAG_LIB=1 next to MTL_LIB_PATH
            raise MtlError(f"{LIB_PATH} is a diagnostic build (mtl_build_flags() = {flags}); set MTL_ALLOW_DIAG_LIB=1 to load it from a tool")
        _lib = l
    return _lib


def prof_rows(cap=256):
    """launch-profiler records since mtl_prof_enable(1): [{kernel, kind ('flops' | 'bytes'), launches, total_ms, min_ms, max_ms, work}]"""
    rows = (ProfRow * cap)()
    n = lib().mtl_prof_read(rows, cap)
    if n < 0:
        check(n, "mtl_prof_read")
    return [{"kernel": rows[i].name.decode(), "kind": "flops" if rows[i].kind == 0 else "bytes", "launches": int(rows[i].launches),
             "total_ms": rows[i].total_ms, "min_ms": rows[i].min_ms, "max_ms": rows[i].max_ms, "work": rows[i].total_work}
            for i in range(n) if rows[i].launches > 0]


def check(code, what=""):
    if code != 0:
        raise MtlError(f"{what}: {lib().mtl_strerror(code).decode()} (code {code})")


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
