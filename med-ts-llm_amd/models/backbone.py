"""Frozen GPT-2 / Llama backbone for the HIP path: weight preparation (once) + the two C-ABI stack calls.

Replaces `AutoModel.from_pretrained(...)` + `self.llm(inputs_embeds=...)` of the reference
(R:models/medtsllm.py:175-185,350). Weights come from a HuggingFace-format directory (config.json +
*.safetensors, read with `safetensors` directly) or from an in-memory state dict (random init for benchmarks).

HBM layout (prepared once, the backbone is frozen — R:models/medtsllm.py:231-233):
  per layer bf16 [out, in] row-major weights for the forward NT GEMMs AND their transposes [in, out] for the
  activation-gradient GEMMs (2x weight memory, trivially affordable in 288 GB; no per-step transposes);
  q/k/v fused into one [(Hq+2Hkv)*hd, d] matrix, Llama gate/up fused into [2*ffn, d];
  fp32 norm parameters / biases; fp32 embedding table (prompt gather) + bf16 [d, Vp] transposed table
  (mapping GEMM operand, bias-carrier ones in column V) + bf16 [V, d] (mapping dW operand).
"""
import ctypes as C
import json
import os

import torch

from ..hip import _native as N
from ..hip import ops

BF16, F32 = torch.bfloat16, torch.float32


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def load_hf_dir(path):
    """(config dict, state dict of CPU tensors) from a HuggingFace-format directory."""
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    sd = {}
    files = sorted(fn for fn in os.listdir(path) if fn.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    for fn in files:
        sd.update(load_file(os.path.join(path, fn)))
    sd = {(k[len("model."):] if k.startswith("model.") else k[len("transformer."):] if k.startswith("transformer.") else k): v
          for k, v in sd.items()}
    return cfg, sd


def _reject(what):
    raise NotImplementedError(f"backbone config option {what} changes the arithmetic and is not implemented by the HIP path "
                              "(loading it anyway would silently compute something else than HF does)")


def normalise_config(cfg):
    """Common view over GPT2Config / LlamaConfig json (SURVEY.md Appendix B constants). Options that change the numerics and
    that the kernels do not implement are REJECTED rather than ignored (the reference gets them applied by HF AutoModel)."""
    mt = cfg["model_type"]
    if mt == "gpt2":
        d, H = cfg["n_embd"], cfg["n_head"]
        if cfg.get("activation_function", "gelu_new") != "gelu_new":
            _reject(f"activation_function={cfg['activation_function']!r} (gelu_new only)")
        if cfg.get("scale_attn_by_inverse_layer_idx", False):
            _reject("scale_attn_by_inverse_layer_idx=true")
        if cfg.get("reorder_and_upcast_attn", False):
            _reject("reorder_and_upcast_attn=true")
        if not cfg.get("scale_attn_weights", True):
            _reject("scale_attn_weights=false")
        if cfg.get("add_cross_attention", False):
            _reject("add_cross_attention=true")
        return dict(arch="gpt2", n_layers=cfg["n_layer"], d=d, n_heads=H, n_kv_heads=H, head_dim=d // H,
                    ffn=cfg.get("n_inner") or 4 * d, eps=cfg.get("layer_norm_epsilon", 1e-5), vocab=cfg["vocab_size"],
                    n_positions=cfg.get("n_positions", 1024),
                    # GPT2Config defaults (HF:models/gpt2/configuration_gpt2.py): active whenever the model is in train mode
                    embd_pdrop=float(cfg.get("embd_pdrop", 0.1)), attn_pdrop=float(cfg.get("attn_pdrop", 0.1)),
                    resid_pdrop=float(cfg.get("resid_pdrop", 0.1)))
    if mt == "llama":
        d, H = cfg["hidden_size"], cfg["num_attention_heads"]
        rp = cfg.get("rope_parameters") or {}
        theta = cfg.get("rope_theta")
        if theta is None:
            theta = rp.get("rope_theta", 10000.0)
        scaling = cfg.get("rope_scaling") or ({k: v for k, v in rp.items() if k != "rope_theta"} if rp.get("rope_type", "default") != "default" else None)
        if scaling and scaling.get("rope_type", scaling.get("type", "default")) != "default":
            _reject(f"rope_scaling={scaling!r} (Llama-3.1-style frequency scaling; plain RoPE only)")
        if cfg.get("attention_bias", False):
            _reject("attention_bias=true")
        if cfg.get("mlp_bias", False):
            _reject("mlp_bias=true")
        if cfg.get("hidden_act", "silu") != "silu":
            _reject(f"hidden_act={cfg['hidden_act']!r} (silu only)")
        if float(cfg.get("attention_dropout", 0.0) or 0.0) != 0.0:
            _reject("attention_dropout > 0")
        return dict(arch="llama", n_layers=cfg["num_hidden_layers"], d=d, n_heads=H,
                    n_kv_heads=cfg.get("num_key_value_heads") or H, head_dim=cfg.get("head_dim") or d // H,
                    ffn=cfg["intermediate_size"], eps=cfg.get("rms_norm_eps", 1e-6), vocab=cfg["vocab_size"],
                    rope_theta=float(theta))
    raise ValueError(f"unsupported backbone model_type {mt!r} (HIP path implements gpt2 and llama)")


class FrozenBackbone:
    """Device-resident frozen stack. Not an nn.Module on purpose: nothing here is a parameter of the trainer."""

    def __init__(self, cfg, state_dict, device, n_layers=-1, stream_dtype=F32):
        """stream_dtype: dtype of the residual stream through the stack — fp32 (the reference's setup.dtype = "mixed" / "fp32") or bf16 (its
        setup.dtype = "bf16", R:tasks/base.py:261-262: the whole model is cast to bf16, so norm parameters, biases and position embeddings hold
        bf16 VALUES there too: they are rounded here, once, and kept as fp32 arrays for the kernels)"""
        if stream_dtype not in (F32, BF16):
            raise ValueError(f"residual stream dtype {stream_dtype}: fp32 or bf16")
        self.stream_dtype = stream_dtype
        self.cfg = c = normalise_config(cfg)
        if 0 < n_layers < c["n_layers"]:     # R:models/medtsllm.py:145-146 (llm_layers)
            c["n_layers"] = n_layers
        self.device = torch.device(device)
        self.arch = c["arch"]
        L, d = c["n_layers"], c["d"]
        sd = state_dict
        dev = self.device

        def bf(t):
            return t.to(dev, BF16).contiguous()

        def f32(t):
            t = t.to(dev, F32)
            return (t.to(BF16).to(F32) if stream_dtype == BF16 else t).contiguous()

        k = {n: [] for n in ("w_qkv", "w_qkv_t", "b_qkv", "w_o", "w_o_t", "b_o", "w_fc", "w_fc_t", "b_fc", "w_proj", "w_proj_t",
                             "b_proj", "ln1_w", "ln1_b", "ln2_w", "ln2_b")}
        for i in range(L):
            if self.arch == "gpt2":
                p = f"h.{i}."
                # HF Conv1D stores [in, out]  (HF:pytorch_utils.py:117-121): that IS the transposed copy
                for name, key in (("qkv", "attn.c_attn"), ("o", "attn.c_proj"), ("fc", "mlp.c_fc"), ("proj", "mlp.c_proj")):
                    w_io = sd[p + key + ".weight"]
                    k["w_" + name + "_t"].append(bf(w_io))
                    k["w_" + name].append(bf(w_io.t()))
                    k["b_" + name].append(f32(sd[p + key + ".bias"]))
                k["ln1_w"].append(f32(sd[p + "ln_1.weight"])); k["ln1_b"].append(f32(sd[p + "ln_1.bias"]))
                k["ln2_w"].append(f32(sd[p + "ln_2.weight"])); k["ln2_b"].append(f32(sd[p + "ln_2.bias"]))
            else:
                p = f"layers.{i}."
                wq = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0)
                # gate / up rows INTERLEAVED (2j = gate_j, 2j+1 = up_j): MTL_EPI_SWIGLU needs the pair in one lane's 4 columns
                wg = torch.stack([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], dim=1).reshape(-1, d)
                for name, w_oi in (("qkv", wq), ("o", sd[p + "self_attn.o_proj.weight"]), ("fc", wg), ("proj", sd[p + "mlp.down_proj.weight"])):
                    k["w_" + name].append(bf(w_oi))
                    k["w_" + name + "_t"].append(bf(w_oi.t()))
                k["ln1_w"].append(f32(sd[p + "input_layernorm.weight"]))
                k["ln2_w"].append(f32(sd[p + "post_attention_layernorm.weight"]))
        self._t = k
        if self.arch == "gpt2":
            self.lnf_w, self.lnf_b = f32(sd["ln_f.weight"]), f32(sd["ln_f.bias"])
            self.wpe = f32(sd["wpe.weight"])
            emb = sd["wte.weight"]
            self.rope = None
        else:
            self.lnf_w, self.lnf_b = f32(sd["norm.weight"]), None
            self.wpe = None
            emb = sd["embed_tokens.weight"]
            self.rope = {}
        self.embed_f32 = emb.to(dev, F32).contiguous()            # [V_full, d] prompt-token gather table (the bf16-stream assembly kernel rounds what it gathers)
        self._arrays = {n: _ptr_array(v) for n, v in k.items() if v}
        self._structs = {}

    # ---- RoPE tables (HF:models/llama/modeling_llama.py:113-127), fp32, cached per T
    def _rope(self, T):
        if T not in self.rope:
            hd, theta = self.cfg["head_dim"], self.cfg["rope_theta"]
            inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
            freqs = torch.arange(T).float().unsqueeze(1) * inv_freq.unsqueeze(0)
            emb = torch.cat([freqs, freqs], dim=-1)
            self.rope[T] = (emb.cos().to(self.device).contiguous(), emb.sin().to(self.device).contiguous())
        return self.rope[T]

    def _struct(self, T):
        if T in self._structs:
            return self._structs[T]
        c = self.cfg
        w = N.BackboneWeights()
        w.arch = N.ARCH_GPT2 if self.arch == "gpt2" else N.ARCH_LLAMA
        w.n_layers, w.d, w.n_heads, w.n_kv_heads = c["n_layers"], c["d"], c["n_heads"], c["n_kv_heads"]
        w.head_dim, w.ffn, w.eps = c["head_dim"], c["ffn"], c["eps"]
        for n in ("w_qkv", "w_qkv_t", "b_qkv", "w_o", "w_o_t", "b_o", "w_fc", "w_fc_t", "b_fc", "w_proj", "w_proj_t", "b_proj",
                  "ln1_w", "ln1_b", "ln2_w", "ln2_b"):
            if n in self._arrays:
                setattr(w, n, C.cast(self._arrays[n], N.PP))
        w.lnf_w = self.lnf_w.data_ptr()
        w.lnf_b = self.lnf_b.data_ptr() if self.lnf_b is not None else None
        w.stream_dtype = N.MTL_BF16 if self.stream_dtype == BF16 else N.MTL_F32
        if self.arch == "llama":
            cos, sin = self._rope(T)
            w.rope_cos, w.rope_sin = cos.data_ptr(), sin.data_ptr()
        self._structs[T] = w
        return w

    @staticmethod
    def _drop_struct(drop):
        """drop = (attn_p, resid_p, seed) or None -> ctypes pointer (NULL when off)"""
        if not drop or (drop[0] <= 0 and drop[1] <= 0):
            return None
        return C.byref(N.BackboneDropout(float(drop[0]), float(drop[1]), int(drop[2]) & 0xFFFFFFFF))

    def prefix_cache(self, h0_prefix, key, T):
        """Prompt-row forward cache (SURVEY.md 7 "legal shortcut i"): per-layer keys / values of a CONSTANT prompt, bf16
        [L, n_prefix, 2 Hkv hd], built by one forward of the single prompt sequence h0_prefix f32 [1, n_prefix, d] and kept until `key`
        (the caller's identity of the prompt: its token ids) changes. The frozen weights never change (a new FrozenBackbone is built when
        they do). Returns (cache, n_prefix) for run_forward / BackboneFn."""
        n_prefix = h0_prefix.shape[1]
        ent = getattr(self, "_prefix", None)
        if ent is not None and ent[0] == key and ent[1].device == h0_prefix.device:
            return ent[1], n_prefix
        w = self._struct(T)                      # (RoPE table of the full sequence: rows [0, n_prefix) are the prompt's positions)
        lib = N.lib()
        dev = h0_prefix.device
        cache = torch.empty(lib.mtl_backbone_prefix_bytes(C.byref(w), n_prefix), dtype=torch.uint8, device=dev)
        saved = torch.empty(lib.mtl_backbone_saved_bytes(C.byref(w), 1, n_prefix), dtype=torch.uint8, device=dev)
        work = torch.empty(lib.mtl_backbone_work_bytes(C.byref(w), 1, n_prefix), dtype=torch.uint8, device=dev)
        h = h0_prefix.detach().contiguous()
        N.check(lib.mtl_backbone_prefix_build(C.byref(w), N.ptr(h), N.ptr(cache), N.ptr(saved), N.ptr(work), n_prefix, N.stream()),
                "mtl_backbone_prefix_build")
        self._prefix = (key, cache)
        self.prefix_builds = getattr(self, "prefix_builds", 0) + 1
        return cache, n_prefix

    def run_forward(self, h0, n_last, keep=True, drop=None, n_save=None, prefix=None):
        """h0 [B,T,d] of the stream dtype (wpe already added for GPT-2) -> (out bf16 [B,n_last,d], saved buffer).
        prefix = (cache, n_prefix) from prefix_cache(): forward on the last T - n_prefix rows of every sample only.
        drop = (attn_p, resid_p, seed): GPT-2's train-mode dropouts inside the stack (the backward needs the same tuple).
        n_save: trailing tokens per sample whose backward-only state (MLP pre-activations) is stored — the n_grad the backward
        will use; default all T when the buffer is kept, 0 otherwise."""
        B, T, d = h0.shape
        if h0.dtype != self.stream_dtype:
            raise ValueError(f"backbone input is {h0.dtype}, the stack was prepared for a {self.stream_dtype} residual stream")
        n_save = (T if keep else 0) if n_save is None else min(max(int(n_save), 0), T)
        if self.arch == "gpt2" and T > self.cfg["n_positions"]:
            raise ValueError(f"sequence length {T} exceeds GPT-2's {self.cfg['n_positions']} learned positions")
        w = self._struct(T)
        lib = N.lib()
        saved = torch.empty(lib.mtl_backbone_saved_bytes(C.byref(w), B, T), dtype=torch.uint8, device=h0.device)
        work = torch.empty(lib.mtl_backbone_work_bytes(C.byref(w), B, T), dtype=torch.uint8, device=h0.device)
        out = torch.empty((B, n_last, d), dtype=BF16, device=h0.device)
        if drop and self.arch != "gpt2" and (drop[0] > 0 or drop[1] > 0):
            raise ValueError("dropout inside the frozen stack exists for GPT-2 only (Llama has none)")
        dstruct = self._drop_struct(drop)
        if prefix is not None and (dstruct is not None or prefix[1] > T - n_last):
            prefix = None                         # dropout makes the prompt rows step-dependent: full forward
        pk, n_prefix = (N.ptr(prefix[0]), int(prefix[1])) if prefix is not None else (None, 0)
        N.check(lib.mtl_backbone_fwd(C.byref(w), N.ptr(h0), N.ptr(out), N.ptr(saved), N.ptr(work), B, T, n_last, n_save,
                                     dstruct, pk, n_prefix, N.stream()), "mtl_backbone_fwd")
        n_save = min(n_save, T - n_prefix)
        saved._n_save = n_save
        self.last_n_prefix = n_prefix
        tap = getattr(self, "tap_hidden", None)
        if tap is not None:      # parity tests: the fp32 residual stream after the requested layers, rows of the last n_last tokens of every sample
            for layer in list(tap):
                off = lib.mtl_backbone_saved_hidden_offset(C.byref(w), B, T, int(layer))
                tap[layer] = saved[off:off + B * T * d * h0.element_size()].view(h0.dtype).view(B, T, d)[:, T - n_last:, :].clone()
        return out, (saved if keep else None)

    def run_backward(self, h0, dout, saved, n_last, n_grad=None, drop=None):
        """n_grad: trailing tokens per sample that need a gradient (default all T); see mtl_backbone_bwd."""
        B, T, d = h0.shape
        n_grad = T if n_grad is None else max(int(n_grad), n_last)
        if saved is None:
            raise RuntimeError("backbone backward without saved activations (forward ran under no_grad)")
        if n_grad > getattr(saved, "_n_save", T):
            raise RuntimeError(f"backbone backward over {n_grad} tokens per sample, but the forward stored its state for {saved._n_save}")
        w = self._struct(T)
        lib = N.lib()
        work = torch.empty(lib.mtl_backbone_work_bytes(C.byref(w), B, T), dtype=torch.uint8, device=h0.device)
        dh0 = torch.empty_like(h0)
        N.check(lib.mtl_backbone_bwd(C.byref(w), N.ptr(h0), N.ptr(dout), N.ptr(dh0), N.ptr(saved), N.ptr(work), B, T, n_last,
                                     n_grad, self._drop_struct(drop), N.stream()), "mtl_backbone_bwd")
        return dh0

    def state_tensors(self):
        return self._t


def random_state_dict(cfg, seed=0, std=0.02, device="cpu", dtype=torch.float32):
    """Seeded random-init weights of the given architecture (bench / tests: no checkpoints are downloadable).
    `device`/`dtype` let multi-billion-parameter backbones be generated directly in HBM in bf16."""
    c = normalise_config(cfg)
    g = torch.Generator(device=device).manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    sd = {}
    d, L, ffn = c["d"], c["n_layers"], c["ffn"]
    if c["arch"] == "gpt2":
        sd["wte.weight"] = rn(c["vocab"], d)
        sd["wpe.weight"] = rn(c["n_positions"], d, s=0.01)
        for i in range(L):
            p = f"h.{i}."
            sd[p + "ln_1.weight"] = 1 + rn(d, s=0.05); sd[p + "ln_1.bias"] = rn(d)
            sd[p + "attn.c_attn.weight"] = rn(d, 3 * d); sd[p + "attn.c_attn.bias"] = rn(3 * d)
            sd[p + "attn.c_proj.weight"] = rn(d, d); sd[p + "attn.c_proj.bias"] = rn(d)
            sd[p + "ln_2.weight"] = 1 + rn(d, s=0.05); sd[p + "ln_2.bias"] = rn(d)
            sd[p + "mlp.c_fc.weight"] = rn(d, ffn); sd[p + "mlp.c_fc.bias"] = rn(ffn)
            sd[p + "mlp.c_proj.weight"] = rn(ffn, d); sd[p + "mlp.c_proj.bias"] = rn(d)
        sd["ln_f.weight"] = 1 + rn(d, s=0.05); sd["ln_f.bias"] = rn(d)
    else:
        H, Hkv, hd = c["n_heads"], c["n_kv_heads"], c["head_dim"]
        sd["embed_tokens.weight"] = rn(c["vocab"], d)
        for i in range(L):
            p = f"layers.{i}."
            sd[p + "input_layernorm.weight"] = 1 + rn(d, s=0.05)
            sd[p + "self_attn.q_proj.weight"] = rn(H * hd, d); sd[p + "self_attn.k_proj.weight"] = rn(Hkv * hd, d)
            sd[p + "self_attn.v_proj.weight"] = rn(Hkv * hd, d); sd[p + "self_attn.o_proj.weight"] = rn(d, H * hd)
            sd[p + "post_attention_layernorm.weight"] = 1 + rn(d, s=0.05)
            sd[p + "mlp.gate_proj.weight"] = rn(ffn, d); sd[p + "mlp.up_proj.weight"] = rn(ffn, d)
            sd[p + "mlp.down_proj.weight"] = rn(d, ffn)
        sd["norm.weight"] = 1 + rn(d, s=0.05)
    return sd
