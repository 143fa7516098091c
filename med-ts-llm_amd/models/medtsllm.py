"""Drop-in `MedTsLLM` for the reference's plugin surface, running on the gfx950 HIP path.

Mirrors R:models/medtsllm.py: same constructor `(config, dataset)`, same derived sizes, same trainable parameter
names / shapes (checkpoints interchange), same `forward(dict) -> Tensor`, `state_dict()` filtering and
`load_pretrained`. The arithmetic does not go through ATen/HF: every device op is a kernel of
libmedtsllm_hip.so (see hip/ops.py); there is no CPU fallback — forward on a non-ROCm device raises.

Numerics follow the reference's default `setup.dtype = "mixed"`: fp32 master weights and fp32 residual stream,
bf16 GEMM/attention operands with fp32 accumulation, fp32 norm/softmax statistics.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import ops
from ..hip.ops import (PatchTokenizeFn, LinearFn, LinearPairFn, ChannelMixFn, MappingFn, MappingTrainableFn, CrossAttnFn, AssembleFn, BackboneFn, EmbdDropoutFn, RevinDenormFn, pad64, pad_vocab,
                       mapping_split_k)
from . import prompt as P
from .backbone import FrozenBackbone, load_hf_dir, normalise_config

BF16 = torch.bfloat16

FORECAST_LIKE = ("forecasting", "reconstruction", "anomaly_detection", "pretraining")


class _TokenEmbedding(nn.Module):
    """Parameter holder named like R:models/layers/embed.py:29-42 (Conv1d weight [d_model, patch_len, 3], kaiming init)."""

    def __init__(self, c_in, d_model):
        super().__init__()
        self.tokenConv = nn.Conv1d(c_in, d_model, kernel_size=3, padding=1, padding_mode="circular", bias=False)
        nn.init.kaiming_normal_(self.tokenConv.weight, mode="fan_in", nonlinearity="leaky_relu")


class _PatchEmbedding(nn.Module):
    def __init__(self, d_model, patch_len, stride, dropout):
        super().__init__()
        self.patch_len, self.stride, self.p = patch_len, stride, dropout
        self.value_embedding = _TokenEmbedding(patch_len, d_model)


class _ReprogrammingLayer(nn.Module):
    """Parameter holder named like R:models/medtsllm.py:555-564."""

    def __init__(self, d_model, n_heads, d_keys, d_llm):
        super().__init__()
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_llm, d_keys * n_heads)
        self.value_projection = nn.Linear(d_llm, d_keys * n_heads)
        self.out_projection = nn.Linear(d_keys * n_heads, d_llm)
        self.n_heads = n_heads


class _FlattenHead(nn.Module):
    def __init__(self, nf, target_window):
        super().__init__()
        self.linear = nn.Linear(nf, target_window)


class MedTsLLM(nn.Module):

    supported_tasks = ["forecasting", "reconstruction", "anomaly_detection", "semantic_segmentation", "segmentation", "pretraining"]
    supported_modes = ["univariate", "multivariate"]

    def __init__(self, config, dataset, backbone_state=None):
        """`backbone_state=(hf_config_dict, state_dict)` bypasses the on-disk HF directory (benchmarks, tests)."""
        super().__init__()
        self.config = config
        self.model_config = config.models.medtsllm if "medtsllm" in config.models else config.models.timellm
        mc = self.model_config
        self.device = None
        self.pred_len, self.seq_len = config.pred_len, config.history_len
        self.task = config.task
        self.task_description = P.task_description(self.task, self.pred_len, self.seq_len, dataset)
        self.dataset_description = dataset.description
        self.d_ff, self.d_model = mc.d_ff, mc.d_model
        self.n_attention_heads, self.num_tokens = mc.n_heads, mc.num_tokens
        self.dropout = config.training.dropout
        self.n_lags = P.N_LAGS
        self.patch_len, self.stride = mc.patching.patch_len, mc.patching.stride
        self.n_patches = int((self.seq_len - self.patch_len) / self.stride + 2)     # R:models/medtsllm.py:52
        self.d_patch = self.d_model
        self.covariate_mode = mc.covariate_mode
        self.n_features = dataset.n_features
        self.n_classes = dataset.n_classes if self.task in ["classification", "semantic_segmentation"] else 0

        if self.task in FORECAST_LIKE:
            self.n_outputs_per_step = self.n_features
        elif self.task == "semantic_segmentation":
            self.n_outputs_per_step = self.n_classes if self.n_classes > 2 else 1
        elif self.task == "segmentation":
            self.n_outputs_per_step = 1
            assert config.tasks.segmentation.mode in ["boundary-prediction", "steps-to-boundary"]
        else:
            raise ValueError(f"Task {self.task} is not supported.")
        self.n_outputs = self.n_outputs_per_step * self.pred_len

        cm = self.covariate_mode
        if cm == "univariate":
            assert self.n_features == 1
        elif cm == "interleave":
            self.n_patches *= self.n_features
        elif cm == "concat":
            self.d_model *= self.n_features
        elif cm == "merge-end":
            self.feature_weighting = nn.Linear(self.n_features * self.n_outputs_per_step, self.n_outputs_per_step)
        elif cm == "weighted-average":
            self.feature_weighting = nn.Linear(self.n_features, 1)
        elif cm not in ("independent", "add"):
            raise ValueError(f"Unknown covariate mode {cm}")

        self._setup_llm(backbone_state)

        self.mapping_layer = nn.Linear(self.vocab_size, self.num_tokens)
        self._map_shard = None
        self.llm_dropout = True      # GPT-2 train-mode dropouts (set False to freeze them off, e.g. for benchmarking parity)
        self.patch_embedding = _PatchEmbedding(self.d_patch, self.patch_len, self.stride, self.dropout)
        self.reprogramming_layer = _ReprogrammingLayer(self.d_model, self.n_attention_heads, self.d_ff, self.d_llm)
        self.output_projection = _FlattenHead(self.d_ff * self.n_patches, self.n_outputs)
        self.embedding_downsample_mode = mc.embedding_downsample_mode
        if self.embedding_downsample_mode == "linear":
            self.embedding_downsample_layer = nn.Linear(self.d_llm, self.d_ff)
        elif self.embedding_downsample_mode == "average":
            assert self.d_llm % self.d_ff == 0
        elif self.embedding_downsample_mode != "truncate":
            raise ValueError(f"Unknown embedding downsample mode {self.embedding_downsample_mode}")
        self.lora_enabled = False
        self._id_cache = {}
        self.prune_dead_prompt_grads = True   # exact: skips gradients nobody consumes (set False for the full dh0)
        self.prompt_row_cache = True          # constant prompt + deterministic stack: per-layer prompt K/V cached, forward on the patch rows only
        self.fixed_prompt_ids = None   # int32 [1 or B, n_tok]: synthetic-benchmark prompt (no tokenizer files needed)
        self.debug_tap = None          # dict -> stage tensors (and, after backward, their gradients as "grad:<name>") for parity tests

    def _tap(self, name, t):
        """parity-test probe: no-op unless `debug_tap` is a dict. The i-th tensor tapped under `name` in a forward (encode_ts runs
        twice with "examples" prompting) is stored as "name@i", its gradient after backward as "grad:name@i"; "name" is the latest."""
        tap = self.debug_tap
        if tap is not None:
            key = f"{name}@{sum(1 for k in tap if k.startswith(name + '@'))}"
            tap[key] = tap[name] = t.detach()
            if t.requires_grad:
                t.register_hook(lambda g, k=key: tap.__setitem__("grad:" + k, g.detach()))
        return t

    # ------------------------------------------------------------------ construction (a11)
    def _setup_llm(self, backbone_state):
        """R:models/medtsllm.py:129-233 without HF model classes: config + safetensors -> FrozenBackbone (lazily on device)."""
        llm = self.model_config.llm
        if not llm.enabled:
            raise NotImplementedError("llm.enabled = false (llm_replacement MLP) is outside the HIP hot path")
        if llm.get("load_in_4bit", False) or llm.get("load_in_8bit", False) or ("lora" in self.model_config and self.model_config.lora.enabled):
            raise NotImplementedError("LoRA / 4-8 bit quantised backbones are outside the HIP hot path (SURVEY.md §2 #16)")
        self.llm_enabled, self.llm_id, self.llm_layers = True, llm.llm, llm.llm_layers
        if backbone_state is None:
            hf_cfg, sd = load_hf_dir(self.llm_id)
        else:
            hf_cfg, sd = backbone_state
        self._hf_cfg, self._hf_state = hf_cfg, sd
        bc = normalise_config(hf_cfg)
        self.d_llm = bc["d"]
        self._head_dim = bc["head_dim"]
        emb = sd["wte.weight"] if bc["arch"] == "gpt2" else sd["embed_tokens.weight"]
        if emb.shape[0] > 100_000:
            # R:models/medtsllm.py:220-222: 100 000 linspace-sampled rows become a fresh, TRAINABLE nn.Parameter (Llama-3)
            inds = torch.linspace(0, emb.shape[0] - 1, 100_000, dtype=torch.long)
            self.word_embeddings = nn.Parameter(emb.detach()[inds.to(emb.device), :].float().clone())
        else:
            # registered like the reference (alias of the frozen input-embedding table; filtered from state_dict)
            self.word_embeddings = nn.Parameter(emb.detach().float().clone(), requires_grad=False)
        self.vocab_size = self.word_embeddings.shape[0]
        self.backbone = None
        self.tokenizer = None
        self._tok_dir = self.llm_id if (isinstance(self.llm_id, str) and os.path.isdir(self.llm_id)) else None

    def _get_tokenizer(self):
        if self.tokenizer is None:
            if self._tok_dir is None:
                raise RuntimeError("text prompts need a tokenizer directory (config.models.*.llm.llm)")
            from transformers import AutoTokenizer
            tok = AutoTokenizer.from_pretrained(self._tok_dir)
            if tok.eos_token:                      # R:models/medtsllm.py:212-217
                tok.pad_token = tok.eos_token
            else:
                tok.add_special_tokens({"pad_token": "[PAD]"})
                tok.pad_token = "[PAD]"
            self.tokenizer = tok
        return self.tokenizer

    def _ensure_backbone(self, device):
        if device.type != "cuda":
            raise RuntimeError("MedTsLLM (HIP path) needs a ROCm GPU tensor; there is no CPU fallback")
        if self.d_ff not in (32, 64, 128) or self._head_dim not in (32, 64, 128):
            raise ValueError(f"HIP attention kernels support head dims 32/64/128 (d_ff={self.d_ff}, backbone head_dim={self._head_dim})")
        if (self.n_attention_heads * self.d_ff) % 64 or self.d_llm % 64:
            raise ValueError("HIP GEMMs need n_heads*d_ff and d_llm to be multiples of 64")
        # setup.dtype = "bf16" (R:tasks/base.py:261-262: the whole model in bf16): bf16 trainable parameters -> a bf16 residual stream through the
        # frozen stack too (FrozenBackbone(stream_dtype)); fp32 parameters ("mixed" / "fp32") -> the fp32 stream
        sdt = BF16 if self.mapping_layer.weight.dtype == BF16 else torch.float32
        if self.backbone is None or self.backbone.device != device or self.backbone.stream_dtype != sdt:
            self.backbone = FrozenBackbone(self._hf_cfg, self._hf_state, device, n_layers=self.llm_layers, stream_dtype=sdt)
            bb = self.backbone
            V, d = self.vocab_size, self.d_llm
            Vp = pad_vocab(V + 1) if not self.word_embeddings.requires_grad else pad64(V + 1)
            if not self.word_embeddings.requires_grad:       # frozen table: bf16 operands prepared once
                wT = torch.zeros((d, Vp), dtype=BF16, device=device)
                wT[:, :V] = bb.embed_f32.t().to(BF16)
                wT[:, V] = 1.0                               # bias carrier (see MappingFn)
                self._wT, self._w = wT, bb.embed_f32.to(BF16).contiguous()
            nkt = Vp // 64
            tiles = ((self.num_tokens + 127) // 128) * ((d + 127) // 128)
            self._map_split_k = max(1, min(nkt, 16, (512 + tiles - 1) // tiles))
        return self.backbone

    def state_dict(self, *args, **kwargs):
        """R:models/medtsllm.py:235-246 — only trainable front/back-end weights are checkpointed.
        With a row-sharded mapping layer (shard_mapping_layer) this is a COLLECTIVE: every rank must call it; the
        returned mapping_layer.* tensors are the gathered full [num_tokens, ...] ones, as the reference stores them."""
        sd = super().state_dict(*args, **kwargs)
        for k in [k for k in sd.keys() if k[:4] == "llm."]:
            del sd[k]
        if "word_embeddings" in sd:
            del sd["word_embeddings"]
        if self._map_shard is not None:
            from .. import parallel
            _, world, _, _, group = self._map_shard
            for k in [k for k in sd if k.endswith("mapping_layer.weight") or k.endswith("mapping_layer.bias")]:
                sd[k] = parallel.gather_rows(sd[k], world, group)
        su = getattr(self, "_opt_shards", None)
        if su is not None and su.world > 1:
            # row-sharded optimiser step that publishes bf16 rows only (parallel.ShardedUpdate): a rank's fp32 master is current for the
            # rows it owns; the checkpoint gathers every owner's rows (a collective, like the mapping rows above)
            from .. import parallel
            for it in su.items:
                for k in [k for k in sd if k == it["name"] or k.endswith("." + it["name"])]:
                    if it["shadow"] is not None:
                        sd[k] = parallel.gather_rows(sd[k][it["r0"]:it["r1"]].contiguous(), su.world, su.group)
        return sd

    def load_state_dict(self, state_dict, *args, **kwargs):
        if self._map_shard is not None:
            _, _, r0, r1, _ = self._map_shard
            state_dict = dict(state_dict)
            for k in [k for k in state_dict if k.endswith("mapping_layer.weight") or k.endswith("mapping_layer.bias")]:
                if state_dict[k].shape[0] == self.num_tokens:
                    state_dict[k] = state_dict[k][r0:r1]
        return super().load_state_dict(state_dict, *args, **kwargs)

    def shard_mapping_layer(self, rank, world, group=None, force=False):
        """Data parallelism: row-shard mapping_layer.{weight [num_tokens, V], bias} over the ranks.
        Rank r keeps rows [r*S/N, (r+1)*S/N) as its own leaf parameters, computes only those prototype rows of
        `source` and all-gathers them (parallel.AllGatherRows). Removes the replicated S x V x d GEMMs (fwd + dW), the
        Adam traffic and 70 % of the gradient all-reduce payload from every rank. Call after .to(device), before the
        optimiser is built. Returns False (layer stays replicated) when it does not apply. force: also in a ONE-rank group (the collectives
        then run against the real backend with a single participant: how the RCCL calls are exercised on a 1-GPU box)."""
        if (world <= 1 and not force) or self.word_embeddings.requires_grad or self.num_tokens % world != 0:
            return False
        from .. import parallel
        r0, r1 = parallel.shard_range(self.num_tokens, rank, world)
        ml = self.mapping_layer
        w, b = nn.Parameter(ml.weight.data[r0:r1].clone()), nn.Parameter(ml.bias.data[r0:r1].clone())
        w._dp_sharded = b._dp_sharded = True
        ml.weight, ml.bias, ml.out_features = w, b, r1 - r0
        self._map_shard = (rank, world, r0, r1, group)
        self._map_shadow = None
        return True

    def load_pretrained(self, saved_state):
        """R:models/medtsllm.py:515-527."""
        for k in ("word_embeddings", "output_projection.linear.bias", "output_projection.linear.weight"):
            if k in saved_state:
                del saved_state[k]
        incompat = self.load_state_dict(saved_state, strict=False)
        assert len(incompat.unexpected_keys) == 0, f"Unexpected keys in model state: {incompat.unexpected_keys}"
        return list(saved_state.keys())

    # ------------------------------------------------------------------ prompt (a6, host side)
    def build_prompt(self, inputs):
        cfg = self.model_config.get("prompting")
        if cfg is None:
            cfg = P.DEFAULT_PROMPTING
        on = any((cfg.get(k, False) if hasattr(cfg, "get") else cfg[k]) for k in ("dataset", "clip", "input_stats", "task", "examples"))
        bos = self._get_tokenizer().bos_token if on else None
        return P.build_prompt_parts(inputs, cfg, self.dataset_description, self.task_description, bos)

    def _prompt_ids(self, inputs, device):
        """-> (ids, splice): ids int32 [B or 1, n_tok] left-padded token ids (None when prompting is off); constant
        parts are tokenised once. splice = None, or for "examples" prompting (R:models/medtsllm.py:313-319,403,
        R:datasets/ecg.py:139-166) a dict(emb [B, P_ex, d_llm], pos [B]) describing where each sample's example
        embeddings replace the placeholder (pad) tokens that reserve their rows in `ids`."""
        if self.fixed_prompt_ids is not None:
            if not getattr(self, "_fixed_ids_checked", None) is self.fixed_prompt_ids:
                self._check_ids(self.fixed_prompt_ids.tolist())
                self._fixed_ids_checked = self.fixed_prompt_ids
            return self.fixed_prompt_ids.to(device=device, dtype=torch.int32), None
        prompts = self.build_prompt(inputs)
        if len(prompts[0]) == 0:
            return None, None
        tok = self._get_tokenizer()
        # each part is tokenised separately (R:models/medtsllm.py:300-301); the per-sample statistics strings are new on
        # every step, so all uncached parts of the batch go through ONE batched tokenizer call
        if len(self._id_cache) > 8192:
            self._id_cache.clear()
        new = sorted({p for parts in prompts for p in parts if isinstance(p, str) and p not in self._id_cache})
        if new:
            for p, ids in zip(new, tok(new, padding=False, truncation=False).input_ids):
                self._id_cache[p] = ids
        tensors = [[p for p in parts if not isinstance(p, str)] for parts in prompts]
        splice = None
        if any(tensors):
            # tensor parts go through encode_ts like the main input (R: encode_part); one per sample, equal shapes
            if any(len(t) != 1 for t in tensors) or self.covariate_mode in ("independent", "merge-end"):
                raise NotImplementedError("'examples' prompting: exactly one tensor part per sample, and a covariate mode "
                                          "whose encode_ts keeps the batch size (the reference's own concat fails otherwise)")
            ex = torch.cat([t[0].to(device) for t in tensors], dim=0)
            emb = self.encode_ts(ex)[0]                                   # [B, P_ex, d_llm] bf16, differentiable
            pad = [tok.pad_token_id] * emb.shape[1]
            id_lists = [[self._id_cache[p] if isinstance(p, str) else pad for p in parts] for parts in prompts]
            rows = P.left_pad_ids(id_lists, tok.pad_token_id)
            pos = []
            for parts, ids, row in zip(prompts, id_lists, rows):
                k = next(i for i, p in enumerate(parts) if not isinstance(p, str))
                pos.append(len(row) - sum(len(x) for x in ids) + sum(len(x) for x in ids[:k]))
            splice = {"emb": emb, "pos": torch.tensor(pos, dtype=torch.long, device=device), "first": min(pos)}
        else:
            id_lists = [[self._id_cache[p] for p in parts] for parts in prompts]
            rows = P.left_pad_ids(id_lists, tok.pad_token_id)
            if all(r == rows[0] for r in rows):
                rows = rows[:1]                       # one shared prompt: the kernel broadcasts it
        self._check_ids(rows)
        self._last_prompt_rows = ("ids", tuple(rows[0])) if len(rows) == 1 else None
        return torch.tensor(rows, dtype=torch.int32, device=device), splice

    def late_parameters(self):
        """parameters a forward reads only AFTER the frozen backbone: an optimiser may update them under the next step's front end and backbone
        (HipAdam.defer) as long as `optimizer_wait` runs before they are read — predict() calls it in front of the down-sample GEMM"""
        ps = list(self.output_projection.parameters())
        if isinstance(getattr(self, "embedding_downsample_layer", None), nn.Linear):
            ps += list(self.embedding_downsample_layer.parameters())
        if self.covariate_mode == "merge-end":
            ps += list(self.feature_weighting.parameters())
        return ps

    optimizer_wait = None      # callable set by whoever defers updates of late_parameters() (BaseTask.build_optimizer, bench.py)

    def _await_rows(self, *params):
        """DP with a row-sharded optimiser step (parallel.ShardedUpdate.publish(async_op=True)): the other ranks' updated rows of these
        tensors must have arrived before a kernel reads them"""
        su = getattr(self, "_opt_shards", None)
        if su is not None:
            for p in params:
                su.wait_published(p)

    def _await_unlisted_rows(self):
        """ShardedUpdate picks tensors by SIZE (setup.shard_optimizer_min_numel is the user's), the model awaits three of them right in front of
        their first reader (mapping weight, word embeddings, flatten head: the ones that are big at the default threshold). Any OTHER tensor
        that was sharded — projections, down-sample layer at a small threshold — is awaited here, before the forward reads anything."""
        su = getattr(self, "_opt_shards", None)
        if su is not None:
            su.wait_published(skip=(self.mapping_layer.weight, self.word_embeddings, self.output_projection.linear.weight))

    def _prompt_key(self, ids):
        """host-side identity of a shared prompt (no device sync: the ids came from host lists / a host tensor checked once)"""
        if self.fixed_prompt_ids is not None:
            return ("fixed", id(self.fixed_prompt_ids), tuple(self.fixed_prompt_ids.shape))
        return self._last_prompt_rows

    def _check_ids(self, rows):
        """prompt token ids index the frozen embedding table inside a kernel that cannot raise: range-check them on the host (the
        reference's nn.Embedding raises an IndexError here, e.g. for a '[PAD]' token added beyond an un-resized table)"""
        V = self._hf_state["wte.weight" if "wte.weight" in self._hf_state else "embed_tokens.weight"].shape[0]
        lo = min(min(r) for r in rows)
        hi = max(max(r) for r in rows)
        if lo < 0 or hi >= V:
            raise IndexError(f"prompt token id out of range: ids span [{lo}, {hi}], the embedding table has {V} rows")

    # ------------------------------------------------------------------ forward
    def forward(self, inputs):
        pred = self.predict(inputs)
        if inputs["x_enc"].dtype == BF16 and pred.dtype != BF16:
            pred = pred.to(BF16)        # setup.dtype = "bf16": bf16 inputs and parameters give a bf16 prediction (R:tasks/base.py:205-208,261-262)
        if not self.training:   # R:models/medtsllm.py:251-259
            if self.task == "semantic_segmentation":
                pred = F.softmax(pred, dim=-1) if self.n_classes > 2 else torch.sigmoid(pred)
            elif self.task == "segmentation" and self.config.tasks.segmentation.mode == "boundary-prediction":
                pred = torch.sigmoid(pred)
        return pred

    def _mapping_shadow(self):
        """Persistent bf16 copy [S, Vp] of mapping_layer.weight (the autocast weight cast). Re-cast only when the fp32
        master changed behind its back; hip.optim.HipAdam writes it while updating the master (bf16_shadows())."""
        W = self.mapping_layer.weight
        sh = getattr(self, "_map_shadow", None)
        if sh is None or sh.param is not W or sh.tensor.device != W.device:
            from ..hip.optim import Bf16Shadow
            sh = Bf16Shadow(W, torch.zeros((W.shape[0], pad_vocab(self.vocab_size + 1)), dtype=torch.bfloat16, device=W.device))
            self._map_shadow = sh
        return sh

    def _vocab_shadows(self):
        """(mapping shadow [S, pad64(V + 1)], word-embedding shadow [V, d]) for the TRAINABLE-vocabulary path (MappingTrainableFn): persistent
        bf16 copies that HipAdam keeps current — what the forward reads and, under a row-sharded optimiser step, what the ranks exchange."""
        W, E = self.mapping_layer.weight, self.word_embeddings
        ent = self.__dict__.get("_vocab_sh")
        if ent is None or ent[0].param is not W or ent[1].param is not E or ent[0].tensor.device != W.device:
            from ..hip.optim import Bf16Shadow
            ent = (Bf16Shadow(W, torch.zeros((W.shape[0], pad64(W.shape[1] + 1)), dtype=torch.bfloat16, device=W.device)),
                   Bf16Shadow(E, torch.zeros(tuple(E.shape), dtype=torch.bfloat16, device=E.device)))
            self.__dict__["_vocab_sh"] = ent
        return ent

    def _linear_shadow(self, lin):
        """Persistent bf16 copy [N, pad64(K)] (zero K padding) of a trainable Linear's weight: the operand of its forward GEMM and,
        read K-major, of its input-gradient GEMM. Same contract as _mapping_shadow."""
        W = lin.weight
        if not W.is_cuda:
            return None
        table = self.__dict__.setdefault("_lin_shadows", {})
        sh = table.get(id(lin))
        if sh is None or sh.param is not W or sh.tensor.device != W.device:
            from ..hip.optim import Bf16Shadow
            sh = Bf16Shadow(W, torch.zeros((W.shape[0], pad64(W.shape[1])), dtype=torch.bfloat16, device=W.device))
            table[id(lin)] = sh
        return sh

    def _kv_shadow_pair(self):
        """(key shadow, value shadow, base): the two projections' bf16 weight copies as the halves of ONE [2 H E, pad64(d_llm)] buffer, so
        that their backward can treat [Wk; Wv] as one weight (LinearPairFn)."""
        rl = self.reprogramming_layer
        Wk, Wv = rl.key_projection.weight, rl.value_projection.weight
        if not Wk.is_cuda or Wk.shape[1] != Wv.shape[1] or Wk.shape[0] % 8 or Wv.shape[0] % 8:
            return self._linear_shadow(rl.key_projection), self._linear_shadow(rl.value_projection), None
        table = self.__dict__.setdefault("_lin_shadows", {})
        ent = table.get("kv_pair")
        if ent is None or ent[0].param is not Wk or ent[1].param is not Wv or ent[2].device != Wk.device:
            from ..hip.optim import Bf16Shadow
            base = torch.zeros((Wk.shape[0] + Wv.shape[0], pad64(Wk.shape[1])), dtype=torch.bfloat16, device=Wk.device)
            ent = (Bf16Shadow(Wk, base[:Wk.shape[0]]), Bf16Shadow(Wv, base[Wk.shape[0]:]), base)
            table["kv_pair"] = ent
            table[id(rl.key_projection)], table[id(rl.value_projection)] = ent[0], ent[1]
        return ent

    def _shadowed_linears(self):
        rl = self.reprogramming_layer
        lins = [rl.query_projection, rl.key_projection, rl.value_projection, rl.out_projection, self.output_projection.linear]
        if isinstance(getattr(self, "embedding_downsample_layer", None), nn.Linear):
            lins.append(self.embedding_downsample_layer)
        return lins

    def bf16_shadows(self):
        """Shadows an optimiser may keep current (HipAdam.register_shadow)."""
        if not self.mapping_layer.weight.is_cuda:
            return []
        out = []
        if ops._LINEAR_XT:                     # (the fallback Linear path casts its weights per call and would never read them)
            self._kv_shadow_pair()             # (key / value shadows are halves of one buffer)
            out = [sh for sh in (self._linear_shadow(m) for m in self._shadowed_linears()) if sh is not None and sh.param.shape[0] % 8 == 0]
        if not self.word_embeddings.requires_grad:
            out.append(self._mapping_shadow())
        elif self.word_embeddings.is_cuda and self.word_embeddings.dim() == 2 and self.word_embeddings.numel() < (1 << 32):
            out.extend(self._vocab_shadows())
        return out

    def encode_ts(self, x_enc):
        """R:models/medtsllm.py:263-297 -> (x_tok bf16 [B', P', d_llm], mean [B,C], stdev [B,C])."""
        if x_enc.ndim == 2:
            x_enc = x_enc.unsqueeze(-1)
        bs, _, C = x_enc.shape
        assert C == self.n_features
        self._await_unlisted_rows()
        concat = self.covariate_mode == "concat"
        rl = self.reprogramming_layer
        # training.dropout: one host seed per encode, split over the two sites (patch embedding R:models/layers/embed.py:197, attention
        # probabilities R:models/medtsllm.py:588); drawn from the host torch RNG, so no device sync
        drop_on = self.training and self.dropout > 0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop_on else 0
        tokens, mean, stdev = PatchTokenizeFn.apply(x_enc, self.patch_embedding.value_embedding.tokenConv.weight,
                                                    self.patch_len, self.stride, concat, float(self.dropout) if drop_on else 0.0, seed ^ 0x2545F491)
        self._tap("tokens", tokens)
        if self.word_embeddings.requires_grad:
            self._await_rows(self.mapping_layer.weight, self.word_embeddings)
            sh_map, sh_emb = self._vocab_shadows() if self.word_embeddings.is_cuda else (None, None)
            source = MappingTrainableFn.apply(self.mapping_layer.weight, self.mapping_layer.bias, self.word_embeddings, self._map_split_k, sh_map, sh_emb)
        else:
            W = self.mapping_layer.weight
            split_k = mapping_split_k(W.shape[0], self.d_llm, self._wT.shape[1])
            source = MappingFn.apply(W, self.mapping_layer.bias, self._wT, self._w, split_k, self._mapping_shadow())
            if self._map_shard is not None:      # rows of the other ranks (DP row sharding)
                from ..parallel import AllGatherRows
                rank, world, _, _, group = self._map_shard
                source = AllGatherRows.apply(source, rank, world, group)
        self._tap("source", source)
        q = self._tap("q", LinearFn.apply(tokens, rl.query_projection.weight, rl.query_projection.bias, self._linear_shadow(rl.query_projection)))
        shk, shv, pair = self._kv_shadow_pair()
        if pair is not None and ops._LINEAR_XT:
            k, v = LinearPairFn.apply(source, rl.key_projection.weight, rl.key_projection.bias, rl.value_projection.weight, rl.value_projection.bias, shk, shv, pair)
            k, v = self._tap("k", k), self._tap("v", v)
        else:
            k = self._tap("k", LinearFn.apply(source, rl.key_projection.weight, rl.key_projection.bias, shk))
            v = self._tap("v", LinearFn.apply(source, rl.value_projection.weight, rl.value_projection.bias, shv))
        if drop_on:   # A = dropout(softmax(.)), R:models/medtsllm.py:588
            a = CrossAttnFn.apply(q, k, v, self.n_attention_heads, self.d_ff, float(self.dropout), seed)
        else:
            a = CrossAttnFn.apply(q, k, v, self.n_attention_heads, self.d_ff)
        enc = self._tap("reprog", LinearFn.apply(a, rl.out_projection.weight, rl.out_projection.bias, self._linear_shadow(rl.out_projection)))     # [B', P, d_llm]
        n_patches, d_llm = enc.shape[1], self.d_llm
        cm = self.covariate_mode
        if cm == "add":              # mean over the channels (R:models/medtsllm.py:286)
            enc = ChannelMixFn.apply(enc.reshape(bs, C, n_patches * d_llm, 1), None, None, BF16).view(bs, n_patches, d_llm)
        elif cm == "weighted-average":   # feature_weighting = Linear(C, 1) on the channel-last view (:288-291)
            if self.debug_tap is not None:     # (parity tests: the Linear's input as the reference sees it)
                self._tap("fw_in", enc.detach().reshape(bs, C, n_patches, d_llm).permute(0, 2, 3, 1).float())
            enc = self._tap("fw_out", ChannelMixFn.apply(enc.reshape(bs, C, n_patches * d_llm, 1), self.feature_weighting.weight,
                                                         self.feature_weighting.bias, BF16).view(bs, n_patches, d_llm, 1)).squeeze(-1)
        elif cm == "interleave":
            enc = enc.reshape(bs, C, -1, d_llm).permute(0, 2, 1, 3).reshape(bs, -1, d_llm)
        return enc, mean, stdev

    def predict(self, inputs):
        x_enc = inputs["x_enc"]
        bs, _, C = x_enc.size()
        if self.device is None:
            self.device = x_enc.device
        bb = self._ensure_backbone(x_enc.device)
        ids, splice = self._prompt_ids(inputs, x_enc.device)      # (example tensors are encoded first: RevIN state is then
        x_tok, mean, stdev = self.encode_ts(x_enc)                 #  overwritten by the main input, as in the reference)
        cm = self.covariate_mode
        if ids is not None and cm in ("independent", "merge-end") and ids.shape[0] != 1:
            ids = ids.repeat_interleave(C, dim=0)        # R:models/medtsllm.py:343-344
        # GPT-2's own dropouts are live whenever the module is in train mode (the reference calls model.train() on the whole
        # model, frozen LLM included): embd_pdrop on inputs_embeds + wpe, attn_pdrop / resid_pdrop inside the stack
        llm_drop = self.training and bb.arch == "gpt2" and self.llm_dropout
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if llm_drop else 0          # host RNG: no device sync
        embd_p = bb.cfg["embd_pdrop"] if llm_drop else 0.0
        fuse_embd = embd_p > 0 and splice is None           # (spliced examples are added before the dropout: separate pass there)
        h0 = AssembleFn.apply(x_tok, ids, bb.embed_f32, bb.wpe, embd_p if fuse_embd else 0.0, seed ^ 0x5bd1e995, bb.stream_dtype)
        # only the x_tok rows of h0 have a trainable ancestor: prompt-row gradients are dead
        n_grad = x_tok.shape[1] if self.prune_dead_prompt_grads else None
        if splice is not None:
            # the placeholder rows hold pad_embedding (+ wpe): adding (example - pad_embedding) puts the example there
            emb, pos = splice["emb"], splice["pos"]
            rows = pos[:, None] + torch.arange(emb.shape[1], device=pos.device)[None, :]
            bidx = torch.arange(emb.shape[0], device=pos.device)[:, None].expand_as(rows)
            delta = emb.float() - bb.embed_f32[self._get_tokenizer().pad_token_id]
            h0 = h0.index_put((bidx, rows), delta.to(h0.dtype), accumulate=True)
            if n_grad is not None:                       # gradients are alive from the first example row on
                n_grad = h0.shape[1] - splice["first"]
        drop = None
        if llm_drop:
            c = bb.cfg
            if embd_p > 0 and not fuse_embd:
                h0 = EmbdDropoutFn.apply(h0, embd_p, seed ^ 0x5bd1e995)
            if c["attn_pdrop"] > 0 or c["resid_pdrop"] > 0:
                drop = (c["attn_pdrop"], c["resid_pdrop"], seed)
        self._tap("h0", h0)
        # prompt-row forward cache: ONE prompt shared by every sample (ids [1, n_tok]: dataset / task text, R:configs/datasets/ludb.toml:49-52)
        # and a deterministic stack (Llama always; GPT-2 outside train mode) -> the prompt rows' keys / values per layer are the same in
        # every sample and every step. Keyed by the token ids; per-sample prompts (statistics, clip descriptions, examples) never cache.
        prefix = None
        if (self.prompt_row_cache and ids is not None and ids.shape[0] == 1 and splice is None and drop is None and embd_p == 0
                and ids.shape[1] <= h0.shape[1] - self.n_patches):
            key = self._prompt_key(ids)
            prefix = bb.prefix_cache(h0[:1, :ids.shape[1]], key, h0.shape[1])
        bb.tap_hidden = self.debug_tap.get("hidden_after") if self.debug_tap is not None else None      # {layer: None} -> filled by run_forward
        dec = self._tap("dec", BackboneFn.apply(h0, bb, self.n_patches, n_grad, drop, prefix))   # [B', n_patches, d_llm] (final norm on the consumed rows only)
        if self.optimizer_wait is not None:
            self.optimizer_wait()        # deferred updates of the tail's parameters (side stream) must have landed before the tail reads them
        mode = self.embedding_downsample_mode
        if mode == "truncate":
            dec = dec[:, :, :self.d_ff]
        elif mode == "linear":
            dec = self._tap("down", LinearFn.apply(dec, self.embedding_downsample_layer.weight, self.embedding_downsample_layer.bias, self._linear_shadow(self.embedding_downsample_layer)))
        else:
            # "average" down-sampling: mean over groups of d_llm / d_ff neighbouring features (R:models/medtsllm.py:360-362) = the channel-mix
            # kernel with the group as its inner axis and constant weights
            grp = dec.shape[-1] // self.d_ff
            wavg = torch.full((1, grp), 1.0 / grp, dtype=torch.float32, device=dec.device)
            dec = ChannelMixFn.apply(dec.reshape(dec.shape[0], 1, self.n_patches * self.d_ff, grp), wavg, None, BF16).view(dec.shape[0], self.n_patches, self.d_ff)
        head_in = dec.permute(0, 2, 1).reshape(dec.shape[0], -1)       # feature index = f * P + p (R:models/medtsllm.py:366,549)
        kp = pad64(head_in.shape[1])
        if kp != head_in.shape[1]:
            head_in = F.pad(head_in, (0, kp - head_in.shape[1]))
        self._await_rows(self.output_projection.linear.weight)
        out = self._tap("head", LinearFn.apply(head_in.contiguous(), self.output_projection.linear.weight, self.output_projection.linear.bias, self._linear_shadow(self.output_projection.linear)))
        if cm == "independent":      # mean of the per-channel predictions (R:models/medtsllm.py:371)
            out = ChannelMixFn.apply(out.reshape(bs, C, self.pred_len * self.n_outputs_per_step, 1), None, None, torch.float32)
            out = out.view(bs, self.pred_len, self.n_outputs_per_step)
        elif cm == "merge-end":      # feature_weighting = Linear(C * n_out, n_out) on the [bs, pred, n_out * C] view (:373-375)
            if self.debug_tap is not None:
                self._tap("fw_in", out.detach().float().view(bs, C, self.pred_len, self.n_outputs_per_step).permute(0, 2, 3, 1).reshape(bs, self.pred_len, -1))
            out = self._tap("fw_out", ChannelMixFn.apply(out.reshape(bs, C, self.pred_len, self.n_outputs_per_step), self.feature_weighting.weight,
                                                         self.feature_weighting.bias, torch.float32))
        else:
            out = out.view(bs, self.pred_len, self.n_outputs_per_step)
        if self.task in FORECAST_LIKE:
            out = RevinDenormFn.apply(out, mean, stdev)
        else:
            out = out.squeeze(-1).float()
        return out
