#!/usr/bin/env python3
"""bench.py — samples/s of the MedTsLLM fwd+bwd(+optimizer) hot path on N MI355X (one process per GPU).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE from env) — or, when started WITHOUT a launcher, it re-runs itself under one. Rank 0 prints
ONE JSON line.

Workload (BASELINE.json metric, SURVEY.md §8d "M"): synthetic [B=32, L=1024, C=12] windows per GPU, patch 16/8 -> P=128,
d_model=32, d_ff=128, 8 heads, 1024 prototype tokens, concat covariates, linear down-sample, forecasting pred_len=96, fixed
128-token prompt -> T=256, frozen GPT-2-small backbone (12 x 768, random init, bf16 operands, fp32 residual/statistics = the
reference's dtype="mixed"), training.dropout 0.1 and GPT-2's own train-mode dropouts live, as the reference trains. A step =
forward + MSE loss + backward + [DP all-reduce] + Adam step + zero_grad. Weak scaling: per-GPU batch fixed at 32.

Extra objects on the JSON line:
  roofline      the dominant MFMA kernel (bf16 GEMM instance with the largest total time): achieved TFLOP/s = algorithmic 2MNK
                FLOPs / kernel duration, measured in this process in a profiled replay of the same steps right after the timed
                region: every launch carries its own start/stop event pair (kernel begin -> end on its stream, the duration
                rocprofv3 --kernel-trace reports). `traffic`: HBM bytes per launch from PMC counters — on the headline line measured
                by THIS run (N = 1: two rocprofv3 --pmc child passes of the workload after everything else, `traffic_source` says so;
                --no-live-traffic skips them); on configs[] and as the fall-back, the committed PMC passes of the workload
                (profiles/pmc_traffic_<workload>.json), null when those were taken with other kernel sources.
  roofline_hbm  the same for the dominant HBM-bound kernel (norm family) against 8 TB/s
  roofline_attention  the same for the dominant attention kernel against the MFMA peak (full-rectangle FLOP convention)
  cpu_baseline  the oracle (a plain-torch port of the reference math, pinned to reference goldens) timed on the host cores,
                rank 0, N = 1 only, on a bounded sample of the same workload (+ BASELINE.json configs[0], the ETTh1-shaped case)
  configs       default run only: the same measurement for BASELINE.json configs[2] (Llama-2-7B backbone), 2 + 5 steps
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GPT2_SMALL = {"model_type": "gpt2", "vocab_size": 50257, "n_positions": 1024, "n_embd": 768, "n_layer": 12, "n_head": 12,
              "layer_norm_epsilon": 1e-5, "embd_pdrop": 0.1, "attn_pdrop": 0.1, "resid_pdrop": 0.1}    # the released gpt2 config
LLAMA2_7B = {"model_type": "llama", "vocab_size": 32000, "hidden_size": 4096, "intermediate_size": 11008, "num_hidden_layers": 32,
             "num_attention_heads": 32, "num_key_value_heads": 32, "rms_norm_eps": 1e-5, "rope_theta": 10000.0}
LLAMA3_8B = {"model_type": "llama", "vocab_size": 128256, "hidden_size": 4096, "intermediate_size": 14336, "num_hidden_layers": 32,
             "num_attention_heads": 32, "num_key_value_heads": 8, "rms_norm_eps": 1e-5, "rope_theta": 500000.0}
WORKLOADS = {
    # name: (hf cfg, B per GPU, L, C, pred_len, n_tok, task)
    "gpt2s_B32_L1024_C12": (GPT2_SMALL, 32, 1024, 12, 96, 128, "forecasting"),
    "gpt2s_etth1_B32_L512_C7": (GPT2_SMALL, 32, 512, 7, 96, 128, "forecasting"),
    # BASELINE.json configs[2]: LUDB-shaped semantic segmentation (4 classes), frozen Llama-2-7B (random init, generated on the GPU)
    "llama2_7b_semseg_B32_L1024_C12": (LLAMA2_7B, 32, 1024, 12, 1024, 128, "semantic_segmentation"),
    # BASELINE.json configs[4] shape: reconstruction, frozen Llama-3-8B (GQA 32/8, vocab 128256 -> 100 000 TRAINABLE sub-sampled rows)
    "llama3_8b_recon_B32_L1024_C12": (LLAMA3_8B, 32, 1024, 12, 1024, 128, "reconstruction"),
    # BASELINE.json configs[3] shape: PSM anomaly detection = reconstruction of 25-channel L=2048 windows (P=256, T=384; head 32768 -> 51200)
    "llama2_7b_psm_B32_L2048_C25": (LLAMA2_7B, 32, 2048, 25, 2048, 128, "anomaly_detection"),
    # covariate_mode = "interleave" (R:models/medtsllm.py:73-74,292-295) — every channel's patches are their own LLM
    # tokens: T = 128 + 128 * 12 = 1664 per sample. K/V no longer fit the LDS: the chunked (flash) causal attention carries the stack.
    "llama2_7b_semseg_interleave_B16_L1024_C12": (LLAMA2_7B, 16, 1024, 12, 1024, 128, "semantic_segmentation", "interleave"),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E spec (6.29 TB/s is the measured achievable, same guide)


def workload(name):
    """(hf cfg, B per GPU, L, C, pred_len, n_tok, task, covariate mode)"""
    w = WORKLOADS[name]
    return w if len(w) == 8 else w + ("concat",)


def model_cfg(L, pred, task="forecasting", cov="concat"):
    return {
        "DEBUG": True, "task": task, "model": "medtsllm", "history_len": L, "pred_len": pred,
        "training": {"dropout": 0.1}, "setup": {"dtype": "mixed"},     # every shipped reference config trains at 0.1 (configs/datasets/*.toml)
        "models": {"timellm": {
            "d_model": 32, "d_ff": 128, "n_heads": 8, "num_tokens": 1024, "covariate_mode": cov,
            "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8},
            "prompting": {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False,
                          "input_stats_dim": 0, "input_stats_select": "all"},
            "llm": {"enabled": True, "llm": "random-init", "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False},
        }},
    }


class DS:
    def __init__(self, C_, n_classes=0):
        self.description, self.n_features, self.n_classes, self.task_description = "synthetic", C_, n_classes, None


def make_batch(B, L, C_, pred, seed, device, task="forecasting"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, C_, generator=g) + (torch.rand(C_, generator=g) * 4 - 2)
    y = torch.randn(B, pred, C_, generator=g) if task != "semantic_segmentation" else torch.randint(0, 4, (B, pred), generator=g)
    return {"x_enc": x.to(device), "y": y.to(device)}


def flops_per_step(cfg, B, T, P, C_, n_out, V, S=1024, d_model=32, d_ff=128, H=8, cov="concat", n_cached=0):
    """Algorithmic fwd+bwd FLOPs: frozen GEMMs 2x fwd, attention 3x fwd, trainables 3x fwd, mapping 2x.
    P = patch rows per sample in the LLM sequence (interleave: C_ times the per-channel count). n_cached: leading prompt rows whose
    forward is served from the prompt-row cache (executed FLOPs only; the algorithmic count never takes a discount)."""
    if cfg["model_type"] == "llama":
        d, L_, ffn = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["intermediate_size"]
        M = B * T
        kv = cfg["num_key_value_heads"] * (d // cfg["num_attention_heads"])
        gemm_fwd = L_ * 2 * M * (2 * d * d + 2 * d * kv + 3 * d * ffn)
    else:
        d, L_, ffn = cfg["n_embd"], cfg["n_layer"], 4 * cfg["n_embd"]
        M = B * T
        gemm_fwd = L_ * 2 * M * (d * 3 * d + d * d + 2 * d * ffn)
    attn_fwd = L_ * 4 * B * T * T * d
    HE = H * d_ff
    q_width = C_ * d_model if cov == "concat" else d_model          # (interleave: P already counts every channel's patches)
    front = 2 * B * P * q_width * HE + 2 * 2 * S * d * HE + 4 * B * P * S * HE + 2 * B * P * HE * d
    tail = 2 * B * P * d * d_ff + 2 * B * d_ff * P * n_out
    mapping = 2 * S * V * d
    algorithmic = 2 * gemm_fwd + 3 * attn_fwd + 3 * (front + tail) + 2 * mapping
    # executed with the exact dead-gradient elimination: backward GEMMs on the P patch rows only; attention backward keeps
    # dQ of the patch queries (3/4 of the causal area) and dK/dV of the patch keys (1/4) -> about half of its FLOPs
    Tq = T - n_cached
    # forward attention of the computed rows only: queries Tq x keys T, minus nothing else (causal counted as the full rectangle)
    executed = gemm_fwd * (Tq / T + P / T) + attn_fwd * (Tq / T + 2 * 0.5) + 3 * (front + tail) + 2 * mapping
    return algorithmic, executed


def cpu_baseline(hf_cfg, sd, L, C_, pred, n_tok, prompt_ids, max_seconds=30.0, task="forecasting", Bs=4, cov="concat"):
    """Oracle fwd+bwd on the host cores on a bounded sample of the same workload (same shapes, smaller batch)."""
    from oracle import medtsllm_oracle as O
    g = torch.Generator().manual_seed(123)
    d = hf_cfg["n_embd"] if hf_cfg["model_type"] == "gpt2" else hf_cfg["hidden_size"]
    n_out = pred * C_ if task != "semantic_segmentation" else pred * 4
    p = {
        "patch_embedding.value_embedding.tokenConv.weight": torch.randn(32, 16, 3, generator=g) * 0.2,
        # (vocabularies > 100 000: the reference maps from 100 000 linspace-sampled rows, R:models/medtsllm.py:219-222)
        "mapping_layer.weight": torch.randn(1024, min(hf_cfg["vocab_size"], 100_000), generator=g) * 0.01,
        "mapping_layer.bias": torch.zeros(1024),
        "reprogramming_layer.query_projection.weight": torch.randn(1024, (C_ if cov == "concat" else 1) * 32, generator=g) * 0.05,
        "reprogramming_layer.query_projection.bias": torch.zeros(1024),
        "reprogramming_layer.key_projection.weight": torch.randn(1024, d, generator=g) * 0.03,
        "reprogramming_layer.key_projection.bias": torch.zeros(1024),
        "reprogramming_layer.value_projection.weight": torch.randn(1024, d, generator=g) * 0.03,
        "reprogramming_layer.value_projection.bias": torch.zeros(1024),
        "reprogramming_layer.out_projection.weight": torch.randn(d, 1024, generator=g) * 0.03,
        "reprogramming_layer.out_projection.bias": torch.zeros(d),
        "embedding_downsample_layer.weight": torch.randn(128, d, generator=g) * 0.03,
        "embedding_downsample_layer.bias": torch.zeros(128),
    }
    P = ((L + 8 - 16) // 8 + 1) * (C_ if cov == "interleave" else 1)
    p["output_projection.linear.weight"] = torch.randn(n_out, 128 * P, generator=g) * 0.01
    p["output_projection.linear.bias"] = torch.zeros(n_out)
    we_kw = {}
    if hf_cfg["vocab_size"] > 100_000:      # ... and that sub-sampled table is a TRAINABLE parameter there (Llama-3), so its gradient is part of the step
        p["word_embeddings"] = O.word_embeddings_of(sd, hf_cfg).detach().float().clone()
        we_kw = {"word_emb": p["word_embeddings"]}
    for t in p.values():
        t.requires_grad_(True)
    semseg = task == "semantic_segmentation"
    m = dict(task=task, pred_len=pred, patch_len=16, stride=8, n_heads=8, d_ff=128, covariate_mode=cov,
             embedding_downsample_mode="linear", n_outputs_per_step=4 if semseg else C_, n_classes=4 if semseg else 0)
    b = make_batch(Bs, L, C_, pred, 7, "cpu", task)
    tok = [[prompt_ids] for _ in range(Bs)]

    def step():
        out = O.medtsllm_forward(b["x_enc"], p, sd, hf_cfg, m, token_ids=tok, pad_token_id=0, training=True, **we_kw)
        if semseg:
            torch.nn.functional.cross_entropy(out.permute(0, 2, 1), b["y"]).backward()
        else:
            torch.nn.functional.mse_loss(out, b["y"]).backward()
        for t in p.values():
            t.grad = None

    # pick the thread count that runs this workload fastest (256 SMT threads are far slower than ~1 per core-complex)
    ncpu = os.cpu_count() or 1
    best = None
    # (`cores` below = the threads actually used, `host_cores` = os.cpu_count() of the box)
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        if best is None:
            step()  # warm-up (first touch, allocator)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
        if time.perf_counter() - t0 > max_seconds / 4:
            break
    torch.set_num_threads(best[0])
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < max_seconds / 3 and n < 10):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(Bs / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} timed fwd+bwd steps of B={Bs} windows [L={L}, C={C_}] (same model/shapes as the GPU workload, fp32, "
                      f"plain-torch oracle; optimizer step excluded), {dt:.2f} s/step",
            # measured in the build container (8 cores, the only machine where both run): the real reference does the same B = 4 step 1.05 x
            # faster than this port (profiles/r04_cpu_reference_vs_oracle_port.txt, tools/ref_vs_oracle_cpu.py)
            "reference_speed_over_port": reference_speed_over_port()}


def reference_speed_over_port():
    """(oracle port time) / (real reference time) of the same CPU step, read from the committed measurement (None if the record is missing)"""
    import re
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_oracle_port.txt")) as f:
            m = re.search(r"oracle/reference time ratio ([0-9.]+)", f.read())
        return float(m.group(1)) if m else None
    except OSError:
        return None


def cpu_baseline_llama(hf_cfg, L, C_, pred, n_tok, prompt_ids, task, cov="concat"):
    """the Llama configs' CPU figure, EXTRAPOLATED — a 32-layer 7B/8B stack in fp32 does not fit a bounded
    sample, so the same oracle step is timed with the stack cut to 2 layers and to 1 layer (llm_layers, R:models/medtsllm.py:145-146; same
    front end, mapping GEMM and head); a layer costs t2 - t1, the full depth t1 + (n_layers - 1) (t2 - t1). Also reported: the cruder
    "2 layers x n_layers / 2" figure of BASELINE.md (it scales the front end and the head along with the stack, i.e. favours the GPU)."""
    from med_ts_llm_amd.models.backbone import random_state_dict
    n_layers = hf_cfg["num_hidden_layers"]
    out = {}
    times = {}
    for k in (1, 2):
        cfg_k = dict(hf_cfg, num_hidden_layers=k)
        sd = random_state_dict(cfg_k, seed=0, std=0.02)
        r = cpu_baseline(cfg_k, sd, L, C_, pred, n_tok, prompt_ids, max_seconds=12.0, task=task, Bs=2, cov=cov)
        times[k] = 2.0 / r["value"]
        out = r
        del sd
    per_layer = max(times[2] - times[1], 1e-9)
    full = times[1] + (n_layers - 1) * per_layer
    out.update({"value": round(2.0 / full, 4), "kind": "port", "extrapolated": True,
                "sample": f"B=2 windows [L={L}, C={C_}], fp32 plain-torch oracle, stack cut to 1 and 2 layers: {times[1]:.2f} s and {times[2]:.2f} s per "
                          f"fwd+bwd step -> {per_layer:.2f} s per layer, {full:.1f} s per step at {n_layers} layers (EXTRAPOLATED, not run)",
                "two_layers_times_half_depth_value": round(2.0 / (times[2] * n_layers / 2), 4)})
    return out


def csrc_sha16():
    """hash of the kernel sources: PMC traffic tables are only valid for the library they were measured with"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "med-ts-llm_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "medtsllm_hip.h")]):
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this workload (profiles/pmc_traffic_<workload>.json, written
    by tools/profile_round.sh from separate rocprofv3 --pmc runs), or None when there is no table for the CURRENT kernel sources."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    if table.get("_meta", {}).get("csrc_sha16") != csrc_sha16():
        return None                      # stale: measured with other kernel sources
    return table_lookup(table, kernel)


def table_lookup(table, kernel):
    """a PMC table's entry for one of the launch profiler's kernel names: the profiler names an instance by the template arguments that
    select it ("norm_bwd_kernel<3, false>"), rocprofv3 by all of them ("norm_bwd_kernel<3, false, 1>") — exact match first, then the one
    entry whose name extends the profiler's argument list"""
    if kernel in table:
        return table[kernel]
    if kernel.endswith(">"):
        ext = [k for k in table if k.startswith(kernel[:-1] + ",")]
        if len(ext) == 1:
            return table[ext[0]]
    return None


def live_pmc_traffic(workload, kernels, timeout_s=150.0, extra_args=()):
    """HBM bytes per launch of `kernels`, measured NOW by this run: two child runs of this very workload (3 steps, nothing else) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` — separate passes, counters only beside the kernel trace, units and
    the gfx950 read-side correction as GUIDE MI355X_MICROARCH §HBM prescribes (tools/pmc_traffic.py holds the arithmetic). Returns
    ({kernel: {...}}, note, step totals for roofline_step); ({}, reason, None) when rocprofv3 is absent, a pass fails or runs out of time — the caller then falls back to the
    committed table of the same kernel sources."""
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found", None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic as pt
    t_start = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="mtl_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", os.path.join(tmp, counter), "-o", "r", "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-extra-configs", "--no-live-traffic",
               *extra_args]
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)      # the process group this call started, nothing else
            p.wait()
            shutil.rmtree(tmp, ignore_errors=True)
            return {}, f"{counter} pass exceeded {timeout_s:.0f} s", None
        import glob
        dbs = glob.glob(os.path.join(tmp, counter, "**", "*.db"), recursive=True)
        if rc != 0 or not dbs:
            shutil.rmtree(tmp, ignore_errors=True)
            return {}, f"{counter} pass failed (rc {rc})", None
        per[counter] = {pt.clean(k): v for k, v in pt.per_kernel(dbs[0], counter, warmup_steps=1).items()}      # (the child's warm-up step and set-up dropped)
    shutil.rmtree(tmp, ignore_errors=True)
    # every dispatch of the two counted steps summed: the step-level HBM traffic (roofline_step)
    n_steps = 2
    step_rd = 2.0 * 1024.0 * sum(v * n for v, n in per["FETCH_SIZE"].values()) / n_steps
    step_wr = 1024.0 * sum(v * n for v, n in per["WRITE_SIZE"].values()) / n_steps
    step = {"hbm_read_bytes": step_rd, "hbm_write_bytes": step_wr, "hbm_bytes": step_rd + step_wr,
            "dispatches_per_step": sum(n for _, n in per["FETCH_SIZE"].values()) / n_steps}
    out = {}
    for k in kernels:
        fk, n = table_lookup(per["FETCH_SIZE"], k) or (None, 0)
        wk, _ = table_lookup(per["WRITE_SIZE"], k) or (None, 0)
        if fk is None or wk is None:
            continue
        rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
        out[k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "dispatches": n}
    return out, (f"live: two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE x2 x 1024 B; WRITE_SIZE x 1024 B) of a 3-step child run of this workload, "
                 f"mean per launch over the two steps after the warm-up step, {time.perf_counter() - t_start:.0f} s"), step


def roofline_step(step_traffic, line, note):
    """the WHOLE step against the HBM roofline (the GPT-2 step is HBM-bound as a whole): bytes every dispatch of one step moved
    (PMC, live: the two counted steps of the child run, all kernels summed) / the timed region's step time, and the step's arithmetic intensity next
    to the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B)."""
    if not step_traffic:
        return None
    b = step_traffic["hbm_bytes"]
    sec = line["ms_per_step"] * 1e-3
    fl = line["executed_tflop_per_step_per_gpu"] * 1e12
    return {"bound": "hbm", "bytes_per_step": int(b), "read_bytes_per_step": int(step_traffic["hbm_read_bytes"]), "write_bytes_per_step": int(step_traffic["hbm_write_bytes"]),
            "achieved": round(b / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / sec / 1e9 / HBM_PEAK_GBS, 4),
            "flop_per_byte": round(fl / b, 1), "ridge_flop_per_byte": round(MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9), 1),
            "dispatches_per_step": round(step_traffic["dispatches_per_step"], 1), "traffic_src": "live", "source": note}


def causal_fraction(kernel, T, Tq):
    """executed / full-rectangle work of a CAUSAL attention launch whose Tq queries are the last rows of a T-key sequence: query i (0-based
    among the Tq) sees T - Tq + i + 1 keys. 1.0 for the non-causal (reprogramming) instances. Tile granularity is ignored (the kernels
    also execute the masked half of the 32-key slabs on the diagonal: < 2 % at T = 1664)."""
    # (the resident and the 32-rows-per-wave kernels exist only as causal instances; the others carry CAUSAL as their second template argument)
    causal = ("_res" in kernel) or ("_w32" in kernel) or (", true," in kernel.split("<", 1)[-1][:12])
    if not causal or not T or not Tq:
        return 1.0
    return (Tq * (T - Tq) + Tq * (Tq + 1) / 2.0) / (Tq * T)


def roofline_objects(rows, workload, geom=None):
    """(roofline, roofline_hbm, all instances) from the launch profiler's rows: the dominant kernel of each family.
    geom = (T, Tq of the forward launches, Tq of the backward launches): lets the attention object carry EXECUTED FLOP/s beside the
    full-rectangle figure."""
    def obj(r, peak, unit, scale):
        per_launch_ms = r["total_ms"] / r["launches"]
        achieved = r["work"] / (r["total_ms"] * 1e-3) / scale
        return {"bound": "mfma" if r["kind"] == "flops" else "hbm", "kernel": r["kernel"], "achieved": round(achieved, 1), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4), "traffic": pmc_traffic(workload, r["kernel"]), "launches": r["launches"],
                "avg_launch_us": round(per_launch_ms * 1e3, 2), "min_launch_us": round(r["min_ms"] * 1e3, 2),
                ("flops_per_launch" if r["kind"] == "flops" else "algorithmic_bytes_per_launch"): r["work"] / r["launches"],
                "timing": "per-dispatch start/stop events (hipExtLaunchKernelGGL) on the launch stream: kernel begin -> end, as rocprofv3 --kernel-trace reports it"}
    mf = [r for r in rows if r["kind"] == "flops" and r["kernel"].startswith("gemm")]
    hb = [r for r in rows if r["kind"] == "bytes" and r["kernel"].startswith("norm")]
    roof = obj(max(mf, key=lambda r: r["total_ms"]), MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", 1e12) if mf else None
    roof_hbm = obj(max(hb, key=lambda r: r["total_ms"]), HBM_PEAK_GBS, "GB/s", 1e9) if hb else None
    # the dominant attention kernel against the same MFMA peak (full-rectangle FLOP convention of SURVEY 8d: a causal kernel executes half of
    # them): at T = 256 these kernels are latency / fill-bound, at long T (interleave covariates, T = 1664) they are the MFMA-bound flash regime
    at = [r for r in rows if r["kind"] == "flops" and r["kernel"].startswith("attn")]
    roof_attn = obj(max(at, key=lambda r: r["total_ms"]), MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", 1e12) if at else None
    if roof_attn is not None:
        roof_attn["flops_convention"] = "full rectangle (4 T_q T_k d per head forward; backward kernels 2x split evenly): causal kernels execute about half"
        if geom:
            T_, tq_f, tq_b = geom
            cf = causal_fraction(roof_attn["kernel"], T_, tq_f if "fwd" in roof_attn["kernel"] else tq_b)
            roof_attn["executed_fraction_of_rectangle"] = round(cf, 4)
            roof_attn["executed_achieved"] = round(roof_attn["achieved"] * cf, 1)
            roof_attn["executed_frac"] = round(roof_attn["frac"] * cf, 4)
    def inst_row(r):
        o = {"kernel": r["kernel"], "launches": r["launches"], "avg_us": round(r["total_ms"] / r["launches"] * 1e3, 2)}
        if r["kind"] == "flops":
            o["tflops"] = round(r["work"] / (r["total_ms"] * 1e-3) / 1e12, 1)
            if geom and r["kernel"].startswith("attn"):
                o["tflops_executed"] = round(o["tflops"] * causal_fraction(r["kernel"], geom[0], geom[1] if "fwd" in r["kernel"] else geom[2]), 1)
        else:
            o["gbs"] = round(r["work"] / (r["total_ms"] * 1e-3) / 1e9, 1)
        return o
    inst = [inst_row(r) for r in sorted(rows, key=lambda r: -r["total_ms"])]
    return roof, roof_hbm, inst, roof_attn


def trainer_loop_rate(model, opt, sync, su, task, device, world, rank, dev_batches, steps, B):
    """samples/s of BaseTask.train_step driven the way get_trainer(...).train() drives it: batches start in (pinned) HOST memory, every step
    logs its loss. Two figures: the product default (the loss is copied out asynchronously and logged one step later: no queue drain) and the
    reference's blocking `loss.item()` per step (setup.deferred_loss_log = false)."""
    from med_ts_llm_amd.tasks import task_lookup
    from med_ts_llm_amd.utils import dict_to_object
    cls = task_lookup[task if task != "anomaly_detection" else "reconstruction"]
    host_batches = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in dev_batches]

    class Quiet:
        history = []

        def log_scores(self, scores):
            self.history.append(scores["train/loss"])

    out = {}
    for label, deferred in (("samples_per_s", True), ("samples_per_s_blocking_loss_item", False)):
        tr = cls.__new__(cls)
        tr.device, tr.dtype, tr.mixed, tr.model, tr.optimizer, tr.grad_sync = device, torch.float32, True, model, opt, sync
        tr.config = dict_to_object({"training": {"batch_size": B * world}, "setup": {"deferred_loss_log": deferred}})
        tr.logger, tr.step, tr.rank, tr.world_size, tr.opt_shards = Quiet(), 0, rank, world, su
        if task == "semantic_segmentation":
            tr.loss_fn = torch.nn.CrossEntropyLoss()
            tr.compute_loss = lambda inputs, tr=tr: tr.loss_fn(tr.model(inputs).permute(0, 2, 1), inputs["y"])
        else:
            tr.loss_fn = torch.nn.MSELoss()
            if task == "forecasting":
                tr.compute_loss = lambda inputs, tr=tr: tr.loss_fn(tr.model(inputs), inputs["y"])
        for i in range(2):
            tr.train_step(host_batches[i % len(host_batches)])
        tr._flush_losses()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.train_step(host_batches[i % len(host_batches)])
        tr._flush_losses()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out[label] = round(B * world * steps / dt, 2)
        assert len(tr.logger.history) >= steps
    out["steps"] = steps
    out["what"] = ("BaseTask.train_step (tasks/base.py): prepare_batch from pinned host memory + autocast + forward + loss + backward + optimizer + "
                   "zero_grad + per-step loss logging; default = loss logged one step late from an async copy, blocking = loss.item() every step")
    return out


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-run under torch.distributed.run, one rank per GPU. With fewer GPUs than
    ranks (a 1-GPU test box) the ranks share devices over gloo, since RCCL refuses two ranks per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
        env.setdefault("MTL_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def run_workload(name, args, ctx, steps, warmup, want_cpu, want_roofline, legs=True, stage=lambda msg: None):
    """one bench line (dict, rank 0; None elsewhere) for WORKLOADS[name]: `warmup` untimed steps, then exactly `steps` timed steps
    bracketed by barrier + synchronize on both sides, MAX over ranks; then (outside the timed region) a profiled replay."""
    from med_ts_llm_amd import parallel
    from med_ts_llm_amd.hip import _native
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    rank, world, device = ctx
    hf_cfg, B, L, C_, pred, n_tok, task, cov = workload(name)
    big = hf_cfg["model_type"] == "llama"
    sd = random_state_dict(hf_cfg, seed=0, std=0.02, device=device if big else "cpu", dtype=torch.bfloat16 if big else torch.float32)
    torch.manual_seed(0)
    model = model_lookup["medtsllm"](dict_to_object(model_cfg(L, pred, task, cov)), DS(C_, 4 if task == "semantic_segmentation" else 0),
                                     backbone_state=(hf_cfg, sd)).to(device)
    # --dtype bf16: the reference's setup.dtype = "bf16" (R:tasks/base.py:261-262,205-208): parameters, inputs and — natively since round 6 — the
    # residual stream of the frozen stack in bf16, no autocast. A separately reported configuration: the headline stays the reference default "mixed".
    pure_bf16 = getattr(args, "dtype", "mixed") == "bf16"
    if pure_bf16:
        model = model.to(torch.bfloat16)
    prompt_ids = torch.randint(0, hf_cfg["vocab_size"], (1, n_tok), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    model.fixed_prompt_ids = prompt_ids
    model.prune_dead_prompt_grads = not args.full_backward
    model.llm_dropout = not args.no_llm_dropout
    model.train()
    # --dp-plumbing (N = 1): the whole data-parallel machinery in a ONE-rank RCCL group — every collective of the N-rank step is issued and
    # waited for (sums over one rank): what the DP plumbing itself costs per step, apart from the wire
    force = bool(getattr(args, "dp_plumbing", False)) and world == 1
    dp = world > 1 or force
    sharded = dp and not args.replicate_mapping and model.shard_mapping_layer(rank, world, None, force)
    torch.manual_seed(1234 + 7919 * rank)      # rank-specific dropout streams (weights above were built from one seed)
    params = [p for p in model.parameters() if p.requires_grad]
    # DP: row-sharded optimiser step for the big replicated tensors (wide flatten heads, Llama-3's trainable vocabulary): reduce-scatter,
    # Adam on the owned rows, all-gather of the bf16 copy the forward reads
    su = parallel.ShardedUpdate(list(model.named_parameters()), rank, world, force_collectives=force) if (dp and not args.replicate_optimizer) else None
    if su is not None:
        model._opt_shards = su             # (the model waits for asynchronously published rows in front of the GEMM that reads them)
    opt_params = su.optimizer_params(params) if su is not None else params
    if args.torch_adam:
        opt = torch.optim.Adam(opt_params, lr=1e-4, fused=True)
    else:
        from med_ts_llm_amd.hip.optim import HipAdam, Bf16Shadow
        opt = HipAdam(opt_params, lr=1e-4)
        if args.optimizer_overlap and su is None:
            opt.defer(model.late_parameters())          # the tail's parameters: updated on a side stream under the next step's front end + backbone
            model.optimizer_wait = opt.wait_deferred
        for sh in model.bf16_shadows():
            if su is not None and id(sh.param) in su._by_param:
                opt.register_shadow(Bf16Shadow(su._by_param[id(sh.param)]["shard"], su.attach_shadow(sh.param, sh.tensor)))
            else:
                opt.register_shadow(sh)
    sync = parallel.FlatGradAllReduce(params, force_collectives=force) if dp else None
    loss_fn = torch.nn.MSELoss() if task != "semantic_segmentation" else torch.nn.CrossEntropyLoss()
    batches = [make_batch(B, L, C_, pred, 1000 + rank * 97 + i, device, task) for i in range(4)]
    if pure_bf16:
        batches = [{k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in b.items()} for b in batches]

    def step(i):
        inputs = batches[i % len(batches)]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not pure_bf16):
            pred_ = model(inputs)
            loss = loss_fn(pred_ if task != "semantic_segmentation" else pred_.permute(0, 2, 1), inputs["y"])
        loss.backward()
        if sync is not None:
            sync()
        if su is not None:
            su.sync()
        opt.step()
        if su is not None:
            su.publish(async_op=True)      # the model waits for a tensor's rows in front of the first kernel that reads it
        opt.zero_grad()
        return loss

    for i in range(warmup):
        step(i)

    def fence():
        if su is not None:
            su.wait_published()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    # the prompt-row cache state of the TIMED steps (read here: the predict leg below runs its last forward with the cache switched off)
    n_cached = int(getattr(model.backbone, "last_n_prefix", 0))

    roofline = roofline_hbm = roofline_attn = instances = optimizer_ms = None
    if want_roofline:
        # profiled replay (outside the timed region). EVERY rank replays the steps — they contain collectives — but only rank 0
        # records: each GEMM / attention / norm / optimiser launch carries its own start/stop event pair
        lib = _native.lib()
        n_replay = min(steps, 5)
        if rank == 0:
            lib.mtl_prof_enable(2 if os.environ.get("MTL_PROF_SHAPES") else 1)      # (tools/gemm_shapes.py: per-shape GEMM rows; read here, not by the library)
        for i in range(n_replay):
            step(i)
        torch.cuda.synchronize()
        rows = _native.prof_rows() if rank == 0 else []
        lib.mtl_prof_enable(0)
        if rows:
            P_ = ((L + 8 - 16) // 8 + 1) * (C_ if cov == "interleave" else 1)
            roofline, roofline_hbm, instances, roofline_attn = roofline_objects(rows, name, (n_tok + P_, n_tok + P_ - n_cached, P_ if not args.full_backward else n_tok + P_))
            adam = [r for r in rows if r["kernel"].startswith("adam")]
            optimizer_ms = sum(r["total_ms"] for r in adam) / n_replay if adam else None

    # the product trainer's own loop body (tasks/base.py::train_step = R:tasks/forecasting.py:19-30): prepare_batch from HOST memory
    # (H2D + cast), autocast, forward, loss, backward, [all-reduce], optimizer, zero_grad, per-step loss logging. `value` above is the same
    # step without the host batch and the logging; this is what a user of get_trainer(...).train() gets.
    stage(f"{name}: timed region {elapsed:.2f} s + profiled replay done")
    loop = None
    if want_roofline and legs:
        loop = trainer_loop_rate(model, opt, sync, su, task, device, world, rank, batches, min(steps, 10), B)

    # forward-only (predict) rate in eval mode, with and without the prompt-row cache (outside the timed region; every rank runs it)
    predict = None
    if want_roofline and legs:
        model.eval()
        predict = {}
        with torch.no_grad():
            for label, on in (("samples_per_s", True), ("samples_per_s_full_forward", False)):
                model.prompt_row_cache = on
                for i in range(2):
                    model(batches[i % len(batches)])
                torch.cuda.synchronize()
                n_pred = max(3, min(steps, 10))
                t0 = time.perf_counter()
                for i in range(n_pred):
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        model(batches[i % len(batches)])
                torch.cuda.synchronize()
                predict[label] = round(B * n_pred / (time.perf_counter() - t0), 2)
                if on:
                    predict["cached_prompt_rows"] = int(getattr(model.backbone, "last_n_prefix", 0))
        model.prompt_row_cache = True
        predict["what"] = "model.eval() forward under no_grad (the predict() path), per GPU; second figure: prompt-row cache switched off"
        model.train()

    stage(f"{name}: trainer-loop / predict legs done")
    cpu = None
    if rank == 0 and world == 1 and want_cpu and big:
        cpu = cpu_baseline_llama(hf_cfg, L, C_, pred, n_tok, prompt_ids[0].tolist(), task, cov)
    if rank == 0 and world == 1 and want_cpu and not big:
        cpu = cpu_baseline(hf_cfg, sd, L, C_, pred, n_tok, prompt_ids[0].tolist())
        if name == "gpt2s_B32_L1024_C12" and args.full_detail:     # BASELINE.json configs[0]: the reference's CPU-runnable case, timed beside the metric workload
            _, _, L1, C1, pred1, n_tok1, _, _ = workload("gpt2s_etth1_B32_L512_C7")
            cpu["configs"] = [dict(cpu_baseline(hf_cfg, sd, L1, C1, pred1, n_tok1, prompt_ids[0].tolist(), max_seconds=20.0),
                                   workload="gpt2s_etth1_B32_L512_C7 (BASELINE.json configs[0]: ETTh1-shaped [B, 512, 7] forecasting, GPT-2-small, CPU fp32)")]

    stage(f"{name}: cpu baseline done")
    out = None
    if rank == 0:
        P = ((L + 8 - 16) // 8 + 1) * (C_ if cov == "interleave" else 1)
        T = n_tok + P
        fl, fl_exec = flops_per_step(hf_cfg, B, T, P, C_, pred * (C_ if task != "semantic_segmentation" else 4), min(hf_cfg["vocab_size"], 100_000),
                                     cov=cov, n_cached=n_cached)
        if args.full_backward:
            fl_exec = fl
        value = B * world * steps / elapsed
        out = {
            "metric": "samples/sec (1024-step, 12-ch windows) through MedTsLLM fwd+bwd", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "setup_dtype": "bf16" if pure_bf16 else "mixed",
            "config": {"workload": f"{name}{'@bf16' if pure_bf16 else ''}: [B={B}/GPU, L={L}, C={C_}] windows, P={P}, prompt {n_tok} tok -> T={T}, "
                                   f"frozen {name.split('_')[0] + '-' + name.split('_')[1] if big else 'GPT-2-small'} (random init) backbone, {cov} covariates, "
                                   f"{task} pred_len={pred}, training.dropout=0.1 (patch-embedding + reprogramming-attention dropout live"
                                   f"{', GPT-2 embd/attn/resid dropouts live' if (not big and not args.no_llm_dropout) else ''}), "
                                   f"step = fwd+loss+bwd+{'allreduce+' if world > 1 else ''}Adam", "global_batch": B * world,
                       "parallelism": f"dp{world}" + (" (mapping layer row-sharded)" if sharded else "")},
            "per_gpu_samples_per_s": round(value / world, 2),
            "dist_backend": (dist.get_backend() if dp else None),
            **({"dp_plumbing": "one-rank RCCL group with every collective of the N-rank step issued (mapping layer 'row-sharded' over the one rank: all-gather + "
                               "all-reduce; bucketed flat gradient all-reduce from the gradient hooks; reduce-scatter / Adam on owned rows / asynchronous bf16 "
                               "all-gather for the big tensors): value / the plain N = 1 value = what the DP machinery costs apart from the wire"} if force else {}),
            "final_loss": final_loss,
            "backward": "full (incl. unused prompt-row input gradients)" if args.full_backward else
                        "exact dead-gradient elimination: prompt rows never depend on a trainable parameter, their input gradient is not computed",
            "forward": (f"prompt-row cache: the {n_cached} prompt rows are one constant prompt shared by every sample and the stack is deterministic, so their "
                        "per-layer keys / values are cached and the forward runs on the patch rows only (executed FLOPs drop; the algorithmic count and every "
                        "roofline denominator keep the full sequence)") if n_cached else "full sequence",
            "trainer_loop": loop, "predict": predict,
            "algorithmic_tflop_per_step_per_gpu": round(fl / 1e12, 3), "executed_tflop_per_step_per_gpu": round(fl_exec / 1e12, 3),
            "step_mfma_frac": round(fl_exec / (elapsed / steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "step_mfma_frac_algorithmic": round(fl / (elapsed / steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "optimizer_ms_per_step": optimizer_ms, "roofline": roofline, "roofline_hbm": roofline_hbm, "roofline_attention": roofline_attn,
            "kernel_instances": instances,
            "cpu_baseline": cpu,
        }
    del model, opt, sync, batches, sd, params, su, opt_params
    torch.cuda.empty_cache()
    return out


def _short_workload(line):
    """'name: [B=32/GPU, L=.., C=..] windows, P=.., prompt .. -> T=.., frozen X backbone, ...' -> 'name: [B=32/GPU, L=.., C=..] T=.. X'"""
    import re
    w = line["config"]["workload"]
    m = re.match(r"(\S+: \[[^\]]*\]).*?(T=\d+), frozen (\S+)", w)
    return f"{m.group(1)} {m.group(2)} {m.group(3)}" if m else w[:120]


def _roof_compact(o):
    """a roofline object for the compact line: numbers only, `traffic` = HBM bytes per launch (PMC) or null"""
    if not o:
        return None
    t = o.get("traffic")
    c = {"bound": o["bound"], "kernel": o["kernel"], "achieved": o["achieved"], "peak": o["peak"], "unit": o["unit"], "frac": o["frac"],
         "traffic": (round(t["hbm_bytes"]) if isinstance(t, dict) else None),
         "traffic_read": (round(t["hbm_read_bytes"]) if isinstance(t, dict) else None),
         "traffic_write": (round(t["hbm_write_bytes"]) if isinstance(t, dict) else None),
         "traffic_src": ("live" if str(o.get("traffic_source", "")).startswith("live") else ("committed" if t is not None else None)),
         "launches": o["launches"], "avg_launch_us": o["avg_launch_us"]}
    for k in ("flops_per_launch", "algorithmic_bytes_per_launch", "executed_achieved", "executed_frac"):
        if k in o:
            c[k] = int(round(o[k])) if k.endswith("per_launch") else o[k]
    return c


def _check_line(line):
    """a whole-step MFMA fraction above its fastest MFMA kernel's is impossible: the executed-FLOP count would be wrong (r03 reported 0.64 for a
    0.43 step because the prompt-row cache state was read after it had been switched off)"""
    inst = line.get("kernel_instances") or []
    best = max([r["tflops"] for r in inst if "tflops" in r and r["kernel"].startswith("gemm")], default=None)
    if best is None:
        return None
    return bool(line["step_mfma_frac"] <= best / MFMA_BF16_PEAK_TFLOPS + 1e-9)


def _short_kernel(k):
    return k.replace("gemm_nt_persist_kernel", "gemm_nt_persist").replace("_kernel", "").replace(", ", ",") if k else k


def compact_line(out):
    """the ONE stdout line (< 4 KB): the contract's fields, the three roofline objects as numbers, the CPU baseline, and one short summary per
    extra config. Everything else (kernel instances, prose, trainer-loop / predict legs) goes to the detail file."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                             "dtype", "data")}
    c["config"] = {"workload": _short_workload(out), "global_batch": out["config"]["global_batch"], "parallelism": out["config"]["parallelism"]}
    for k in ("per_gpu_samples_per_s", "dist_backend", "rccl_ranks", "dp_mode", "dp_modes", "dp_sharded_failed"):
        if out.get(k) is not None:
            c[k] = out[k]
    c["roofline"] = _roof_compact(out.get("roofline"))
    c["roofline_hbm"] = _roof_compact(out.get("roofline_hbm"))
    c["roofline_attention"] = _roof_compact(out.get("roofline_attention"))
    for k in ("roofline_hbm", "roofline_attention"):          # (the read / write split and the launch count of these two: detail file)
        if c[k]:
            for d in ("traffic_read", "traffic_write", "launches"):
                c[k].pop(d, None)
    # every HBM-bound kernel family beside the dominant one (roofline_hbm picks the norm kernel with the largest total time; the other direction and Adam here)
    c["hbm_kernels"] = [{"kernel": r["kernel"], "us": r["avg_us"], "frac": round(r["gbs"] / HBM_PEAK_GBS, 3)}
                        for r in (out.get("kernel_instances") or []) if "gbs" in r and r["kernel"].startswith(("norm", "adam"))][:3]
    rs = out.get("roofline_step")
    c["roofline_step"] = {k: v for k, v in rs.items() if k != "source"} if rs else None
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "host_cores": cb.get("host_cores"), "kind": cb["kind"],
                             "sample": cb["sample"][:72]}
        if cb.get("configs"):
            c["cpu_baseline"]["configs0_etth1_value"] = cb["configs"][0]["value"]
    else:
        c["cpu_baseline"] = None
    c["tflop_per_step"] = {"algorithmic": out["algorithmic_tflop_per_step_per_gpu"], "executed": out["executed_tflop_per_step_per_gpu"]}
    c["step_mfma_frac"] = out["step_mfma_frac"]
    c["step_mfma_frac_algorithmic"] = out["step_mfma_frac_algorithmic"]
    c["optimizer_ms_per_step"] = round(out["optimizer_ms_per_step"], 4) if out.get("optimizer_ms_per_step") else None
    c["checks"] = {"step_frac_le_best_kernel_frac": _check_line(out)}
    c["configs"] = []
    for e in out.get("configs", []):
        r = e.get("roofline") or {}
        ra = e.get("roofline_attention") or {}
        cb = e.get("cpu_baseline")
        wl = _short_workload(e)
        entry = {
            "workload": wl.split(":")[0] + (" " + wl.split("] ", 1)[1].split(" ")[0] if "] " in wl else ""),       # 'name T=..' (the name carries backbone, B, L, C)
            "value": e["value"], "steps": e["steps"], "warmup": e["warmup"], "ms_per_step": e["ms_per_step"],
            "prompt_row_cache": str(e.get("forward", "")).startswith("prompt-row cache"),
            # (executed FLOPs only: with the prompt-row cache the algorithmic count is twice the executed one and is not a utilisation; the detail file keeps it)
            "step_mfma_frac": e["step_mfma_frac"],
            "roofline": {"kernel": _short_kernel(r.get("kernel")), "achieved": r.get("achieved"), "frac": r.get("frac"), "avg_launch_us": r.get("avg_launch_us")},
            "cpu_baseline": ({"value": cb["value"], "cores": cb["cores"], "extrapolated": cb.get("extrapolated", False)} if cb else None)}
        if not _check_line(e):                     # (self-check: a whole-step MFMA fraction above the fastest GEMM's would be a wrong FLOP count)
            entry["checks"] = {"step_frac_le_best_kernel_frac": False}
        if e.get("roofline_step"):
            entry["hbm_bytes_per_step"] = e["roofline_step"]["bytes_per_step"]
        if "interleave" in entry["workload"]:      # the one config whose step is attention-heavy
            entry["roofline_attention"] = {"kernel": _short_kernel(ra.get("kernel")), "frac": ra.get("frac"), "executed_frac": ra.get("executed_frac")}
        c["configs"].append(entry)
    c["detail"] = out.get("detail_file")
    return c


def fit_line(out, limit=4000):
    """the compact line as a string, never longer than `limit`: what does not fit is shed in a fixed order (it all stays in the detail file) —
    per-config extras first, then the secondary HBM kernel list, the CPU sample text, the read / write traffic split"""
    c = compact_line(out)
    sheds = [lambda: [e.pop("roofline_attention", None) for e in c["configs"]],
             lambda: [e.pop("prompt_row_cache", None) for e in c["configs"]],
             lambda: c.get("cpu_baseline") and c["cpu_baseline"].pop("sample", None),
             lambda: c.pop("hbm_kernels", None),
             lambda: [o.pop(k, None) for o in (c.get("roofline"), c.get("roofline_hbm"), c.get("roofline_attention")) if o for k in ("traffic_read", "traffic_write", "traffic_src", "launches")],
             lambda: [e.pop(k, None) for e in c["configs"] for k in ("steps", "warmup")],
             lambda: [e["roofline"].pop("kernel", None) for e in c["configs"]]]
    line = json.dumps(c, separators=(",", ":"))
    for shed in sheds:
        if len(line) < limit:
            break
        shed()
        line = json.dumps(c, separators=(",", ":"))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="gpt2s_B32_L1024_C12", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="mixed", choices=["mixed", "bf16"], help="the reference's setup.dtype: mixed (default: fp32 masters and residual stream, bf16 "
                    "operands) or bf16 (parameters, inputs and the residual stream in bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dp-plumbing", action="store_true", help="N = 1 only: run the step with the DP machinery live in a one-rank RCCL group")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC table only (no rocprofv3 child passes)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the Llama-2-7B summaries that the default run attaches as configs[]")
    ap.add_argument("--extra-configs-dp", action="store_true", help="attach the extra configs under --gpus N > 1 as well (default: N = 1 only)")
    ap.add_argument("--full-detail", action="store_true", help="also: CPU baselines of the extra configs (extrapolated Llama figures, BASELINE.json configs[0]), "
                                                               "trainer-loop and predict legs of every config (the default run keeps to ~1 min)")
    ap.add_argument("--full-backward", action="store_true", help="also compute the (unused) prompt-row input gradients")
    ap.add_argument("--replicate-mapping", action="store_true", help="DP: keep the mapping layer replicated (all-reduce its gradient)")
    ap.add_argument("--replicate-optimizer", action="store_true", help="DP: no row-sharded optimiser step for the big tensors (all-reduce + replicated Adam)")
    ap.add_argument("--single-dp-mode", action="store_true", help="N > 1: time only the selected DP mode (default: the plain all-reduce mode first, then the sharded one)")
    ap.add_argument("--no-llm-dropout", action="store_true", help="GPT-2: switch the frozen LLM's train-mode dropouts (0.1) off")
    ap.add_argument("--optimizer-overlap", action="store_true", help="update the tail's parameters on a side stream under the next step (measured flat; off by default)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of the HIP multi-tensor Adam")
    ap.add_argument("--detail-file", default=None, help="where the full record goes (default: gpurun_out/bench_detail[_<workload>].json under the repo)")
    args = ap.parse_args()
    t_start = time.perf_counter()

    def stage(msg):
        if os.environ.get("RANK", "0") == "0":       # (progress notes: rank 0 only, stderr only)
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    from med_ts_llm_amd import parallel
    rank, world, local_rank = parallel.init_from_env("cuda")
    if args.dp_plumbing and world == 1 and torch.cuda.is_available():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        parallel.settle_backend_output()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())   # (ranks share a GPU only over gloo on a box with fewer GPUs)
    torch.cuda.set_device(device)
    ctx = (rank, world, device)

    # DP mode: the RCCL-native collectives (reduce-scatter AVG, in-place all-gather, row-sharded mapping layer) are checked against known
    # answers on small tensors first; any error or wrong result falls back to the plain mode (one bucketed all-reduce of replicated
    # gradients, replicated Adam) — printed on the line as dp_mode
    dp_mode, rccl_ranks = None, None
    if world > 1:
        ok, why = parallel.preflight_collectives(device)
        rccl_ranks = parallel.count_ranks(device)
        if not ok and not (args.replicate_mapping and args.replicate_optimizer):
            args.replicate_mapping = args.replicate_optimizer = True
            dp_mode = f"plain all-reduce (fallback: {why})"
        elif args.replicate_mapping and args.replicate_optimizer:
            dp_mode = "plain all-reduce (requested)"
        else:
            dp_mode = ("plain all-reduce" if not args.single_dp_mode else
                       "sharded: row-sharded mapping layer + reduce-scatter / owned-row Adam / bf16 all-gather for tensors >= 2^24 + bucketed all-reduce")
        stage(f"DP pre-flight: {'ok' if ok else why}; {rccl_ranks} ranks answered; mode = {dp_mode}")

    full = args.full_detail
    # N > 1: BOTH DP modes in one run. The headline `value` is the PLAIN mode — north_star's split and the trainer's default since round 6: replicated
    # trainables, ONE bucketed all-reduce of their gradients, replicated Adam. The row-sharded mode (setup.shard_mapping / shard_optimizer) is timed
    # second, under a watchdog: if it raises or exceeds its time budget (a hung collective cannot be caught) the line is printed anyway, with
    # `dp_sharded_failed: true` — machine-readable, so a scaling record can never carry the wrong mode's figure silently.
    sharded_wanted = world > 1 and not args.single_dp_mode and not (args.replicate_mapping and args.replicate_optimizer)
    if sharded_wanted:
        import copy
        a_plain = copy.copy(args)
        a_plain.replicate_mapping = a_plain.replicate_optimizer = True
    else:
        a_plain = args
    stage(f"start {args.workload}" + (" [dp mode: plain all-reduce]" if world > 1 else ""))
    t_leg = time.perf_counter()
    out = run_workload(args.workload, a_plain, ctx, args.steps, args.warmup, not args.no_cpu_baseline, not args.no_roofline, legs=True, stage=stage)
    t_leg = time.perf_counter() - t_leg
    if sharded_wanted:
        budget = max(180.0, 15.0 * t_leg)

        def finish_without_sharded(why):
            if rank == 0:
                out["rccl_ranks"], out["dp_mode"] = rccl_ranks, f"plain all-reduce ({why})"
                out["dp_modes"] = {"plain_allreduce": out["value"], "sharded": None}
                out["dp_sharded_failed"] = True
                print(json.dumps(compact_line(out), separators=(",", ":")), flush=True)
            os._exit(0)        # (every rank: the line above carries the failure; a non-zero worker would make the launcher kill rank 0 before it prints)
        import threading
        watchdog = threading.Timer(budget, finish_without_sharded, args=(f"the sharded leg did not finish within {budget:.0f} s: its figure is missing",))
        watchdog.daemon = True
        watchdog.start()
        stage(f"start {args.workload} [dp mode: sharded]")
        try:
            sh = run_workload(args.workload, args, ctx, args.steps, args.warmup, False, False, legs=False, stage=stage)
        except Exception as e:      # noqa: BLE001
            if rank != 0:            # rank 0 may be blocked in a collective this rank just left: its watchdog prints the line; wait for our own
                time.sleep(budget + 30.0)
                os._exit(0)
            stage(f"sharded leg FAILED: {type(e).__name__}: {e}")
            finish_without_sharded(f"the sharded leg raised {type(e).__name__}: {str(e)[:80]}")
        watchdog.cancel()
        if rank == 0:
            out["dp_modes"] = {"plain_allreduce": out["value"], "sharded": sh["value"],
                               "what": "whole-job samples/s of the same step in both DP modes, measured back to back in this run; `value` is the plain (default) one"}
            out["dp_sharded_failed"] = False
    # Extra configs ride on the N = 1 run only: at N > 1 the line is the scaling record of the headline workload — every additional model under DP
    # is more first-contact RCCL surface that could take the headline down with it (--extra-configs-dp asks for them anyway). Each extra is
    # fenced: a failure there is reported in the line's place, never instead of the headline.
    if args.workload == "gpt2s_B32_L1024_C12" and not args.no_extra_configs and (world == 1 or args.extra_configs_dp):
        extras = [
            # the metric workload under setup.dtype = "bf16" (native bf16 residual stream, round 6): never the headline, reported next to it
            ("gpt2s_B32_L1024_C12@bf16", 10, 3, "samples/sec (1024-step, 12-ch windows) through MedTsLLM fwd+bwd, setup.dtype = bf16"),
            # BASELINE.json configs[1] (ETTh1-shaped forecasting on GPT-2-small; configs[0] is the same case on the CPU: cpu_baseline.configs0_etth1_value)
            ("gpt2s_etth1_B32_L512_C7", 5, 2, "samples/sec ([B, 512, 7] ETTh1-shaped windows, GPT-2-small) through MedTsLLM fwd+bwd"),
            # BASELINE.json configs[2] (LUDB-shaped semantic segmentation on a frozen Llama-2-7B): the configuration where the backbone GEMMs are
            # large enough for the >= 40 % MFMA target
            ("llama2_7b_semseg_B32_L1024_C12", 3, 2, "samples/sec ([B, 1024, 12] windows, Llama-2-7B frozen backbone) through MedTsLLM fwd+bwd"),
            # BASELINE.json configs[3] (PSM anomaly detection: 25 channels, L = 2048, Llama-2-7B) and configs[4] (Llama-3-8B, reconstruction), 1 GPU each
            ("llama2_7b_psm_B32_L2048_C25", 3, 2, "samples/sec ([B, 2048, 25] PSM-shaped windows, Llama-2-7B frozen backbone) through MedTsLLM fwd+bwd"),
            ("llama3_8b_recon_B32_L1024_C12", 3, 2, "samples/sec ([B, 1024, 12] windows, Llama-3-8B frozen backbone, 100 000 trainable vocabulary rows) through MedTsLLM fwd+bwd"),
            # the same backbone with interleave covariates -> T = 1664 per sample (flash attention regime)
            ("llama2_7b_semseg_interleave_B16_L1024_C12", 3, 2,
             "samples/sec ([B, 1024, 12] windows, interleave covariates: T = 1664, Llama-2-7B frozen backbone) through MedTsLLM fwd+bwd"),
        ]
        for wl, n_steps, n_warm, metric in extras:
            stage(f"start {wl}")
            a_x = args
            if wl.endswith("@bf16"):
                import copy
                a_x, wl = copy.copy(args), wl[:-len("@bf16")]
                a_x.dtype = "bf16"
            try:
                extra = run_workload(wl, a_x, ctx, steps=n_steps, warmup=n_warm, want_cpu=full and not args.no_cpu_baseline,
                                     want_roofline=not args.no_roofline, legs=full, stage=stage)
            except Exception as e:      # noqa: BLE001 — the headline is already measured: keep it
                if world > 1:
                    raise               # (a rank that stops taking part in collectives would hang the others: fail the run loudly instead)
                stage(f"{wl} FAILED: {type(e).__name__}: {e}")
                torch.cuda.empty_cache()
                continue
            if rank == 0:
                extra["metric"] = metric
                if a_x is not args and world == 1 and not args.no_live_traffic and not args.no_roofline:
                    # the bf16 configuration's whole-step HBM traffic next to the headline's roofline_step (what the bf16 residual stream saves)
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    stage(f"live PMC passes ({wl} --dtype bf16)")
                    _, note_x, step_x = live_pmc_traffic(wl, [], extra_args=("--dtype", "bf16"))
                    extra["roofline_step"] = roofline_step(step_x, extra, note_x)
                out.setdefault("configs", []).append(extra)
    if rank == 0 and out and out.get("cpu_baseline"):
        # extra configs without a measured CPU leg (--full-detail measures them): the headline's MEASURED CPU figure scaled by algorithmic FLOPs per
        # sample — flagged extrapolated; a ratio for orientation, never a target (a large GPU/CPU ratio says nothing about kernel quality)
        head = out["cpu_baseline"]
        f_head = out["algorithmic_tflop_per_step_per_gpu"] / out["config"]["global_batch"] * out["n_gpus"]
        for e in out.get("configs", []):
            if not e.get("cpu_baseline"):
                f_e = e["algorithmic_tflop_per_step_per_gpu"] / e["config"]["global_batch"] * e["n_gpus"]
                e["cpu_baseline"] = {"value": round(head["value"] * f_head / f_e, 4), "unit": head["unit"], "cores": head["cores"], "kind": head["kind"],
                                     "extrapolated": True, "sample": "not run: the headline workload's measured CPU samples/s x (its algorithmic FLOPs per sample / this config's)"}
    if rank == 0 and out:
        out["rccl_ranks"], out["dp_mode"] = rccl_ranks, dp_mode
        committed = f"committed table profiles/pmc_traffic_<workload>.json (separate rocprofv3 --pmc passes of the same kernel sources, csrc_sha16 {csrc_sha16()})"
        for line in [out] + out.get("configs", []):
            for key in ("roofline", "roofline_hbm", "roofline_attention"):
                if line.get(key):
                    line[key]["traffic_source"] = committed if line[key]["traffic"] is not None else None
        if world == 1 and not args.no_live_traffic and not args.no_roofline:
            # the headline line's traffic is MEASURED by this run (everything above is finished and its memory released)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            stage("live PMC passes")
            objs = [out[k] for k in ("roofline", "roofline_hbm", "roofline_attention") if out.get(k)]
            table, note, step_traffic = live_pmc_traffic(args.workload, [o["kernel"] for o in objs])
            for o in objs:
                if o["kernel"] in table:
                    o["traffic"], o["traffic_source"] = table[o["kernel"]], note
                elif o["traffic"] is not None:
                    o["traffic_source"] = f"{committed}; live pass unavailable: {note}"
            out["roofline_step"] = roofline_step(step_traffic, out, note)
    # the ONE JSON line is the LAST thing on stdout: the process group goes down first (on every rank), the C runtime's buffers — librccl
    # prints through them — are flushed, and rank 0 gives the other ranks a moment to do the same before it prints
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    parallel.flush_c_stdio()
    if rank == 0 and out:
        if world > 1:
            time.sleep(1.0)
        # the full record (kernel instances, prose, trainer-loop / predict legs, every config's own line) goes to a side file; stdout gets
        # ONE compact line the driver can parse (r03's 20 KB line could not be)
        path = args.detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail" + ("" if args.workload == "gpt2s_B32_L1024_C12" else "_" + args.workload)
                                                + (f"_n{world}" if world > 1 else "") + ".json")
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            out["detail_file"] = os.path.relpath(path, ROOT)
        except OSError as e:
            out["detail_file"] = None
            print(f"[bench] detail file not written: {e}", file=sys.stderr)
        line = fit_line(out)
        stage(f"done; line = {len(line)} bytes")
        print(line, flush=True)


if __name__ == "__main__":
    main()
