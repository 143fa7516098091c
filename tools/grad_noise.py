"""Noise floor of the full-size trainable gradients across equally valid GEMM tile configurations (GPU box only): the
pruned-vs-full backward tolerance of tests/test_gpu_fullsize.py must sit above it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fullsize as T
from med_ts_llm_amd.hip import _native as N

model = T.model.__wrapped__() if hasattr(T.model, "__wrapped__") else None
if model is None:
    import inspect
    model = inspect.unwrap(T.model)()
lib = N.lib()
from med_ts_llm_amd.hip import ops as ops_mod
x, y = T._x(4), torch.randn(T.B, T.PRED, T.C, generator=torch.Generator().manual_seed(5)).cuda()


def grads(cfg, prune):
    ops_mod._TUNE["gemm"] = (2 if any(cfg) else 0,) + tuple(cfg)      # (reaches the Linear layers' GEMMs; the stack's own launches choose by themselves)
    model.prune_dead_prompt_grads = prune
    try:
        return T._grads(model, x, y)[1]
    finally:
        ops_mod._TUNE["gemm"] = (0, 0, 0, 0, 0)
        model.prune_dead_prompt_grads = True


base = grads((0, 0, 0, 0), False)
for name, cfg, prune in [("auto pruned", (0, 0, 0, 0), True), ("128x128/2/8 full", (128, 128, 2, 8), False), ("128x64/2/4 full", (128, 64, 2, 4), False),
                         ("128x128/2/8 pruned", (128, 128, 2, 8), True), ("256x128/3/16 full", (256, 128, 3, 16), False)]:
    g = grads(cfg, prune)
    errs = {n: float((g[n] - base[n]).norm()) / (float(base[n].norm()) + 1e-12) for n in base}
    worst = max(errs, key=errs.get)
    print(f"{name:22s} vs auto full: mapping_layer.weight {errs['mapping_layer.weight']:.2e}  worst {worst} {errs[worst]:.2e}")
    if name in ("auto pruned", "128x128/2/8 full"):
        for n in errs:
            print(f"      {n:60s} {errs[n]:.2e}")
