"""Which Python lines launch the torch-side copy / fill / cast kernels of a metric-shape training step (GPU box only)."""
import os, sys, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import profile, ProfilerActivity
import test_gpu_fullsize as T
from med_ts_llm_amd.hip.optim import HipAdam

model = inspect.unwrap(T.model)()
params = [p for p in model.parameters() if p.requires_grad]
opt = HipAdam(params, lr=1e-4)
for sh in model.bf16_shadows():
    opt.register_shadow(sh)
x, y = T._x(4), torch.randn(T.B, T.PRED, T.C, generator=torch.Generator().manual_seed(5)).cuda()


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model({"x_enc": x})
        torch.nn.functional.mse_loss(out, y).backward()
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    if not any(k in e.key for k in ("copy", "fill", "zero", "clone", "contiguous", "to", "cat", "mean", "add", "mul", "sub", "index", "sum", "div", "mse")):
        continue
    stack = [s for s in e.stack if "med-ts-llm_amd" in s or "med_ts_llm_amd" in s or "tools/" in s]
    rows.append((dt / 2, e.count / 2, e.key, str(e.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in stack[:3])))
rows.sort(reverse=True)
for dt, cnt, key, shp, st in rows[:45]:
    print(f"{dt:8.1f} us/step  x{cnt:4.1f}  {key:28s} {shp:80s} {st}")
