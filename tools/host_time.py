"""Host (enqueue) time per training step vs GPU time, metric workload (GPU box only): how far ahead of the GPU the Python side runs."""
import os, sys, time, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fullsize as T
from med_ts_llm_amd.hip.optim import HipAdam

T.GPT2_SMALL.update({"embd_pdrop": 0.1, "attn_pdrop": 0.1, "resid_pdrop": 0.1})
model = inspect.unwrap(T.model)()
params = [p for p in model.parameters() if p.requires_grad]
opt = HipAdam(params, lr=1e-4)
for sh in model.bf16_shadows():
    opt.register_shadow(sh)
x, y = T._x(4), torch.randn(T.B, T.PRED, T.C, generator=torch.Generator().manual_seed(5)).cuda()


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model({"x_enc": x})
        loss = torch.nn.functional.mse_loss(out, y)
    loss.backward()
    opt.step()
    opt.zero_grad()


for _ in range(10):
    step()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step (GPU-bound iff enqueue < wall)")
# pure host cost: same steps with the queue drained after each one
ts = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); step(); b = time.perf_counter(); ts.append(b - a)
print(f"host time of one step issued into an empty queue: {1e3 * sorted(ts)[len(ts)//2]:.3f} ms")
