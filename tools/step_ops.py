#!/usr/bin/env python3
"""Which host op launches which device kernel in one metric step (torch.profiler, with input shapes): finds the glue — ATen copies,
fills, casts — between the library's kernels. usage: python tools/step_ops.py [workload]  (on the GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "gpt2s_B32_L1024_C12"
    from med_ts_llm_amd.hip.optim import HipAdam
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    hf_cfg, B, L, C_, pred, n_tok, task = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    big = hf_cfg["model_type"] == "llama"
    sd = random_state_dict(hf_cfg, seed=0, std=0.02, device=dev if big else "cpu", dtype=torch.bfloat16 if big else torch.float32)
    torch.manual_seed(0)
    model = model_lookup["medtsllm"](dict_to_object(bench.model_cfg(L, pred, task)), bench.DS(C_, 4 if task == "semantic_segmentation" else 0),
                                     backbone_state=(hf_cfg, sd)).to(dev)
    model.fixed_prompt_ids = torch.randint(0, hf_cfg["vocab_size"], (1, n_tok), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = HipAdam(params, lr=1e-4)
    for sh in model.bf16_shadows():
        opt.register_shadow(sh)
    loss_fn = torch.nn.MSELoss() if task != "semantic_segmentation" else torch.nn.CrossEntropyLoss()
    batch = bench.make_batch(B, L, C_, pred, 1000, dev, task)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(batch)
            loss = loss_fn(out if task != "semantic_segmentation" else out.permute(0, 2, 1), batch["y"])
        loss.backward()
        opt.step()
        opt.zero_grad()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.device_time_total > 0 and not e.cpu_children:
            rows.append((e.device_time_total, e.name, str(e.input_shapes)[:120]))
    rows.sort(reverse=True)
    print(f"ATen leaf ops with device time in ONE step of {wl} (us):")
    agg = {}
    for t, n, s in rows:
        k = (n, s)
        a = agg.setdefault(k, [0.0, 0])
        a[0] += t
        a[1] += 1
    for (n, s), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{t:9.1f} us  x{c:<3d} {n:28s} {s}")
    print("total ATen device time per step: %.1f us" % sum(t for t, _, _ in rows))
    kern = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            a = kern.setdefault(e.name[:90], [0.0, 0])
            a[0] += e.device_time_total
            a[1] += 1
    print("\ndevice kernels in the step: %d launches, %.1f us" % (sum(c for _, c in kern.values()), sum(t for t, _ in kern.values())))
    for n, (t, c) in sorted(kern.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{t:9.1f} us  x{c:<3d} {n}")


    print("\nATen ops by self device time (with shapes and the Python frame that issued them):")
    ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
    rows2 = [k for k in ka if k.key.startswith("aten::") and k.self_device_time_total > 0]
    for k in sorted(rows2, key=lambda k: -k.self_device_time_total)[:45]:
        frames = [f for f in (k.stack or []) if "med-ts-llm_amd" in f or "bench.py" in f or "step_ops" in f]
        print(f"{k.self_device_time_total:9.1f} us  x{k.count:<3d} {k.key:26s} {str(k.input_shapes)[:70]:70s} {frames[0][-70:] if frames else ''}")


if __name__ == "__main__":
    main()
