"""probe: can two ranks share ONE GPU with the gloo backend on device tensors (to test DP code paths on a 1-GPU box)?"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
x = torch.full((1024,), float(rank + 1), device="cuda")
try:
    dist.all_reduce(x)
    print(rank, "all_reduce cuda ok", x[0].item())
    out = [torch.empty(4, device="cuda") for _ in range(world)]
    dist.all_gather(out, torch.full((4,), float(rank), device="cuda"))
    print(rank, "all_gather ok", [o[0].item() for o in out])
    y = torch.full((8,), float(rank), device="cuda")
    dist.broadcast(y, 0)
    print(rank, "broadcast ok", y[0].item())
except Exception as e:
    print(rank, "FAILED", repr(e))
try:
    dist.destroy_process_group()
    dist.init_process_group("nccl", rank=rank, world_size=world)
    z = torch.ones(4, device="cuda")
    dist.all_reduce(z)
    torch.cuda.synchronize()
    print(rank, "nccl same-device ok", z[0].item())
except Exception as e:
    print(rank, "nccl same-device FAILED", repr(e)[:200])
