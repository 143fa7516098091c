"""Data parallelism for the trainable front/back end (the reference has none — SURVEY.md §2a; this is ADDED).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm; "gloo" on CPU for tests).
The frozen backbone is replicated; samples shard across ranks (nothing in the path couples samples — SURVEY §8e),
so the only exchange step is ONE all-reduce (sum, then / world) of a single flat fp32 buffer holding every
trainable gradient. xGMI is point-to-point (7 links x ~153 GB/s per GPU): one large flat collective lets RCCL
drive all links, instead of dozens of small per-parameter ring steps.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """Initialise the default process group from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world_size, local_rank). No-op (0, 1, 0) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if device_type == "cuda":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl" if device_type == "cuda" else "gloo", rank=rank, world_size=world)
    return rank, world, local_rank


class FlatGradAllReduce:
    """Packs the gradients of `params` into one flat fp32 buffer, all-reduces it once, averages and unpacks.

    The per-rank loss is a mean over the LOCAL batch, so averaging the summed gradients over ranks reproduces the
    single-process gradient of the mean over the GLOBAL batch (equal shard sizes)."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @torch.no_grad()
    def __call__(self):
        if self.world <= 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def shard_batch(batch, rank, world):
    """Contiguous equal shard of every tensor / list in a batch dict along dim 0 (bench + tests)."""
    if world <= 1:
        return batch
    out = {}
    for k, v in batch.items():
        n = len(v)
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out
