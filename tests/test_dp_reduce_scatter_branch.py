"""parallel.ShardedUpdate's RCCL-only branch (reduce_scatter_tensor / in-place all_gather_into_tensor, gradient accumulation across two
backwards, asynchronous publish + wait_published) run on CPU: gloo has neither collective, so the test substitutes equivalents built from
all_reduce / all_gather with the SAME signatures and argument checks, and forces the branch on. What this pins is the branch's own logic
(buffers, slices, the `dirty` re-reduce, the hand-over of handles); the RCCL calls themselves first run on a multi-GPU node."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Done:
    def wait(self):
        return True


def _reduce_scatter_tensor(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert input.is_contiguous() and output.is_contiguous() and input.numel() == world * output.numel() and input.dtype == output.dtype
    tmp = input.clone()
    dist.all_reduce(tmp, op=op, group=group)
    output.copy_(tmp.view(world, -1)[rank].view_as(output))
    return _Done() if async_op else None


def _all_gather_into_tensor(output, input, group=None, async_op=False):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert output.is_contiguous() and input.is_contiguous() and output.numel() == world * input.numel() and input.dtype == output.dtype
    # the in-place form RCCL accepts: the input is this rank's slot of the output
    assert input.data_ptr() == output.data_ptr() + rank * input.numel() * input.element_size()
    parts = [torch.empty_like(input) for _ in range(world)]
    dist.all_gather(parts, input.clone(), group=group)
    for i, t in enumerate(parts):
        output.view(world, -1)[i].copy_(t.view(-1))
    return _Done() if async_op else None


def _worker(rank, world, port, q, with_shadow):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    dist.reduce_scatter_tensor, dist.all_gather_into_tensor = _reduce_scatter_tensor, _all_gather_into_tensor
    g = torch.Generator().manual_seed(3)
    W0 = torch.randn(8 * world, 12, generator=g)
    lin, ref = torch.nn.Linear(12, 8 * world, bias=False), torch.nn.Linear(12, 8 * world, bias=False)
    with torch.no_grad():
        lin.weight.copy_(W0); ref.weight.copy_(W0)
    su = parallel.ShardedUpdate(list(lin.named_parameters()), rank, world, min_numel=16)
    su._rs = True                                    # the branch a "nccl" group takes
    assert [it["name"] for it in su.items] == ["weight"]
    shadow = own = None
    if with_shadow:                                  # the optimiser writes bf16(updated owned rows) here; publish() gathers THIS, not the masters
        shadow = torch.zeros(8 * world, 12, dtype=torch.bfloat16)
        own = su.attach_shadow(lin.weight, shadow)
    opt = torch.optim.SGD(su.optimizer_params([lin.weight]), lr=0.1)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    r0, r1 = su.owned_rows("weight")
    ok = True
    for step in range(3):
        gs = torch.Generator().manual_seed(100 + step)
        X = torch.randn(2, 2 * world, 12, generator=gs)          # two micro-batches per step (gradient accumulation)
        for mb in range(2):
            (lin(X[mb, 2 * rank:2 * rank + 2]).square().mean() / 2).backward()     # second backward: handle pending -> `dirty` -> re-reduce
        su.sync()
        ok = ok and lin.weight.grad is None and su.items[0]["shard"].grad.shape == (8, 12)
        opt.step()
        if with_shadow:
            with torch.no_grad():
                own.copy_(su.items[0]["shard"].to(torch.bfloat16))
        su.publish(async_op=True)
        ok = ok and su.items[0].get("pub") is not None
        su.wait_published(lin.weight)
        ok = ok and su.items[0].get("pub") is None
        opt.zero_grad()
        if with_shadow:
            # every rank's shadow = bf16 of every rank's master rows; the fp32 rows of the OTHER ranks are stale by design, so the next forward
            # (which in the product reads the shadow) gets them from it
            masters = parallel.gather_rows(lin.weight.data[r0:r1].clone(), world)
            ok = ok and torch.equal(shadow, masters.to(torch.bfloat16))
            with torch.no_grad():
                lin.weight.copy_(masters)
                ref.weight.copy_(masters)
            continue
        for mb in range(2):
            (ref(X[mb]).square().mean() / 2).backward()
        ropt.step(); ropt.zero_grad()
        ok = ok and torch.allclose(lin.weight.data, ref.weight.data, rtol=1e-5, atol=1e-6)       # fp32 publish: the whole tensor follows
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("with_shadow", [False, True])
def test_reduce_scatter_branch_with_accumulation_world2(with_shadow):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, with_shadow)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)], res
