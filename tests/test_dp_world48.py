"""Data parallelism beyond two ranks, on CPU over gloo (no multi-GPU box is available to the build): world sizes 4 and 8,
the flat bucketed all-reduce, the row-sharded mapping layer and the row-sharded optimiser step (parallel.ShardedUpdate) together — every rank
ends each step with the parameters of a single process that trained on the global batch — plus the trainable-word_embeddings model shape
(Llama-3: shard_mapping_layer steps aside, the sharded optimiser step takes the big tensors) and a checkpoint round trip."""
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


class Toy(torch.nn.Module):
    """the trainable skeleton of the model: a [S, V] mapping weight (rows shardable), a big "vocabulary" table that may be trainable
    (Llama-3) and a wide head — sizes scaled down, structure kept"""

    def __init__(self, S=8, V=24, d=6, n_out=16, trainable_vocab=False):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.mapping = torch.nn.Parameter(torch.randn(S, V, generator=g) * 0.3)
        self.vocab = torch.nn.Parameter(torch.randn(V, d, generator=g) * 0.3, requires_grad=trainable_vocab)
        self.q = torch.nn.Linear(5, d)
        self.head = torch.nn.Linear(S, n_out)
        with torch.no_grad():
            for p in (self.q.weight, self.q.bias, self.head.weight, self.head.bias):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, x, source=None):
        src = self.mapping @ self.vocab if source is None else source          # [S, d] prototypes, batch independent
        return self.head(torch.tanh(self.q(x) @ src.t()))


def _worker(rank, world, port, q, trainable_vocab):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    model, ref = Toy(trainable_vocab=trainable_vocab), Toy(trainable_vocab=trainable_vocab)
    # (1) mapping rows live on one rank each — only when the vocabulary is frozen, like MedTsLLM.shard_mapping_layer
    shard_map = not trainable_vocab
    if shard_map:
        r0, r1 = parallel.shard_range(model.mapping.shape[0], rank, world)
        model.mapping = torch.nn.Parameter(model.mapping.data[r0:r1].clone())
        model.mapping._dp_sharded = True
    # (2) row-sharded optimiser step for every "big" tensor whose rows divide by the world size (threshold lowered to the toy's scale)
    su = parallel.ShardedUpdate(list(model.named_parameters()), rank, world, min_numel=64)
    names = sorted(it["name"] for it in su.items)
    expect = ["head.weight"] + (["mapping", "vocab"] if trainable_vocab else [])
    ok = names == sorted(expect)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(su.optimizer_params(params), lr=0.05)
    ropt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=0.05)
    sync = parallel.FlatGradAllReduce(params, bucket_elems=32)
    ok = ok and sync.flat.numel() == 1 + sum(p.numel() for p in params if not getattr(p, "_dp_sharded", False) and not getattr(p, "_dp_opt_sharded", False))
    per = 2
    for step in range(3):
        g = torch.Generator().manual_seed(50 + step)
        X, Y = torch.randn(per * world, 5, generator=g), torch.randn(per * world, 16, generator=g)
        xs, ys = X[rank * per:(rank + 1) * per], Y[rank * per:(rank + 1) * per]
        src = None
        if shard_map:
            src = parallel.AllGatherRows.apply(model.mapping @ model.vocab, rank, world, None)
        torch.nn.functional.mse_loss(model(xs, src), ys).backward()          # mean over the LOCAL shard
        sync()
        su.sync()
        opt.step()
        su.publish()
        opt.zero_grad()
        torch.nn.functional.mse_loss(ref(X), Y).backward()                   # mean over the GLOBAL batch, one process
        ropt.step()
        ropt.zero_grad()
        full_map = parallel.gather_rows(model.mapping.detach(), world) if shard_map else model.mapping.detach()
        ok = ok and torch.allclose(full_map, ref.mapping, rtol=1e-4, atol=1e-6)
        for (n, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
            if n != "mapping":
                ok = ok and torch.allclose(a, b, rtol=1e-4, atol=1e-6)       # incl. the rows other ranks own (published every step)
    # optimiser state exists for the owned rows only
    for it in su.items:
        st = opt.state[it["shard"]]
        ok = ok and st["exp_avg"].shape[0] == it["p"].shape[0] // world
    # every rank holds the same parameters
    flat = torch.cat([p.detach().flatten() for n, p in model.named_parameters() if n != "mapping" or not shard_map])
    lst = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(lst, flat)
    ok = ok and all(torch.equal(lst[0], t) for t in lst[1:])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("trainable_vocab", [False, True])
def test_dp_world_4_and_8_match_single_process(world, trainable_vocab):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, trainable_vocab)) for r in range(world)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(240)
        assert p_.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(world)) == {r: True for r in range(world)}


def _bf16_worker(rank, world, port, q):
    """publishing the bf16 shadow instead of the fp32 rows: the forward operand is identical on every rank, the fp32 master is current for the
    owned rows only, and owned_rows() lets a checkpoint writer gather the rest"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    torch.manual_seed(0)
    W = torch.nn.Parameter(torch.randn(16, 8))
    shadow = W.detach().to(torch.bfloat16).clone()
    su = parallel.ShardedUpdate([("w", W)], rank, world, min_numel=16)
    mine = su.attach_shadow(W, shadow)
    it = su.items[0]
    ok = su.owned_rows("w") == (it["r0"], it["r1"]) and mine.shape[0] == 16 // world and mine.data_ptr() == shadow[it["r0"]:it["r1"]].data_ptr()
    (W * (rank + 1)).sum().backward()
    su.sync()
    expect_g = torch.full((16 // world, 8), sum(range(1, world + 1)) / world)
    ok = ok and torch.allclose(it["shard"].grad, expect_g) and W.grad is None
    with torch.no_grad():
        it["shard"].add_(it["shard"].grad, alpha=-0.1)             # "the optimiser": owned rows of the master + their bf16 copy
        mine.copy_(it["shard"].to(torch.bfloat16))
    su.publish(async_op=True)
    su.wait_published(W)                                                                   # (what the model does in front of the GEMM that reads it)
    full = parallel.gather_rows(W.detach()[it["r0"]:it["r1"]].contiguous(), world)        # what a checkpoint writer does
    ok = ok and torch.equal(shadow, full.to(torch.bfloat16))                               # every rank's shadow = bf16(every owner's rows)
    stale = torch.ones(16, dtype=torch.bool)
    stale[it["r0"]:it["r1"]] = False
    ok = ok and not torch.equal(W.detach()[stale], full[stale])                            # (non-owned master rows are indeed stale)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bf16_shadow_publishing_world_4():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 4, port, q)) for r in range(4)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(240)
        assert p_.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(4)) == {r: True for r in range(4)}
