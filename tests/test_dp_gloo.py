"""world_size-2 data-parallel tests on CPU (gloo): the flat-gradient all-reduce reproduces the single-process
full-batch gradient, and the sampler/shard helpers partition a batch without overlap."""
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    r, w, _ = parallel.init_from_env("cpu")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    frozen = torch.nn.Linear(3, 3)
    for p_ in frozen.parameters():
        p_.requires_grad = False
    g = torch.Generator().manual_seed(1)
    batch = {"x_enc": torch.randn(8, 6, generator=g), "y": torch.randn(8, 3, generator=g), "descriptions": [f"d{i}" for i in range(8)]}
    shard = parallel.shard_batch(batch, rank, world)
    assert shard["descriptions"] == batch["descriptions"][rank * 4:(rank + 1) * 4]
    loss = torch.nn.functional.mse_loss(frozen(model(shard["x_enc"])), shard["y"])      # mean over the LOCAL shard
    loss.backward()
    sync = parallel.FlatGradAllReduce(list(model.parameters()) + list(frozen.parameters()))
    assert sync.flat.numel() == sum(p_.numel() for p_ in model.parameters()) + 1      # frozen params are not communicated (+ the control slot)
    sync()
    grads = [p_.grad.clone() for p_ in model.parameters()]
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        ref.load_state_dict(model.state_dict())
        torch.nn.functional.mse_loss(frozen(ref(batch["x_enc"])), batch["y"]).backward()  # mean over the GLOBAL batch
        ok = all(torch.allclose(a, b.grad, rtol=1e-5, atol=1e-6) for a, b in zip(grads, ref.parameters()))
        q.put(ok)
    # both ranks end with identical gradients
    t = torch.cat([g_.flatten() for g_ in grads])
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    assert torch.equal(lst[0], lst[1])
    dist.destroy_process_group()


def test_flat_grad_allreduce_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    assert q.get(timeout=5) is True


def test_single_process_is_a_noop():
    from med_ts_llm_amd import parallel
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    assert parallel.init_from_env("cpu") == (0, 1, 0)
    lin = torch.nn.Linear(2, 2)
    lin(torch.ones(1, 2)).sum().backward()
    g0 = lin.weight.grad.clone()
    parallel.FlatGradAllReduce(lin.parameters())()
    assert torch.equal(lin.weight.grad, g0)


# ---- row-sharded mapping layer (parallel.AllGatherRows): CPU semantics
def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    g = torch.Generator().manual_seed(3)
    S, V, d, B = 8, 11, 5, 6
    W, b, E = torch.randn(S, V, generator=g), torch.randn(S, generator=g), torch.randn(V, d, generator=g)
    X, Y = torch.randn(B, d, generator=g), torch.randn(B, S, generator=g)
    r0, r1 = parallel.shard_range(S, rank, world)
    Wl, bl = W[r0:r1].clone().requires_grad_(), b[r0:r1].clone().requires_grad_()
    Wl._dp_sharded = bl._dp_sharded = True
    other = torch.nn.Linear(S, 1)
    torch.manual_seed(0)
    with torch.no_grad():
        other.weight.copy_(torch.linspace(-1, 1, S)[None]), other.bias.zero_()
    src = parallel.AllGatherRows.apply(Wl @ E + bl[:, None], rank, world, None)          # [S, d] prototypes
    xs, ys = X[rank * 3:(rank + 1) * 3], Y[rank * 3:(rank + 1) * 3]
    loss = ((other((xs @ src.t())) - ys[:, :1]) ** 2).mean()                             # mean over the LOCAL batch
    loss.backward()
    sync = parallel.FlatGradAllReduce([Wl, bl, *other.parameters()])
    assert sync.flat.numel() == sum(p_.numel() for p_ in other.parameters()) + 1        # sharded rows are not communicated (+ the control slot)
    sync()
    Wf, bf = W.clone().requires_grad_(), b.clone().requires_grad_()
    ref_other = torch.nn.Linear(S, 1)
    ref_other.load_state_dict(other.state_dict())
    ((ref_other(X @ (Wf @ E + bf[:, None]).t()) - Y[:, :1]) ** 2).mean().backward()       # mean over the GLOBAL batch
    ok = (torch.allclose(Wl.grad, Wf.grad[r0:r1], rtol=1e-5, atol=1e-6) and torch.allclose(bl.grad, bf.grad[r0:r1], rtol=1e-5, atol=1e-6)
          and torch.allclose(other.weight.grad, ref_other.weight.grad, rtol=1e-5, atol=1e-6))
    full = parallel.gather_rows(Wl.detach(), world)
    ok = ok and torch.equal(full, W)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_row_sharded_layer_gets_global_mean_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


# ---- overlapped path: buckets launched from post-accumulate-grad hooks during backward, two consecutive steps
def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Tanh(), torch.nn.Linear(4, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Tanh(), torch.nn.Linear(4, 3))
    ref.load_state_dict(model.state_dict())
    sync = parallel.FlatGradAllReduce(model.parameters(), bucket_elems=16)      # several buckets, hooks armed before backward
    assert len(sync.buckets) >= 3 and sync.buckets[0]["items"][0][0] is list(model.parameters())[-1]   # head first
    opt, ropt = torch.optim.SGD(model.parameters(), lr=0.1), torch.optim.SGD(ref.parameters(), lr=0.1)
    ok = True
    for step in range(2):
        g = torch.Generator().manual_seed(10 + step)
        batch = {"x_enc": torch.randn(8, 6, generator=g), "y": torch.randn(8, 3, generator=g)}
        shard = parallel.shard_batch(batch, rank, world)
        if step == 1 and rank == 1:
            sync.request_flag()                 # what BaseTask.handle_termination does under DP (signal-handler safe)
        torch.nn.functional.mse_loss(model(shard["x_enc"]), shard["y"]).backward()
        launched = sum(b["handle"] is not None for b in sync.buckets)
        ok = ok and launched == len(sync.buckets)                                # every bucket went out DURING backward
        sync()
        # the pre-emption flag rides in the last bucket: raised by ONE rank before step 1's backward, seen by BOTH after that step's sync
        ok = ok and (float(sync.flag_value()) > 0) == (step == 1)
        torch.nn.functional.mse_loss(ref(batch["x_enc"]), batch["y"]).backward()
        ok = ok and all(torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6) for a, b in zip(model.parameters(), ref.parameters()))
        opt.step(), ropt.step()
        opt.zero_grad(), ropt.zero_grad()
    ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(model.parameters(), ref.parameters()))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_overlapped_bucketed_allreduce_two_steps():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


# ---- gradient accumulation: two backwards before one sync() (the hooks have already sent the first micro-batch's buckets)
def _accum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref.load_state_dict(model.state_dict())
    sync = parallel.FlatGradAllReduce(model.parameters(), bucket_elems=16)
    ok = True
    for step in range(2):                       # two optimiser steps of two micro-batches each
        for micro in range(2):
            g = torch.Generator().manual_seed(100 + 10 * step + micro)
            batch = {"x_enc": torch.randn(8, 6, generator=g), "y": torch.randn(8, 3, generator=g)}
            shard = parallel.shard_batch(batch, rank, world)
            torch.nn.functional.mse_loss(model(shard["x_enc"]), shard["y"]).backward()
            torch.nn.functional.mse_loss(ref(batch["x_enc"]), batch["y"]).backward()        # accumulates over the two micro-batches
        sync()
        ok = ok and all(torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6) for a, b in zip(model.parameters(), ref.parameters()))
        model.zero_grad(), ref.zero_grad()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_accumulation_before_sync_is_not_dropped():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


def _forced_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from med_ts_llm_amd import parallel
    dist.init_process_group("gloo", rank=0, world_size=1)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 16, generator=g)

    def run(forced):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 16))
        params = list(model.parameters())
        su = sync = None
        if forced:
            su = parallel.ShardedUpdate(list(model.named_parameters()), 0, 1, min_numel=256, force_collectives=True)
            assert su._live and [it["name"] for it in su.items] == ["2.weight"]
            sync = parallel.FlatGradAllReduce(params, bucket_elems=64, force_collectives=True)
            assert sync._live and sync._hooks and len(sync.buckets) > 1
        else:
            assert not parallel.FlatGradAllReduce(params)._live          # a one-rank group stays silent unless forced
        opt = torch.optim.Adam(su.optimizer_params(params) if su else params, lr=1e-2)
        for i in range(3):
            # (one backward per step: on gloo ShardedUpdate reduces p.grad in place and refuses accumulated gradients by design; the
            #  accumulation path of the reduce-scatter branch is covered on the device, tests/test_gpu_rccl.py)
            torch.nn.functional.mse_loss(model(x), y).backward()
            if sync is not None:
                sync()
                su.sync()
            opt.step()
            if su is not None:
                su.publish()
            opt.zero_grad()
        return [p.detach().clone() for p in model.parameters()]

    try:
        a, b = run(False), run(True)
        ok = all(torch.equal(u, v) for u, v in zip(a, b))
        q.put("ok" if ok else "forced collectives changed the result")
    except Exception as e:      # noqa: BLE001 — the parent prints it
        q.put(repr(e))
        raise
    finally:
        dist.destroy_process_group()


def test_one_rank_group_with_forced_collectives_equals_the_plain_run():
    """force_collectives (how the RCCL calls are exercised on a 1-GPU box, tests/test_gpu_rccl.py) on the CPU backend: every collective of the
    N-rank step runs in a one-rank group — hook-launched buckets, reduce / owned-rows update / publish of ShardedUpdate — and three Adam
    steps end bit-identical to the plain single-process run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=120)
    p.join(60)
    assert msg == "ok" and p.exitcode == 0, msg


def test_sharded_update_marks_the_full_parameter_changed_for_torch_optimizers():
    """the optimiser steps an nn.Parameter VIEW of the owned rows, so the FULL parameter's version counter never moved and every
    persistent bf16 copy of it (LinearFn / MappingTrainableFn shadows trust shadow.version == W._version) stayed on the step-0 weights under
    torch SGD / Adam. publish() now bumps the full parameter unless the optimiser maintains the published shadow itself."""
    from med_ts_llm_amd import parallel
    from med_ts_llm_amd.hip.optim import Bf16Shadow
    p = torch.nn.Parameter(torch.randn(8, 4))
    q = torch.nn.Parameter(torch.randn(8, 4))
    su = parallel.ShardedUpdate([("p", p), ("q", q)], 0, 1, min_numel=1)
    su.attach_shadow(q, torch.zeros(8, 4, dtype=torch.bfloat16))          # q's published copy is optimiser-maintained (HipAdam route)
    sh = Bf16Shadow(p, torch.zeros(8, 4, dtype=torch.bfloat16))
    sh.version = p._version
    assert sh.fresh()
    opt = torch.optim.SGD(su.optimizer_params([p, q]), lr=0.1)
    before, vq = p.detach().clone(), q._version
    p.grad, q.grad = torch.ones_like(p), torch.ones_like(q)
    su.sync()
    opt.step()
    su.publish()
    assert not torch.equal(p.detach(), before)          # the view wrote through to the full tensor ...
    assert not sh.fresh()                                # ... and the full parameter now says so
    assert q._version == vq                              # (a shadow the optimiser keeps current is not invalidated)


def test_preflight_and_native_switch_on_a_plain_backend(monkeypatch):
    from med_ts_llm_amd import parallel
    assert parallel.preflight_collectives(torch.device("cpu")) == (True, "single rank")
    assert parallel.native_collectives() is False        # no process group / gloo: the plain forms
    monkeypatch.setitem(parallel._NATIVE, "enabled", True)
    parallel.disable_native_collectives("test")
    assert parallel._NATIVE["enabled"] is False and parallel._NATIVE["why"] == "test"


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).to(torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 6, generator=g).to(torch.bfloat16), torch.randn(8, 3, generator=g).to(torch.bfloat16)
    sl = slice(rank * 4, (rank + 1) * 4)
    torch.nn.functional.mse_loss(model(x[sl]).float(), y[sl].float()).backward()
    sync = parallel.FlatGradAllReduce(list(model.parameters()))
    assert sync.flat.dtype == torch.bfloat16          # setup.dtype = "bf16": the buckets take the parameters' dtype (p.grad must have it)
    sync()
    assert all(p_.grad.dtype == torch.bfloat16 for p_ in model.parameters())
    grads = torch.cat([p_.grad.float().flatten() for p_ in model.parameters()])
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        ref.load_state_dict({k: v.float() for k, v in model.state_dict().items()})
        torch.nn.functional.mse_loss(ref(x.float()), y.float()).backward()
        want = torch.cat([p_.grad.flatten() for p_ in ref.parameters()])
        q.put(bool((grads - want).norm() / want.norm() < 2e-2))        # bf16 activations, bf16 gradients, a bf16 average of two ranks
    # a mixed-dtype parameter set is refused (one flat buffer, one dtype)
    try:
        parallel.FlatGradAllReduce(list(model.parameters()) + [torch.nn.Parameter(torch.zeros(3))])
        q.put(False)
    except ValueError:
        pass
    dist.destroy_process_group()


def test_flat_grad_allreduce_with_bf16_parameters():
    """setup.dtype = "bf16" under data parallelism (round 6: the restriction is lifted): bf16 buckets, bf16 p.grad, the global-batch gradient"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    assert q.get(timeout=5) is True
    assert q.empty()
