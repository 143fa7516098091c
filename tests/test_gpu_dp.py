"""Data-parallel path on the device with TWO ranks: one rank per GPU over RCCL ("nccl") when the box has two GPUs; on a 1-GPU box the
two ranks share the device over gloo (RCCL refuses two ranks per GPU; MTL_DIST_BACKEND selects it). The row-sharded mapping layer + flat
gradient all-reduce + HIP Adam reproduce the single-process full-batch step."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
B, L, C, PRED, S = 8, 64, 3, 16, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_env(rank, world, port):
    """torchrun-style environment of one rank; the backend follows the hardware: RCCL with one rank per GPU when there are enough GPUs"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() >= world:
        os.environ.pop("MTL_DIST_BACKEND", None)
        return "nccl"
    os.environ["MTL_DIST_BACKEND"] = "gloo"
    return "gloo"


def _build(shard=None, kind="gpt2"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import hf_cfg, model_config, FakeDataset
    from med_ts_llm_amd.models import model_lookup
    from med_ts_llm_amd.models.backbone import random_state_dict
    from med_ts_llm_amd.utils import dict_to_object
    cfg = hf_cfg("gpt2") if kind == "gpt2" else hf_cfg("llama_gqa", vocab=100_100)      # bigvocab: word_embeddings + mapping both train
    sd = random_state_dict(cfg, seed=7, std=0.06)
    off = {"dataset": False, "task": False, "clip": False, "input_stats": False, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"}
    torch.manual_seed(11)
    model = model_lookup["medtsllm"](dict_to_object(model_config("forecasting", L, PRED, "concat", "linear", off, num_tokens=S)),
                                     FakeDataset(C), backbone_state=(cfg, sd)).to("cuda")
    model.fixed_prompt_ids = torch.randint(0, cfg["vocab_size"], (1, 9), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    model.train()
    if shard is not None:
        assert model.shard_mapping_layer(*shard) is (kind == "gpt2")        # (a trainable vocabulary keeps the mapping layer whole: ShardedUpdate takes it)
    return model


def _batch():
    g = torch.Generator().manual_seed(5)
    return {"x_enc": torch.randn(B, L, C, generator=g).cuda(), "y": torch.randn(B, PRED, C, generator=g).cuda()}


def _step(model, inputs, sync):
    from med_ts_llm_amd.hip.optim import HipAdam
    params = [p for p in model.parameters() if p.requires_grad]
    opt = HipAdam(params, lr=1e-3)
    for sh in model.bf16_shadows():
        opt.register_shadow(sh)
    sync = sync(params) if sync is not None else None      # hooks armed before backward: buckets go out while it runs
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.mse_loss(model(inputs), inputs["y"])
    loss.backward()
    if sync is not None:
        assert all(b["handle"] is not None for b in sync.buckets)
        sync()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    opt.step()
    with torch.autocast("cuda", dtype=torch.bfloat16):                       # second forward uses the Adam-written bf16 shadow
        loss2 = torch.nn.functional.mse_loss(model(inputs), inputs["y"])
    return float(loss.detach()), float(loss2.detach()), grads


def _worker(rank, world, port, q):
    backend = _dist_env(rank, world, port)
    import torch.distributed as dist
    from med_ts_llm_amd import parallel
    assert parallel.init_from_env("cuda")[:2] == (rank, world) and dist.get_backend() == backend
    if backend == "nccl":
        ok, why = parallel.preflight_collectives(torch.device("cuda", torch.cuda.current_device()))
        assert ok and parallel.count_ranks(torch.device("cuda", torch.cuda.current_device())) == world, why
    model = _build(shard=(rank, world))
    assert model.mapping_layer.weight.shape[0] == S // world
    inputs = parallel.shard_batch(_batch(), rank, world)
    loss, loss2, grads = _step(model, inputs, lambda params: parallel.FlatGradAllReduce(params, bucket_elems=20000))
    for k in ("mapping_layer.weight", "mapping_layer.bias"):               # gather the row shards for the comparison
        grads[k] = parallel.gather_rows(grads[k], world)
    sd = model.state_dict()                                                  # collective: gathers the sharded rows
    assert sd["mapping_layer.weight"].shape[0] == S
    losses = [torch.zeros(2, device="cuda") for _ in range(world)]
    dist.all_gather(losses, torch.tensor([loss, loss2], device="cuda"))
    if rank == 0:
        # numpy (pickled by value): torch tensors travel through a Queue as shared-memory handles that die with the sender
        q.put(({k: v.float().cpu().numpy() for k, v in grads.items()}, {k: tuple(v.shape) for k, v in sd.items()},
               torch.stack(losses).mean(0).cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_step_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    grads2, sd2, losses2 = q.get(timeout=240)
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    model = _build()
    loss, loss2, grads1 = _step(model, _batch(), None)
    sd1 = model.state_dict()
    assert abs(float(losses2[0]) - loss) < 2e-3 * abs(loss), (losses2, loss)
    assert abs(float(losses2[1]) - loss2) < 5e-3 * abs(loss2), (losses2, loss2)
    assert loss2 < loss
    params = dict(model.named_parameters())
    worst = {}
    for k, g1 in grads1.items():
        g1, g2 = g1.cpu().float(), torch.from_numpy(grads2[k])
        # analytically-zero gradients (key bias: softmax shift invariance) are compared on an absolute scale
        scale = max(float(g1.norm()), 1e-3 * float(params[k].detach().norm()) + 1e-6)
        worst[k] = float((g1 - g2).norm()) / scale
    print("\n2-rank DP vs single process, per-gradient relative difference:", {k: round(v, 5) for k, v in worst.items()})
    for k, err in worst.items():
        # two half batches vs one full batch: the bf16 roundings fall on different partial sums (a few 1e-3 on the large weights);
        # gradients with few elements that are sums with cancellation (bias vectors) see that noise amplified
        small = grads1[k].numel() < 4096
        assert err < (1e-1 if small else 3e-2), (k, err)
    assert set(sd1) == set(sd2)
    for k in sd1:
        assert tuple(sd1[k].shape) == sd2[k], k


# ---- row-sharded optimiser step on the device: HipAdam on the owned rows, bf16 shadow rows published to the other rank
def _train3(model, batches, world, rank, sharded, optimizer="hipadam", min_numel=4096):
    from med_ts_llm_amd import parallel
    from med_ts_llm_amd.hip.optim import HipAdam, Bf16Shadow
    params = [p for p in model.parameters() if p.requires_grad]
    su = None
    if sharded:
        su = parallel.ShardedUpdate(list(model.named_parameters()), rank, world, min_numel=min_numel)
        assert {it["name"] for it in su.items} >= {"output_projection.linear.weight"}
        model._opt_shards = su
    if optimizer == "sgd":
        # a torch optimiser maintains no bf16 shadow: the forward must notice by itself that the sharded tensors changed (ShardedUpdate.publish
        # bumps the full parameter's version; before that fix every later forward read the step-0 bf16 weights)
        opt = torch.optim.SGD(su.optimizer_params(params) if su else params, lr=0.05, momentum=0.9, nesterov=True)
    else:
        opt = HipAdam(su.optimizer_params(params) if su else params, lr=1e-3)
        for sh in model.bf16_shadows():
            if su is not None and id(sh.param) in su._by_param:
                opt.register_shadow(Bf16Shadow(su._by_param[id(sh.param)]["shard"], su.attach_shadow(sh.param, sh.tensor)))
            else:
                opt.register_shadow(sh)
    sync = parallel.FlatGradAllReduce(params, bucket_elems=20000) if world > 1 else None
    losses = []
    for inputs in batches:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.mse_loss(model(inputs), inputs["y"])
        loss.backward()
        if sync is not None:
            sync()
        if su is not None:
            su.sync()
        opt.step()
        if su is not None:
            su.publish(async_op=True)
        opt.zero_grad()
        losses.append(float(loss.detach()))
    if su is not None:
        su.wait_published()
    return losses, su, opt


def _sharded_worker(rank, world, port, q, kind="gpt2"):
    _dist_env(rank, world, port)
    import torch.distributed as dist
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cuda")
    model = _build(shard=(rank, world), kind=kind)
    full = [_batch()]
    g = torch.Generator().manual_seed(6)
    full.append({"x_enc": torch.randn(B, L, C, generator=g).cuda(), "y": torch.randn(B, PRED, C, generator=g).cuda()})
    full.append(full[0])
    losses, su, opt = _train3(model, [parallel.shard_batch(b, rank, world) for b in full], world, rank, sharded=True)
    head = model.output_projection.linear
    it = su._by_param[id(head.weight)]
    assert opt.state[it["shard"]]["exp_avg"].shape[0] == head.weight.shape[0] // world         # moments for the owned rows only
    sh = model._linear_shadow(head)
    shadows = [torch.zeros_like(sh.tensor) for _ in range(world)]
    dist.all_gather(shadows, sh.tensor)
    assert torch.equal(shadows[0], shadows[1])                                                  # what the forward reads: identical on both ranks
    emb_rows = None
    if kind != "gpt2":
        # trainable vocabulary: both vocabulary-side tensors are row-sharded in the optimiser and PUBLISHED AS bf16 (their shadows travel)
        assert {it["name"] for it in su.items} >= {"word_embeddings", "mapping_layer.weight"}
        for vsh in model._vocab_shadows():
            it_v = su._by_param[id(vsh.param)]
            assert it_v["shadow"] is vsh.tensor and vsh.fresh()
            both = [torch.zeros_like(vsh.tensor) for _ in range(world)]
            dist.all_gather(both, vsh.tensor)
            assert torch.equal(both[0], both[1])
            own = vsh.param.detach()[it_v["r0"]:it_v["r1"]]
            assert torch.equal(vsh.tensor[it_v["r0"]:it_v["r1"], :own.shape[1]], own.to(torch.bfloat16))     # bf16(master) on the owned rows
        it_e = su._by_param[id(model.word_embeddings)]
        emb_rows = parallel.gather_rows(model.word_embeddings.detach()[it_e["r0"]:it_e["r1"]].contiguous(), world)   # every owner's master rows
    sd = model.state_dict()                                                                     # collective: gathers every owner's master rows
    assert torch.equal(sd["output_projection.linear.weight"].to(torch.bfloat16), sh.tensor[:, :head.weight.shape[1]])
    mine = sd["output_projection.linear.weight"][it["r0"]:it["r1"]]
    assert torch.equal(mine, head.weight.detach()[it["r0"]:it["r1"]])
    ls = [torch.zeros(3, device="cuda") for _ in range(world)]
    dist.all_gather(ls, torch.tensor(losses, device="cuda"))
    if rank == 0:
        out = {k: v.float().cpu().numpy() for k, v in sd.items()}
        if emb_rows is not None:
            out["word_embeddings"] = emb_rows.float().cpu().numpy()
        q.put((torch.stack(ls).mean(0).cpu().numpy(), out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["gpt2", "bigvocab"])
def test_two_rank_sharded_optimizer_matches_single_process(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p_ in procs:
        p_.start()
    losses2, sd2 = q.get(timeout=240)
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    model = _build(kind=kind)
    full = [_batch()]
    g = torch.Generator().manual_seed(6)
    full.append({"x_enc": torch.randn(B, L, C, generator=g).cuda(), "y": torch.randn(B, PRED, C, generator=g).cuda()})
    full.append(full[0])
    losses1, _, _ = _train3(model, full, 1, 0, sharded=False)
    for a, b in zip(losses1, losses2):
        assert abs(a - float(b)) < 5e-3 * abs(a), (losses1, losses2)
    assert losses1[2] < losses1[0]
    sd1 = model.state_dict()
    keys = ["output_projection.linear.weight", "mapping_layer.weight", "reprogramming_layer.out_projection.weight"]
    if kind != "gpt2":
        sd1["word_embeddings"] = model.word_embeddings.detach()
        keys.append("word_embeddings")
    for k in keys:
        w1, w2 = sd1[k].float().cpu(), torch.from_numpy(sd2[k])
        moved = 3e-3 * (w1.numel() ** 0.5)       # three Adam steps of lr 1e-3 move every element by <= 3e-3
        assert float((w1 - w2).norm()) < 0.1 * moved, k      # the two runs agree to a small fraction of the distance the weights moved


# ---- a torch optimiser on the owned-rows views never moves the FULL parameter's version counter
def _sgd_worker(rank, world, port, q):
    _dist_env(rank, world, port)
    import torch.distributed as dist
    from med_ts_llm_amd import parallel
    parallel.init_from_env("cuda")
    model = _build(shard=(rank, world))
    full = [_batch()] * 4
    # threshold low enough that the projections and the down-sample layer are sharded too: tensors the model does NOT await one by one
    losses, su, _ = _train3(model, [parallel.shard_batch(b, rank, world) for b in full], world, rank, sharded=True, optimizer="sgd", min_numel=1024)
    names = sorted(it["name"] for it in su.items)
    ls = [torch.zeros(4, device="cuda") for _ in range(world)]
    dist.all_gather(ls, torch.tensor(losses, device="cuda"))
    if rank == 0:
        q.put((torch.stack(ls).mean(0).cpu().numpy(), names))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sgd_forward_sees_updated_weights():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sgd_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    losses2, names = q.get(timeout=240)
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    assert "output_projection.linear.weight" in names and len(names) >= 3, names
    model = _build()
    losses1, _, _ = _train3(model, [_batch()] * 4, 1, 0, sharded=False, optimizer="sgd")
    print("\nSGD, same batch four times: single process", [round(x, 5) for x in losses1], "two ranks (sharded update)", [round(float(x), 5) for x in losses2])
    assert losses1[3] < 0.98 * losses1[0]                      # the single process learns ...
    for a, b in zip(losses1, losses2):                          # ... and the two ranks follow it step by step (stale bf16 weights would freeze the loss)
        assert abs(a - float(b)) < 1e-2 * abs(a), (losses1, losses2)
