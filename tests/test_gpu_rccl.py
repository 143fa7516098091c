"""The data-parallel classes against the REAL RCCL backend on the one GPU of the test box: a one-rank "nccl" process group with
`force_collectives=True`, so that every collective the N-rank path issues — the bucketed asynchronous flat all-reduce launched from
gradient hooks, the row-sharded mapping layer's all-gather / all-reduce pair, ShardedUpdate's reduce-scatter (the `_rs` branch that gloo
does not have) and its in-place bf16 all-gather published asynchronously — really goes through RCCL (own stream, work handles, in-place
aliasing rules, bf16 payloads) instead of through gloo or stand-ins. With one participant every sum is the identity and the divisor is 1,
so the trained weights must equal the plain single-process run's BIT FOR BIT; what can break is the plumbing (stream ordering between
RCCL's stream and the compute stream, handle waits, buffer aliasing), and that is what the test is for. RCCL refuses two ranks on one
device, so N = 2 on this box stays with gloo (tests/test_gpu_dp.py); a true multi-GPU run is the driver's SCALE job."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _train(model, batches, dp):
    from med_ts_llm_amd import parallel
    from med_ts_llm_amd.hip.optim import HipAdam, Bf16Shadow
    params = [p for p in model.parameters() if p.requires_grad]
    su = sync = None
    if dp:
        su = parallel.ShardedUpdate(list(model.named_parameters()), 0, 1, min_numel=4096, force_collectives=True)
        assert su._rs and su._live and {it["name"] for it in su.items} >= {"output_projection.linear.weight"}
        model._opt_shards = su
    opt = HipAdam(su.optimizer_params(params) if su else params, lr=1e-3)
    for sh in model.bf16_shadows():
        if su is not None and id(sh.param) in su._by_param:
            opt.register_shadow(Bf16Shadow(su._by_param[id(sh.param)]["shard"], su.attach_shadow(sh.param, sh.tensor)))
        else:
            opt.register_shadow(sh)
    if dp:
        sync = parallel.FlatGradAllReduce(params, bucket_elems=20000, force_collectives=True)
        assert sync._live, "collectives not forced"
        assert sync._hooks and len(sync.params) > 3, (len(sync._hooks), len(sync.params), len(sync.buckets))
    losses = []
    for i, inputs in enumerate(batches):
        for micro in range(2 if i == 2 else 1):          # step 2 accumulates two backwards: the "dirty bucket" / second reduce-scatter paths
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = torch.nn.functional.mse_loss(model(inputs), inputs["y"])
            loss.backward()
        if sync is not None:
            if i != 2:
                assert all(b["handle"] is not None for b in sync.buckets)      # every bucket went out from its hook, during backward
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
    torch.cuda.synchronize()
    return losses


def _worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        from test_gpu_dp import _build, _batch, B, L, C, PRED
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        g = torch.Generator().manual_seed(6)
        batches = [_batch(), {"x_enc": torch.randn(B, L, C, generator=g).cuda(), "y": torch.randn(B, PRED, C, generator=g).cuda()}, _batch(), _batch()]

        plain = _build()
        torch.manual_seed(123)                               # the dropout seeds of both runs come from the same host RNG sequence
        losses_plain = _train(plain, batches, dp=False)
        sd_plain = {k: v.clone() for k, v in plain.state_dict().items()}

        model = _build(shard=(0, 1, None, True))             # mapping layer "row-sharded" over the one rank: AllGatherRows runs through RCCL
        assert model._map_shard is not None
        torch.manual_seed(123)
        losses_dp = _train(model, batches, dp=True)
        sd_dp = model.state_dict()                           # (collective: gathers the mapping rows)

        assert losses_plain == losses_dp, (losses_plain, losses_dp)
        assert losses_dp[-1] < losses_dp[0]
        assert set(sd_plain) == set(sd_dp)
        for k in sd_plain:
            assert torch.equal(sd_plain[k], sd_dp[k]), k
        head = model.output_projection.linear
        sh = model._linear_shadow(head)
        assert torch.equal(sh.tensor[:, :head.weight.shape[1]], head.weight.detach().to(torch.bfloat16))     # the published bf16 rows = bf16(master)
        dist.barrier()
        dist.destroy_process_group()
        q.put("ok")
    except BaseException as e:      # noqa: BLE001 — the parent prints it
        import traceback
        q.put("".join(traceback.format_exception(type(e), e, e.__traceback__)))
        raise


def test_dp_path_through_real_rccl_with_one_rank():
    from test_gpu_dp import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=300)
    p.join(120)
    assert msg == "ok", msg
    assert p.exitcode == 0
