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


# The RCCL-native forms of the exchanges (reduce_scatter_tensor with ReduceOp.AVG, in-place all_gather_into_tensor, ReduceOp.AVG all-reduce)
# are the default on "nccl"; MTL_DP_NATIVE_COLLECTIVES=0 — or a failed preflight_collectives() — selects the plain forms every backend has
# (all_reduce SUM + divide, all_gather into parts + copy), which is also what gloo runs.
_NATIVE = {"enabled": os.environ.get("MTL_DP_NATIVE_COLLECTIVES", "1") != "0", "why": None}


def native_collectives(group=None):
    return bool(_NATIVE["enabled"]) and dist.is_initialized() and dist.get_backend(group) == "nccl"


def disable_native_collectives(why):
    _NATIVE["enabled"], _NATIVE["why"] = False, why


def count_ranks(device):
    """number of ranks that answer an all-reduce of ones (what a bench line reports as rccl_ranks)"""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def preflight_collectives(device, group=None):
    """Known-answer test of every RCCL-native exchange the DP path uses, on small tensors, BEFORE the first step: the mean all-reduce with the
    control slot, reduce_scatter_tensor(AVG) from a full gradient, the in-place all_gather_into_tensor whose input aliases its own slot of
    the output (fp32 and bf16), and AllGatherRows' pair. Any exception or wrong answer on ANY rank switches every rank to the plain forms
    (disable_native_collectives) — the decision itself travels through a plain SUM all-reduce, so the ranks agree. Returns (ok, reason)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return True, "single rank"
    if not native_collectives(group):
        return True, "plain collectives selected (" + (_NATIVE["why"] or "backend / MTL_DP_NATIVE_COLLECTIVES") + ")"
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    why = None
    try:
        n = 8 * world
        # 1. mean all-reduce: rank r contributes r + 1 everywhere -> (world + 1) / 2
        a = torch.full((n + 1,), float(rank + 1), dtype=torch.float32, device=device)
        dist.all_reduce(a, op=dist.ReduceOp.AVG, group=group)
        if not torch.allclose(a, torch.full_like(a, (world + 1) / 2.0)):
            why = "all_reduce(AVG) wrong"
        # 2. reduce-scatter (mean) of a [world * 8, 4] gradient whose row i holds i * (rank + 1)
        g = (torch.arange(n, dtype=torch.float32, device=device)[:, None] * float(rank + 1)).repeat(1, 4).contiguous()
        own = torch.empty((8, 4), dtype=torch.float32, device=device)
        dist.reduce_scatter_tensor(own, g, op=dist.ReduceOp.AVG, group=group)
        want = torch.arange(rank * 8, rank * 8 + 8, dtype=torch.float32, device=device)[:, None].repeat(1, 4) * (world + 1) / 2.0
        if why is None and not torch.allclose(own, want):
            why = "reduce_scatter_tensor(AVG) wrong"
        # 3. in-place all-gather: every rank owns rows [8 r, 8 r + 8) of `full`, input = that slice of the output
        for dt in (torch.float32, torch.bfloat16):
            full = torch.full((n, 4), -1.0, dtype=dt, device=device)
            full[rank * 8:rank * 8 + 8] = float(rank + 1)
            h = dist.all_gather_into_tensor(full.view(-1), full[rank * 8:rank * 8 + 8].reshape(-1), group=group, async_op=True)
            h.wait()
            want = torch.arange(1, world + 1, dtype=torch.float32, device=device).repeat_interleave(8)[:, None].repeat(1, 4).to(dt)
            if why is None and not torch.equal(full, want):
                why = f"in-place all_gather_into_tensor ({dt}) wrong"
        torch.cuda.synchronize() if device.type == "cuda" else None
    except Exception as e:      # noqa: BLE001 — whatever the backend raises, the answer is the plain mode
        why = f"{type(e).__name__}: {e}"[:120]
    bad = torch.tensor([0.0 if why is None else 1.0], dtype=torch.float32, device=device)
    try:
        dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
        n_bad = int(round(float(bad.item())))
    except Exception as e:      # noqa: BLE001
        n_bad, why = world, why or f"{type(e).__name__}: {e}"[:120]
    if n_bad:
        why = why or f"{n_bad} other rank(s) failed the pre-flight"
        disable_native_collectives(why)
        return False, why
    return True, "native collectives verified"


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
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        # MTL_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses that): how the DP path is tested on a 1-GPU box
        default = "nccl" if device_type == "cuda" else "gloo"
        # more ranks on this node than it has GPUs (the driver's `torch.distributed.run --nproc-per-node N` on a smaller box): RCCL would fail with
        # "Duplicate GPU detected"; the ranks share devices over gloo instead — a functional run, said on stderr and visible as dist_backend
        # Only when the launcher SAYS how many ranks share this node (torch.distributed.run sets LOCAL_WORLD_SIZE): a SLURM / mpirun style
        # launch that exports only RANK / WORLD_SIZE on a 2 x 8-GPU job must keep nccl (the global world size says nothing about this node, and
        # nodes deciding differently would hang in init_process_group) — ADVICE r05.
        local_world = int(os.environ["LOCAL_WORLD_SIZE"]) if "LOCAL_WORLD_SIZE" in os.environ else 0
        if device_type == "cuda" and torch.cuda.is_available() and local_world > torch.cuda.device_count() and "MTL_DIST_BACKEND" not in os.environ:
            default = "gloo"
            if rank == 0:
                import sys
                print(f"[parallel] {local_world} local ranks on {torch.cuda.device_count()} GPU(s): using gloo (RCCL refuses two ranks per device)", file=sys.stderr, flush=True)
        backend = os.environ.get("MTL_DIST_BACKEND", default)
        import datetime
        # (a rank that never arrives must end the job, not hang it: 10 min covers the slowest first RCCL communicator set-up)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=int(os.environ.get("MTL_DIST_TIMEOUT_MIN", "10"))))
        if backend == "nccl":
            settle_backend_output()
    return rank, world, local_rank


def flush_c_stdio():
    """flush the C runtime's stdio buffers of this process (librccl prints through them, Python's sys.stdout does not see that buffer)"""
    import ctypes
    import sys
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass
    sys.stdout.flush()


def settle_backend_output():
    """RCCL creates its communicator at the FIRST collective and prints a version banner to the C stdout then; with stdout on a pipe that
    text sits in the C buffer until the process exits — i.e. it lands AFTER whatever the program printed itself (a result line a caller
    parses from the end of stdout). One barrier right after init creates the communicator, the flush sends the banner out now."""
    if dist.is_initialized():
        dist.barrier()
    flush_c_stdio()


def broadcast_object(obj, src=0, group=None):
    """rank `src`'s picklable object on every rank (control decisions that gate collectives must agree across ranks)"""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def shard_range(n, rank, world):
    """contiguous equal row range [r0, r1) of `n` rows owned by `rank` (n % world == 0)"""
    assert n % world == 0, (n, world)
    per = n // world
    return rank * per, (rank + 1) * per


def gather_rows(t, world, group=None):
    """all ranks: concatenate every rank's [n/world, ...] row shard into the full [n, ...] tensor (a collective)"""
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim=0)


class AllGatherRows(torch.autograd.Function):
    """full[n, d] = concat over ranks of local[n/world, d]; backward hands every rank the GLOBAL-MEAN gradient of ITS rows:
    d_local = (sum over ranks of d_full)[own rows] / world — each rank's loss is a mean over its LOCAL batch, so this
    is the gradient of the mean over the global batch (same convention as FlatGradAllReduce). The exchange is one
    small all-reduce of d_full in fp32 (num_tokens x d_llm: 3 MB for GPT-2), which is what lets the mapping layer's
    [num_tokens, vocab] weight gradient (51.5 M elements, 70 % of the trainable payload) stay OFF the wire."""

    @staticmethod
    def forward(ctx, local, rank, world, group):
        ctx.meta = (rank, world, group, local.shape[0])
        if native_collectives(group):
            # RCCL: ONE collective straight into the full tensor (no per-rank parts, no concatenation) — this exchange sits on the step's
            # critical path, right in front of the reprogramming layer's key / value projections
            local = local.contiguous()
            full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(full, local, group=group)
            return full
        return gather_rows(local, world, group)

    @staticmethod
    def backward(ctx, d_full):
        rank, world, group, n_loc = ctx.meta
        g = d_full.float().contiguous()
        if native_collectives(group):
            # reduce-scatter with the mean formed inside the collective: a rank receives the global-mean gradient of ITS rows and nothing else
            # (half the all-reduce's wire bytes, no slice / divide afterwards)
            own = torch.empty((n_loc,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(own, g, op=dist.ReduceOp.AVG, group=group)
            return own.to(d_full.dtype), None, None, None
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        return (g[rank * n_loc:(rank + 1) * n_loc] / world).to(d_full.dtype), None, None, None


class FlatGradAllReduce:
    """Averages the trainable gradients over the ranks through a few large flat buffers (fp32; bf16 for an all-bf16 model).

    Buckets are filled in REVERSE parameter order (the order backward produces gradients: the output head first, the
    patch/reprogramming front end last) and each bucket's all-reduce is launched asynchronously from a
    post-accumulate-grad hook the moment its last gradient lands, so the head's payload travels while the frozen
    backbone's backward is still computing; `sync()` after backward launches whatever is left, waits, divides by the
    world size and points every p.grad at its slice of the flat buffer (no copy back). xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): few large collectives let RCCL drive all links; `bucket_elems` bounds a bucket.

    The per-rank loss is a mean over the LOCAL batch, so averaging the summed gradients over ranks reproduces the
    single-process gradient of the mean over the GLOBAL batch (equal shard sizes)."""

    def __init__(self, params, group=None, bucket_elems=32 * 1024 * 1024, overlap=True, force_collectives=False, close_at_elems=4 * 1024 * 1024):
        # row-sharded parameters (p._dp_sharded, see AllGatherRows) already hold globally averaged gradients of rows no
        # other rank owns: they are neither communicated nor averaged again
        self.params = [p for p in params if p.requires_grad and not getattr(p, "_dp_sharded", False) and not getattr(p, "_dp_opt_sharded", False)]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collectives: issue every collective in a ONE-rank group too (sums over one rank, / 1: results unchanged) — the way the
        # RCCL calls of this class are exercised against the real backend on a 1-GPU box (tests/test_gpu_rccl.py)
        self._live = self.world > 1 or (force_collectives and dist.is_initialized())
        # RCCL averages inside the collective (ReduceOp.AVG: the sum scaled by 1 / world in its last step) — no separate pass over the flat
        # buffer afterwards; gloo has no AVG: SUM, then one division
        self._avg = native_collectives(group)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        # + one control slot behind the gradients (it travels in the LAST bucket launched): the ranks' pre-emption flag, so that a
        # decision every rank must take together (checkpoint and exit: collectives) costs no collective of its own
        # the buckets take the parameters' dtype: fp32 masters ("mixed" / "fp32"), or bf16 when the whole model is bf16 (setup.dtype = "bf16":
        # p.grad must have p's dtype, and a bf16 model's gradients are bf16 tensors in the reference too). One dtype per model.
        dts = {p.dtype for p in self.params}
        if len(dts) > 1:
            raise ValueError(f"FlatGradAllReduce: trainable parameters of several dtypes {sorted(map(str, dts))} (one flat buffer, one dtype)")
        self.flat = torch.zeros(n + 1, dtype=dts.pop() if dts else torch.float32, device=dev)
        self._flag = False
        self.buckets, self._where = [], {}
        off, cur = 0, None
        for p in reversed(self.params):
            # a bucket is closed as soon as it holds close_at_elems (16 MB: large enough for RCCL to drive every xGMI link) — NOT only when the next
            # tensor would overflow bucket_elems: the metric model's 22 M trainable elements fit ONE 32 M bucket, which would leave at the very end
            # of backward with nothing left to hide under; this way the output head (19 M elements, the first gradients backward produces) travels
            # under the frozen backbone's whole backward and only the front end's 3 M elements go out last
            if cur is None or (cur["n"] > 0 and (cur["n"] >= close_at_elems or cur["n"] + p.numel() > bucket_elems)):
                cur = {"start": off, "n": 0, "items": [], "pending": 0, "handle": None}
                self.buckets.append(cur)
            view = self.flat[off:off + p.numel()].view_as(p)
            cur["items"].append((p, view))
            cur["n"] += p.numel()
            self._where[id(p)] = (cur, view)
            off += p.numel()
        if cur is None:
            cur = {"start": 0, "n": 0, "items": [], "pending": 0, "handle": None}
            self.buckets.append(cur)
        cur["n"] += 1                       # the control slot
        self.views = [self._where[id(p)][1] for p in self.params]
        self._hooks = []
        if overlap and self._live and hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._arm()

    def _arm(self):
        for b in self.buckets:
            b["pending"], b["handle"], b["packed"], b["dirty"] = len(b["items"]), None, set(), False

    def request_flag(self):
        """raise this rank's control flag (safe from a signal handler: a plain attribute write); it is sent with the next step's
        last bucket and flag_value() is > 0 on EVERY rank after that step's sync"""
        self._flag = True

    def flag_value(self):
        """0-dim device tensor: (number of ranks whose flag was up at the last sync) / world"""
        return self.flat[-1]

    def _launch(self, b):
        if b is self.buckets[-1]:
            self.flat[-1:].fill_(1.0 if self._flag else 0.0)
        b["handle"] = dist.all_reduce(self.flat[b["start"]:b["start"] + b["n"]], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM,
                                      group=self.group, async_op=True)

    @torch.no_grad()
    def _on_grad(self, p):
        b, view = self._where[id(p)]
        if b["handle"] is not None or id(p) in b["packed"]:
            # a second backward before sync() (gradient accumulation): p.grad now holds MORE than what was packed / sent.
            # The bucket is repacked from the accumulated p.grad and reduced again in __call__.
            b["dirty"] = True
            return
        b["packed"].add(id(p))
        b["pending"] -= 1
        if b["pending"] == 0:
            # the bucket's last gradient has landed: ONE multi-tensor copy packs all of them (a copy kernel per parameter costs more in launches
            # than in bytes: ~20 tensors of the front end are a few KB each), then the bucket goes out
            torch._foreach_copy_([v for q, v in b["items"]], [q.grad for q, v in b["items"]])
            self._launch(b)

    @torch.no_grad()
    def __call__(self):
        if not self._live:
            return
        for b in self.buckets:           # whatever the hooks did not launch (no hooks, unused parameters, ...)
            if b["dirty"]:               # gradients accumulated after the bucket was packed: the in-flight result is stale
                if b["handle"] is not None:
                    b["handle"].wait()
                b["handle"], b["packed"] = None, set()
            if b["handle"] is None:
                # (a bucket is packed as a whole when its last gradient lands; one that never completed — no hooks, unused parameters,
                # gradient accumulation — is packed here from whatever the parameters hold now)
                have = [(view, p.grad) for p, view in b["items"] if p.grad is not None]
                for p, view in b["items"]:
                    if p.grad is None:
                        view.zero_()
                if have:
                    torch._foreach_copy_([v for v, g in have], [g for v, g in have])
                self._launch(b)
        for b in self.buckets:
            b["handle"].wait()
        if not self._avg:
            self.flat.div_(self.world)
        for p, view in zip(self.params, self.views):
            p.grad = view                # the optimiser reads the averaged gradient straight from the flat buffer
        self._arm()


class ShardedUpdate:
    """Row-sharded optimiser step for the BIG replicated tensors (ZeRO-1 restricted to tensors of >= min_numel elements whose rows
    divide by the world size): the PSM flatten head (51 200 x 32 768 = 1.68 G elements), Llama-3's trainable sub-sampled word
    embeddings (100 000 x 4096) and its mapping weight. Per step and tensor:

      backward        every rank forms the full local gradient (the forward needs the full weight anyway)
      reduce-scatter  rank r receives the sum of rows [r R/N, (r+1) R/N) — launched from a post-accumulate-grad hook, so the head's
                      payload travels under the frozen backbone's backward (gloo has no reduce-scatter: all-reduce + slice, same sums)
      optimiser       on the owned rows only: `shard` is an nn.Parameter VIEW of those rows of p, so any optimiser updates p's storage
                      in place; its moments exist for 1/N of the tensor (the 8.2 ms replicated Adam of the PSM head becomes ~1 ms at N = 8)
      all-gather      of what the forward READS: the bf16 shadow rows (2 B per element on the wire) when the optimiser maintains one
                      (HipAdam + Bf16Shadow), else the fp32 rows themselves

    Wire bytes per element: 4 (reduce-scatter) + 2 (bf16 all-gather) instead of the all-reduce's 4 + 4. With bf16 publishing the fp32
    master rows of OTHER ranks go stale locally; `owned_rows(name)` tells a checkpoint writer what to gather (MedTsLLM.state_dict does).
    Results equal the replicated path's up to the summation order of the collective."""

    def __init__(self, named_params, rank, world, group=None, min_numel=1 << 24, force_collectives=False):
        self.rank, self.world, self.group = rank, world, group
        self._live = world > 1 or (force_collectives and dist.is_initialized())     # (see FlatGradAllReduce: one-rank RCCL exercise)
        self.items, self._by_param = [], {}
        self._rs = native_collectives(group)
        for name, p in named_params:
            if (not p.requires_grad or getattr(p, "_dp_sharded", False) or p.dim() < 2 or p.numel() < min_numel or p.shape[0] % world
                    or not p.is_contiguous()):
                continue
            r0, r1 = shard_range(p.shape[0], rank, world)
            shard = torch.nn.Parameter(p.data[r0:r1])
            p._dp_opt_sharded = True                       # FlatGradAllReduce leaves it alone
            it = {"name": name, "p": p, "shard": shard, "r0": r0, "r1": r1, "handle": None, "dirty": False, "shadow": None,
                  "gshard": torch.empty_like(p.data[r0:r1])}
            self.items.append(it)
            self._by_param[id(p)] = it
            if self._live and hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
                p.register_post_accumulate_grad_hook(lambda q, it=it: self._on_grad(it))

    def optimizer_params(self, params):
        """`params` (list of tensors or of param-group dicts) with every sharded tensor replaced by its owned-rows parameter"""
        def swap(ps):
            return [self._by_param[id(q)]["shard"] if id(q) in self._by_param else q for q in ps]
        if params and isinstance(params[0], dict):
            return [{**g, "params": swap(g["params"])} for g in params]
        return swap(list(params))

    def owned_rows(self, name):
        for it in self.items:
            if it["name"] == name:
                return it["r0"], it["r1"]
        return None

    def attach_shadow(self, param, shadow_tensor):
        """the optimiser keeps shadow_tensor[r0:r1] (bf16) = bf16(updated owned rows): publish() then gathers the shadow instead of the
        master. Returns the owned-rows slice (what the optimiser's shadow hook must write)."""
        it = self._by_param[id(param)]
        it["shadow"] = shadow_tensor
        return shadow_tensor[it["r0"]:it["r1"]]

    @torch.no_grad()
    def _reduce(self, it):
        g = it["p"].grad
        if self._rs:
            it["handle"] = dist.reduce_scatter_tensor(it["gshard"], g.contiguous(), op=dist.ReduceOp.AVG, group=self.group, async_op=True)   # (RCCL: mean inside the collective)
        else:
            it["handle"] = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @torch.no_grad()
    def _on_grad(self, it):
        if it["handle"] is not None:
            it["dirty"] = True                         # a second backward before sync(): reduce again from the accumulated gradient
            return
        self._reduce(it)

    @torch.no_grad()
    def sync(self):
        """after backward: every owned-rows parameter gets the global-mean gradient of its rows; the full local gradient is released"""
        for it in self.items:
            p = it["p"]
            if p.grad is None:
                it["shard"].grad = None
                continue
            if self._live:
                if it["dirty"] and it["handle"] is not None:
                    it["handle"].wait()
                    if not self._rs:
                        raise RuntimeError("gradient accumulation with an all-reduce fallback would sum the first micro-batch twice: "
                                           "call sync() after every backward on this backend")
                    it["handle"] = None
                if it["handle"] is None:
                    self._reduce(it)
                it["handle"].wait()
                if not self._rs:
                    it["gshard"].copy_(p.grad[it["r0"]:it["r1"]])
                    it["gshard"].div_(self.world)
            else:
                it["gshard"].copy_(p.grad[it["r0"]:it["r1"]])
            it["shard"].grad = it["gshard"]
            p.grad = None
            it["handle"], it["dirty"] = None, False

    @torch.no_grad()
    def publish(self, async_op=False):
        """after optimizer.step(): every rank's updated rows reach every rank — as bf16 shadow rows when there is one, else as fp32.
        async_op: the all-gathers are only LAUNCHED (RCCL's stream); wait_published(name) must run before the tensor is read again — the model
        calls it right in front of the first GEMM that reads it, so that e.g. the flatten head's 3.4 GB gather (PSM, bf16) travels under the NEXT
        step's whole frozen-backbone forward instead of in front of it."""
        # The optimiser updated an nn.Parameter VIEW of the owned rows (and the other ranks' rows arrive through p.data below): the FULL
        # parameter's autograd version never moves by itself, and every persistent bf16 copy of it (LinearFn / MappingTrainableFn shadows)
        # trusts `shadow.version == p._version`. Unless the optimiser maintains the copy this class publishes (HipAdam + attach_shadow), the
        # full parameter is marked changed here, so that the next forward re-casts it from the updated master (torch SGD / Adam under DP).
        for it in self.items:
            if it["shadow"] is None:
                torch.autograd.graph.increment_version(it["p"])
        if not self._live:
            return
        for it in self.items:
            full = it["shadow"] if it["shadow"] is not None else it["p"].data
            per = it["r1"] - it["r0"]
            if self._rs and full.is_contiguous():
                h = dist.all_gather_into_tensor(full.view(-1), full[it["r0"]:it["r1"]].reshape(-1), group=self.group, async_op=async_op)
                it["pub"] = h if async_op else None
            else:
                parts = [torch.empty_like(full[:per]) for _ in range(self.world)]
                dist.all_gather(parts, full[it["r0"]:it["r1"]].contiguous(), group=self.group)
                for i, t in enumerate(parts):
                    if i != self.rank:
                        full[i * per:(i + 1) * per].copy_(t)
                it["pub"] = None

    def wait_published(self, param=None, skip=()):
        """block (stream-ordered on a GPU) until the rows published asynchronously have arrived: for `param` (the full tensor) or for all
        (except the tensors in `skip`, which their reader awaits itself later)"""
        for it in self.items:
            if any(it["p"] is q for q in skip):
                continue
            if (param is None or it["p"] is param) and it.get("pub") is not None:
                it["pub"].wait()
                it["pub"] = None


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
