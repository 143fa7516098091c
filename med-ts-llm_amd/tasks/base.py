"""BaseTask-compatible trainer (reference: tasks/base.py). The optimisation step is the reference's loop body
(R:tasks/forecasting.py:19-30) in the same order:

    prepare_batch (H2D + cast) -> autocast(bf16) iff dtype == "mixed" -> model(inputs) -> loss -> backward
    -> [DP: one flat all-reduce of the trainable grads] -> optimizer.step -> zero_grad -> log_step(loss.item())

What drives the hot path is built: device/dtype resolution, loaders, model/optimizer/scheduler/loss construction, fine-tuning from
a pre-trained run (two parameter groups + frozen / warm-up epochs), the checkpoint surface incl. optimiser state, the pre-emption
hook, resume from a run directory. wandb / tensorboard sinks are out of scope.
"""
import json
import os
import signal
import threading
from abc import ABC, abstractmethod
from datetime import datetime
from pathlib import Path

import torch
from torch import optim
from torch.utils.data import DataLoader, default_collate
from torch.utils.data.distributed import DistributedSampler

from ..models import model_lookup
from ..utils import set_seed
from .. import parallel
from .synthetic import get_dataset


def logdir_base(config):
    """R:loggers/base_logger.py:14-17 — config.paths.logdir, else outputs/logs under the working directory"""
    paths = config.get("paths", {}) if hasattr(config, "get") else {}
    base = paths.get("logdir") if hasattr(paths, "get") else None
    return Path(base or "outputs/logs")


class PrintLogger:
    """Minimal sink with the reference logger's surface (log_scores / save_state / log_end); rank 0 only. Writes what
    R:loggers/base_logger.py writes: <logdir>/<run_id>/config.json on a new run and checkpoints/<name>.pt — the reference's
    fields plus `optimizer` (the reference omits the optimiser state, so a resumed run restarts Adam's moments)."""

    def __init__(self, trainer, config, newrun=True):
        self.trainer, self.config = trainer, config
        self.debug = bool(config.get("DEBUG", False))
        self.logdir = logdir_base(config) / trainer.run_id
        self.history = []
        if newrun and not self.debug and trainer.rank == 0:
            self.logdir.mkdir(parents=True, exist_ok=True)
            with open(self.logdir / "config.json", "w") as f:
                json.dump(config.to_dict(), f, indent="\t")

    def log_scores(self, scores):
        self.history.append(dict(scores))
        if self.trainer.rank == 0 and not self.config.setup.get("quiet", False):
            print(" ".join(f"{k}={v:.6g}" if isinstance(v, float) else f"{k}={v}" for k, v in scores.items()))

    def save_state(self, name):
        """Checkpoint format of R:loggers/base_logger.py:29-40 (model.state_dict() is already filtered) + the optimiser state."""
        if self.debug:
            return
        # collectives when the mapping layer is row-sharded (the rows and their Adam moments are gathered): every rank takes part
        if hasattr(self.trainer.optimizer, "wait_deferred"):
            self.trainer.optimizer.wait_deferred()
        if getattr(self.trainer, "opt_shards", None) is not None:
            self.trainer.opt_shards.wait_published()       # asynchronously published rows must have landed before they are read / gathered
        model_state = self.trainer.model.state_dict()
        optim_state = self.trainer.optimizer_state()
        if self.trainer.rank != 0:
            return
        d = self.logdir / "checkpoints"
        d.mkdir(parents=True, exist_ok=True)
        torch.save({"run_id": self.trainer.run_id, "epoch": self.trainer.epoch, "step": self.trainer.step,
                    "datetime": datetime.now().isoformat(), "model": model_state, "optimizer": optim_state,
                    "epochs_done": self.trainer.epochs_done}, d / f"{name}.pt")

    def log_end(self):
        pass


class BaseTask(ABC):
    target_key = "y"

    def __init__(self, run_id, config, newrun=True):
        self.run_id, self.config, self.newrun = run_id, config, newrun
        self.task = config.task
        self.device = self.get_device()
        self.dtype = self.get_dtype()
        if self.dtype == torch.float16:
            # R:tasks/base.py:263-264 casts the whole model AND the inputs to fp16. Not built: the kernels' 16-bit operand type is bf16.
            raise ValueError(f"setup.dtype = {config.setup.dtype!r} (fp16 parameters) is not supported by the MI355X path: use "
                             "\"mixed\" (fp32 masters, bf16 operands — the reference's default), \"bf16\" or \"fp32\"")
        # setup.dtype = "bf16" (R:tasks/base.py:261-262,205-208: the whole model and every floating-point input are cast to bf16, no autocast):
        # NATIVE since round 6 — the trainable parameters are bf16 (checkpoints, optimiser updates, gradients in bf16; Adam's moments stay fp32), the
        # inputs arrive in bf16, the prediction leaves in bf16, and the frozen stack runs on a bf16 RESIDUAL STREAM (FrozenBackbone(stream_dtype):
        # bf16 hidden states saved for the backward, every residual add rounded to bf16, a bf16 gradient stream) — half the stream's HBM bytes of
        # "mixed". Data parallel: the gradient buckets take the parameters' dtype (parallel.FlatGradAllReduce).
        self.rank, self.world_size, self.local_rank = parallel.init_from_env(self.device.type)
        if self.device.type == "cuda" and self.world_size > 1:
            self.device = torch.device("cuda", self.local_rank % torch.cuda.device_count())
        if self.world_size > 1:
            # known-answer test of every RCCL-native exchange BEFORE the objects that use them are built (FlatGradAllReduce / ShardedUpdate read the
            # decision at construction): a failure on any rank switches all ranks to the plain all-reduce / all-gather forms
            self.dp_preflight = parallel.preflight_collectives(self.device)
        set_seed(self.config.setup.seed)
        self.build_datasets()
        self.build_dataloaders()
        self.model = self.build_model().to(self.device, self.dtype)
        self.load_pretrained()
        if self.world_size > 1:     # identical initial weights on every rank (seeded above), rank-specific dropout streams from here on
            torch.manual_seed(self.config.setup.seed + 7919 * (self.rank + 1))
        # DP default since round 6: PLAIN data parallelism (north_star's split: replicated trainables, one bucketed all-reduce of their gradients).
        # The row-sharded forms below are opt-in (setup.shard_mapping / setup.shard_optimizer = true) until a hardware run on >= 2 ranks has passed
        # the pre-flight: every multi-GPU box so far was out of reach (SCALE_r01..r05 skipped), so the default is the exchange every backend has.
        if self.world_size > 1 and self.config.setup.get("shard_mapping", False) and hasattr(self.model, "shard_mapping_layer"):
            self.model.shard_mapping_layer(self.rank, self.world_size)      # DP: rows of the mapping layer live on one rank each
        # DP: the big replicated tensors (flatten head of the wide configs, Llama-3's trainable vocabulary) get a row-sharded optimiser
        # step — reduce-scatter of their gradient, Adam on the owned rows, all-gather of the bf16 copy the forward reads (parallel.ShardedUpdate)
        self.opt_shards = None
        if self.world_size > 1 and self.config.setup.get("shard_optimizer", False):
            su = parallel.ShardedUpdate(list(self.model.named_parameters()), self.rank, self.world_size,
                                        min_numel=int(self.config.setup.get("shard_optimizer_min_numel", 1 << 24)))
            if su.items:
                self.opt_shards = self.model._opt_shards = su
        self.optimizer = self.build_optimizer()
        self.scheduler = self.build_scheduler()
        self.loss_fn = self.build_loss().to(device=self.device)
        self.grad_sync = parallel.FlatGradAllReduce(self.model.parameters()) if self.world_size > 1 else None
        self.epoch, self.step = 1, 0
        self.epochs_done = 0           # completed epochs (checkpointed: a resumed run continues with epoch epochs_done + 1)
        self._stop_requested = False   # set by the SIGUSR1 handler, acted on at a step boundary when several ranks must agree
        metric_dir = self.config.training.eval_metric_direction
        self.best_score = float("inf") if metric_dir == "min" else float("-inf")
        self.logger = PrintLogger(self, self.config, self.newrun)
        if threading.current_thread() is threading.main_thread() and hasattr(signal, "SIGUSR1"):
            signal.signal(signal.SIGUSR1, self.handle_termination)       # pre-emption notice: R:tasks/base.py:55

    # ---- construction (R:tasks/base.py:81-108,157-198,248-275)
    def build_model(self):
        self.model = model_lookup[self.config.model](self.config, self.train_dataset)
        assert self.task in self.model.supported_tasks, f"{self.task} not supported by {self.config.model}"
        return self.model

    def load_pretrained(self):
        """R:tasks/base.py:143-155 — fine-tuning from a pre-trained run: its checkpoint's trainable front / back end minus the output
        head is loaded into the model (MedTsLLM.load_pretrained); `loaded_params` then get their own optimiser group."""
        if "finetuning" not in self.config or not self.config.finetuning.enabled:
            self.finetuning = False
            return
        assert hasattr(self.model, "load_pretrained"), "Only TimeLLM / MedTsLLM support finetuning"      # (R: config.model == "timellm")
        cfg = self.config.finetuning
        self.finetuning = True
        path = logdir_base(self.config) / cfg.pretrained_id / "checkpoints" / f"{cfg.pretrained_ckpt}.pt"
        saved_state = torch.load(path, map_location="cpu")["model"]
        self.loaded_params = self.model.load_pretrained(saved_state)

    def build_optimizer(self):
        if self.finetuning:      # R:tasks/base.py:88-91: group 0 = the new parameters, group 1 = the pre-trained ones (own LR schedule)
            named = list(self.model.named_parameters())
            params = [{"params": [p for n, p in named if n not in self.loaded_params and p.requires_grad]},
                      {"params": [p for n, p in named if n in self.loaded_params]}]
        else:
            params = [p for p in self.model.parameters() if p.requires_grad]
        su = getattr(self, "opt_shards", None)
        if su is not None:
            params = su.optimizer_params(params)          # owned-rows views instead of the full tensors
        lr = self.config.training.learning_rate
        opt = self.config.training.optimizer
        if self.device.type == "cuda" and opt in ("adam", "adamw"):
            # same update rule as torch.optim.Adam/AdamW, one streaming launch for all tensors, and the bf16 autocast
            # copy of the big mapping weight is written by the same kernel (hip/optim.py)
            from ..hip.optim import HipAdam
            o = HipAdam(params, lr=lr, weight_decay=0.01 if opt == "adamw" else 0.0, decoupled_weight_decay=opt == "adamw")
            if self.config.setup.get("overlap_optimizer", False) and hasattr(self.model, "late_parameters") and su is None:
                # opt-in: the tail's parameters are read only after the backbone, their update can run on a side stream under the next step's
                # front end + backbone. Bit-identical runs (tests/test_gpu_model.py), but measured FLAT on 1 GPU (metric step 5.57 vs 5.57-5.66 ms,
                # PSM with its 8 ms head update 219.7 vs 219.7 ms: the update's HBM traffic slows the GEMMs it hides under by what it saves)
                o.defer(self.model.late_parameters())
                self.model.optimizer_wait = o.wait_deferred
            from ..hip.optim import Bf16Shadow
            for sh in getattr(self.model, "bf16_shadows", lambda: [])():
                if su is not None and id(sh.param) in su._by_param:
                    # the optimiser writes the bf16 copy of the rows it owns; ShardedUpdate.publish() gathers the others' (2 B / element)
                    it = su._by_param[id(sh.param)]
                    o.register_shadow(Bf16Shadow(it["shard"], su.attach_shadow(sh.param, sh.tensor)))
                else:
                    o.register_shadow(sh)
            return o
        if opt == "adam":
            return optim.Adam(params, lr=lr)
        if opt == "adamw":
            return optim.AdamW(params, lr=lr, weight_decay=0.01)
        if opt == "sgd":
            return optim.SGD(params, lr=lr, momentum=0.9, nesterov=True)
        raise ValueError(f"Invalid optimizer selection: {opt}")

    def build_scheduler(self):
        st = self.config.training.get("lr_scheduler")
        if st not in (None, "none", "constant"):
            raise ValueError(f"Invalid scheduler selection: {st}")
        if not self.finetuning:
            return optim.lr_scheduler.StepLR(self.optimizer, step_size=1, gamma=1)
        # R:tasks/base.py:118-139: per-epoch LR factor of the PRE-TRAINED group — frozen (0) for the first epochs, or a linear warm-up
        # from warmup_factor to 1; the new parameters always train at the full rate. (The reference's own guard reads
        # `assert not (frozen > 0) and (warmup > 0)`, which by precedence also rejects every frozen-epochs run; the evident intent —
        # the two are mutually exclusive — is what is enforced here.)
        cfg = self.config.finetuning
        frozen, warm = int(cfg.get("frozen_epochs", 0)), int(cfg.get("warmup_epochs", 0))
        assert not (frozen > 0 and warm > 0), "Frozen epochs and warmup epochs are mutually exclusive"
        if frozen > 0:
            return optim.lr_scheduler.LambdaLR(self.optimizer, [lambda _: 1.0, lambda epoch: 0.0 if epoch < frozen else 1.0])
        if warm > 0:
            factors = torch.linspace(float(cfg.warmup_factor), 1.0, warm)
            return optim.lr_scheduler.LambdaLR(self.optimizer, [lambda _: 1.0, lambda epoch: factors[epoch].item() if epoch < warm else 1.0])
        return optim.lr_scheduler.StepLR(self.optimizer, step_size=1, gamma=1)

    # ---- optimiser state in checkpoints, keyed by parameter NAME so that it survives a changed parameter order
    def optimizer_state(self):
        named = {id(p): n for n, p in self.model.named_parameters()}
        su = getattr(self, "opt_shards", None)
        owned = {id(it["shard"]): it for it in su.items} if su is not None else {}
        named.update({k: it["name"] for k, it in owned.items()})
        shard = getattr(self.model, "_map_shard", None)
        out = {"type": type(self.optimizer).__name__, "state": {}, "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.optimizer.param_groups],
               "group_of": {}}
        for gi, g in enumerate(self.optimizer.param_groups):
            for p in g["params"]:
                out["group_of"][named[id(p)]] = gi
        for p, st in self.optimizer.state.items():
            rec = {}
            for k, v in st.items():
                if torch.is_tensor(v) and v.dim() > 0 and shard is not None and getattr(p, "_dp_sharded", False):
                    v = parallel.gather_rows(v, shard[1], shard[4])              # row-sharded mapping layer: full moments, like the weights
                elif torch.is_tensor(v) and v.dim() > 0 and id(p) in owned:
                    v = parallel.gather_rows(v, su.world, su.group)              # row-sharded optimiser step: the moments of every rank's rows
                rec[k] = v.detach().cpu() if torch.is_tensor(v) else v
            out["state"][named[id(p)]] = rec
        return out

    def load_optimizer_state(self, saved):
        if not saved or saved.get("type") != type(self.optimizer).__name__:
            return False
        by_name = dict(self.model.named_parameters())
        su = getattr(self, "opt_shards", None)
        owned_rows = {}
        if su is not None:
            for it in su.items:
                by_name[it["name"]] = it["shard"]
                owned_rows[id(it["shard"])] = (it["r0"], it["r1"])
        shard = getattr(self.model, "_map_shard", None)
        for g, sg in zip(self.optimizer.param_groups, saved["param_groups"]):
            g.update({k: v for k, v in sg.items() if k in g and k != "params"})
        for n, rec in saved["state"].items():
            p = by_name.get(n)
            if p is None:
                continue
            st = {}
            for k, v in rec.items():
                if torch.is_tensor(v) and v.dim() > 0:
                    if shard is not None and getattr(p, "_dp_sharded", False) and v.shape[0] != p.shape[0]:
                        v = v[shard[2]:shard[3]]
                    elif id(p) in owned_rows and v.shape[0] != p.shape[0]:
                        v = v[owned_rows[id(p)][0]:owned_rows[id(p)][1]]
                    # HipAdam keeps fp32 moments also for bf16 parameters (setup.dtype = "bf16"; mtl_adam_step reads m / v as float*): never narrow them
                    fdt = torch.float32 if type(self.optimizer).__name__ == "HipAdam" else p.dtype
                    v = v.to(device=p.device, dtype=fdt if v.is_floating_point() else v.dtype).contiguous().clone()
                st[k] = v
            self.optimizer.state[p] = st
        return True

    def build_datasets(self):
        self.train_dataset = get_dataset(self.config, "train")
        self.val_dataset = get_dataset(self.config, "val")
        self.test_dataset = get_dataset(self.config, "test")

    def build_dataloaders(self):
        nw = self.config.setup.num_workers
        if nw == "auto":
            n_cpu = os.environ.get("SLURM_CPUS_ON_NODE")
            nw = (int(n_cpu) if n_cpu else os.cpu_count()) // 2
        collate = getattr(self.train_dataset, "collate_fn", default_collate)
        bs = self.config.training.batch_size
        shuffle = self.config.training.get("shuffle", True)

        def mk(ds, train):
            sampler = None
            # evaluation is not sharded: window stitching (tasks/evalpath.py) needs every window in order, so each rank
            # scores the full split (identical results on all ranks, no collective in the eval path)
            if self.world_size > 1 and train:
                sampler = DistributedSampler(ds, num_replicas=self.world_size, rank=self.rank, shuffle=train and shuffle,
                                             seed=self.config.setup.seed, drop_last=train)
            return DataLoader(ds, batch_size=bs // self.world_size if (self.world_size > 1 and train) else bs, collate_fn=collate,
                              shuffle=(train and shuffle and sampler is None), sampler=sampler, num_workers=nw,
                              pin_memory=(self.device.type == "cuda"), drop_last=(train and self.world_size > 1))

        self.train_dataloader = mk(self.train_dataset, True)
        self.val_dataloader = mk(self.val_dataset, False)
        self.test_dataloader = mk(self.test_dataset, False)

    def prepare_batch(self, batch):
        """R:tasks/base.py:200-211."""
        if isinstance(batch, dict):
            return {k: self.prepare_batch(v) for k, v in batch.items()}
        if isinstance(batch, (list, tuple)):
            return [self.prepare_batch(x) for x in batch]
        if isinstance(batch, torch.Tensor):
            # pinned source (the loaders pin): an asynchronous copy ordered on the current stream; a blocking one would wait for the GPU
            # queue to drain before every step (the host allocator keeps a pinned block alive until the copies that read it have run)
            batch = batch.to(self.device, non_blocking=batch.is_pinned())
            if batch.dtype.is_floating_point:
                batch = batch.to(self.dtype)
            return batch
        return batch

    def get_device(self):
        d = self.config.setup.device
        if d == "auto":
            return torch.device("cuda" if torch.cuda.is_available() else "cpu")
        return torch.device(d)

    def get_dtype(self):
        self.use_gpu = self.device.type == "cuda"
        name = self.config.setup.dtype
        self.mixed = name == "mixed"
        if name in ("bfloat16", "bf16") and self.use_gpu:
            return torch.bfloat16
        if name in ("float16", "half", "fp16", "16", 16):
            return torch.float16
        if name in ("float32", "float", "fp32", "32", 32, "mixed"):
            return torch.float32
        raise ValueError(f"Invalid dtype selection: {name}")

    # ---- the optimisation step (a10)
    def compute_loss(self, inputs):
        pred = self.model(inputs)
        return self.loss_fn(pred, inputs[self.target_key])

    def train_step(self, inputs):
        inputs = self.prepare_batch(inputs)
        with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=self.mixed):
            loss = self.compute_loss(inputs)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        if self.opt_shards is not None:
            self.opt_shards.sync()
        self.optimizer.step()
        if self.opt_shards is not None:
            # launched only: the model waits for a tensor's rows right before the first kernel that reads it (MedTsLLM._await_rows)
            self.opt_shards.publish(async_op=self.device.type == "cuda")
        self.optimizer.zero_grad()
        self._log_loss(loss)
        return loss

    # ---- per-step loss logging without draining the GPU queue
    # R:tasks/forecasting.py:30 logs `loss.item()` after every step: a blocking D2H copy that waits for the whole step, after which the host
    # starts enqueueing the next step's ~230 launches into an EMPTY queue (the GPU idles for the host's enqueue latency every step). On a
    # GPU the loss (and, under DP, the ranks' agreed pre-emption flag that travelled in the gradient all-reduce) is instead copied to a pinned
    # host slot asynchronously, and each step logs the PREVIOUS step's value once its copy event has fired — the same numbers in the same
    # order, one step later; the epoch's last loss is flushed before validation. setup.deferred_loss_log = false restores the blocking read.
    def _log_loss(self, loss):
        vals = [loss.detach().float().reshape(())]
        if self.grad_sync is not None:
            vals.append(self.grad_sync.flag_value().reshape(()))
        if self.device.type != "cuda" or not self.config.setup.get("deferred_loss_log", True):
            host = torch.stack(vals).tolist()
            self._loss_arrived(host)
            return
        if not hasattr(self, "_loss_ring"):
            self._loss_ring = [(torch.empty(2, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
            self._loss_pending, self._loss_i = [], 0
        slot, ev = self._loss_ring[self._loss_i]
        self._loss_i ^= 1
        slot[:len(vals)].copy_(torch.stack(vals), non_blocking=True)
        ev.record()
        self._loss_pending.append((slot, ev, len(vals)))
        self._flush_losses(keep=1)

    def _flush_losses(self, keep=0):
        pend = getattr(self, "_loss_pending", None)
        while pend and len(pend) > keep:
            slot, ev, n = pend.pop(0)
            ev.synchronize()
            self._loss_arrived(slot[:n].tolist())

    def _loss_arrived(self, host):
        self.log_step(host[0])
        if len(host) > 1 and host[1] > 0 and not getattr(self, "_exiting", False):      # some rank was told to stop: every rank sees the same sum at the same step
            self._checkpoint_and_exit()

    def _agree_on_stop(self):
        """DP: a pre-emption notice that arrived outside the training steps (during validation / the final test, after the last step of the
        run) never rides in a gradient all-reduce. At epoch / validation / test boundaries the ranks therefore agree explicitly — one tiny
        all-reduce of the flags — and checkpoint + exit together (R:tasks/base.py:277-281 checkpoints immediately; a collective cannot)."""
        if self.world_size <= 1 or getattr(self, "_exiting", False):
            return
        t = torch.tensor([1.0 if self._stop_requested else 0.0], dtype=torch.float32, device=self.device)
        torch.distributed.all_reduce(t)
        if float(t.item()) > 0:
            self._checkpoint_and_exit()

    def train(self):
        for epoch in range(self.epochs_done, self.config.training.epochs):     # (a run resumed by from_run_id continues where it stopped)
            if self.rank == 0:
                print(f"Epoch {epoch + 1}/{self.config.training.epochs}")
            if self.world_size > 1:
                self.train_dataloader.sampler.set_epoch(epoch)
            self.model.train()
            for inputs in self.train_dataloader:
                self.train_step(inputs)
            self._flush_losses()
            if hasattr(self.optimizer, "wait_deferred"):
                self.optimizer.wait_deferred()
            if self.opt_shards is not None:
                self.opt_shards.wait_published()
            self._agree_on_stop()
            val_scores = self.val()
            self.epochs_done = epoch + 1
            self.log_epoch(val_scores)
            self.scheduler.step()
            self._agree_on_stop()
        self.model.eval()

    def _eval_loss(self, loader, prefix):
        self.model.eval()
        tot, n = 0.0, 0          # (the running sum stays on the device: ONE D2H read per split instead of one queue drain per batch)
        with torch.no_grad():
            for inputs in loader:
                inputs = self.prepare_batch(inputs)
                with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=self.mixed):
                    loss = self.compute_loss(inputs)
                bs = inputs["x_enc"].shape[0]
                tot, n = tot + loss.detach().double() * bs, n + bs
        scores = {f"{prefix}/{self.config.training.eval_metric}": float(tot) / max(n, 1)}
        self.log_scores(scores)
        return scores

    def val(self):
        return self._eval_loss(self.val_dataloader, "val")

    def test(self):
        scores = self._eval_loss(self.test_dataloader, "test")
        self._agree_on_stop()
        return scores

    def predict(self, dataloader):
        self.model.eval()
        preds = []
        with torch.no_grad():
            for inputs in dataloader:
                preds.append(self.model(self.prepare_batch(inputs)).float().cpu())
        return torch.cat(preds, dim=0)

    @abstractmethod
    def build_loss(self):
        pass

    # ---- logging / checkpoint hooks (R:tasks/base.py:213-246)
    def log_end(self):
        self.logger.log_end()

    def log_step(self, loss):
        self.step += self.config.training.batch_size
        self.logger.log_scores({"train/loss": loss})

    def log_scores(self, scores={}, **kw):
        self.logger.log_scores({**scores, **kw})

    def log_epoch(self, scores={}, **kw):
        lrs = self.scheduler.get_last_lr()
        lrs = {"train/lr": lrs[0]} if len(lrs) == 1 else ({"train/lr": lrs[0], "train/finetune_lr": lrs[1]} if len(lrs) == 2 else {})
        scores = {**scores, **kw, **lrs}
        self.logger.log_scores(scores)
        self.logger.save_state("latest")
        metric = "val/" + self.config.training.eval_metric
        d = self.config.training.eval_metric_direction
        better = metric in scores and ((d == "min" and scores[metric] < self.best_score) or (d == "max" and scores[metric] > self.best_score))
        if self.world_size > 1:
            # save_state is a collective when the mapping layer is row-sharded (state_dict gathers the rows): every rank must take
            # the same branch, so rank 0's comparison decides (each rank scores the validation split itself; 1 ulp would do)
            better = parallel.broadcast_object(bool(better), src=0)
        if better:
            self.best_score = scores[metric]
            if self.config.training.get("save_best", True):
                self.logger.save_state("best")
        if self.epoch < self.config.training.epochs:
            self.epoch += 1

    def handle_termination(self, signum, frame):
        """R:tasks/base.py:277-281 — SIGUSR1 (a scheduler's pre-emption notice): checkpoint "latest", close the logger, exit.
        The reference is single-process and checkpoints from inside the handler. With data parallelism the checkpoint is a COLLECTIVE
        (row-sharded mapping layer: state_dict / optimizer_state gather rows) and the ranks receive the signal at different points of
        their steps, so the handler only raises a flag; the flag rides in the next gradient all-reduce (FlatGradAllReduce's extra
        slot), every rank sees the same sum at the same step boundary and all of them checkpoint and exit together."""
        if self.world_size > 1:
            self._stop_requested = True
            self.grad_sync.request_flag()
            return
        self._checkpoint_and_exit()

    def _checkpoint_and_exit(self):
        if getattr(self, "_exiting", False):       # (_flush_losses below re-enters through _loss_arrived while the agreed flag is still up)
            return
        self._exiting = True
        print("Interrupted!")
        self._flush_losses()
        self.logger.save_state("latest")
        self.log_end()
        raise SystemExit(0)

    @classmethod
    def from_run_id(cls, run_id, cfg=None, ckpt="latest", basepath=None):
        """R:tasks/base.py:283-306: rebuild the trainer of a finished / interrupted run from its log directory — the config the
        run was started with (config.json; config.toml when a TOML reader is importable), overridden by `cfg` (a dict or a config
        object) — and load its checkpoint: model, epoch, step, and (ours) the optimiser state."""
        from ..utils import dict_to_object
        if cfg is not None and hasattr(cfg, "to_dict"):
            cfg = cfg.to_dict()
        base = Path(basepath) if basepath is not None else logdir_base(dict_to_object(cfg or {}))
        rundir = base / run_id
        if (rundir / "config.json").exists():
            with open(rundir / "config.json") as f:
                config = json.load(f)
        elif (rundir / "config.toml").exists():
            import tomli
            with open(rundir / "config.toml", "rb") as f:
                config = tomli.load(f)
        else:
            config = {}
        config = dict_to_object({**config, **(cfg or {})})
        trainer = cls(run_id, config, newrun=False)
        state = torch.load(rundir / f"checkpoints/{ckpt or 'latest'}.pt", map_location="cpu")
        _, unexpected = trainer.model.load_state_dict(state["model"], strict=False)
        assert not unexpected, f"Unexpected keys in model state: {unexpected}"
        trainer.epoch, trainer.step = state["epoch"], state["step"]
        trainer.load_optimizer_state(state.get("optimizer"))
        # (ours) the LR schedule and the epoch counter continue too: without them LambdaLR restarts at epoch 0 and a resumed fine-tuning
        # run re-freezes / re-warms the pre-trained group; checkpoints of the reference's format (no such fields) restart like the reference
        trainer.epochs_done = int(state.get("epochs_done", 0))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # ("lr_scheduler.step() before optimizer.step()": intended, this is a fast-forward)
            for _ in range(trainer.epochs_done):     # StepLR / LambdaLR are pure functions of the epoch count
                trainer.scheduler.step()
        return trainer
