"""Adam / AdamW on the HIP path: torch.optim.Adam semantics (R:tasks/base.py:97,99), one streaming launch for all
trainable tensors (`mtl_adam_step`), optionally emitting the bf16 autocast copy of a weight while it is being updated.

state_dict layout matches torch.optim.Adam ("step", "exp_avg", "exp_avg_sq"), so checkpoints interchange."""
import ctypes as C

import torch

from . import _native as N


class Bf16Shadow:
    """bf16 [rows, ld] copy of an fp32 [rows, cols] parameter, kept current by HipAdam. `version` is the parameter's
    `_version` the shadow corresponds to; owners compare it with `param._version` before trusting the copy."""

    def __init__(self, param, tensor):
        assert param.dim() == 2 and tensor.dim() == 2 and tensor.dtype == torch.bfloat16
        assert tensor.shape[0] == param.shape[0] and tensor.stride(1) == 1 and tensor.stride(0) >= param.shape[1]
        self.param, self.tensor, self.version = param, tensor, -1

    def fresh(self):
        return self.version == self.param._version


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled_weight_decay=decoupled_weight_decay)
        super().__init__(params, defaults)
        self._shadows = {}
        self._late, self._side, self._late_done = set(), None, None

    def register_shadow(self, shadow):
        self._shadows[id(shadow.param)] = shadow

    def defer(self, params):
        """Parameters that the NEXT forward reads only after the frozen backbone (down-sample layer, flatten head): their update runs on a side
        stream, ordered after everything the step has enqueued so far, and overlaps the next step's front end and backbone — HBM-bound Adam traffic
        under MFMA-bound GEMMs (the PSM head: 8 ms of a 226 ms step). Whoever reads such a parameter next must call wait_deferred() first
        (MedTsLLM.predict does, in front of the down-sample GEMM; the trainer before validation / checkpoints). Same arithmetic, same results."""
        self._late = {id(p) for p in params}
        if self._late and self._side is None:
            self._side = torch.cuda.Stream()

    def wait_deferred(self):
        """make the current stream wait for the deferred updates launched by the last step() (no-op when there are none)"""
        if self._late_done is not None:
            torch.cuda.current_stream().wait_event(self._late_done)
            self._late_done = None

    def load_state_dict(self, state_dict):
        """torch's Optimizer.load_state_dict casts every floating-point state tensor to its parameter's dtype; the moments of bf16 parameters
        (setup.dtype = "bf16") must stay fp32 — the kernel reads and writes them as float*."""
        super().load_state_dict(state_dict)
        # (re-widening the narrowed copies would keep their bf16 rounding: take the moments from the saved tensors themselves)
        mine = [p for g in self.param_groups for p in g["params"]]
        saved = [i for g in state_dict["param_groups"] for i in g["params"]]
        for idx, p in zip(saved, mine):
            rec = state_dict["state"].get(idx)
            if not rec:
                continue
            for k in ("exp_avg", "exp_avg_sq"):
                if torch.is_tensor(rec.get(k)):
                    self.state[p][k] = rec[k].detach().to(device=p.device, dtype=torch.float32).contiguous().clone()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) and p.is_contiguous()):
                    raise N.MtlError("HipAdam needs contiguous fp32 (or, with setup.dtype = \"bf16\", bf16) parameters on the GPU (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    # moments are fp32 also for bf16 parameters (setup.dtype = "bf16"): the update is formed in fp32 and the parameter rounded once
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                by_step.setdefault(st["step"], []).append(p)
            batches = []
            for step, ps in by_step.items():
                late = [p for p in ps if id(p) in self._late]
                batches.append((step, [p for p in ps if id(p) not in self._late], False))
                if late:
                    batches.append((step, late, True))
            for step, ps, deferred in batches:
                if not ps:
                    continue
                if deferred:
                    self._side.wait_stream(torch.cuda.current_stream())       # after the backward / all-reduce that produced the gradients
                    with torch.cuda.stream(self._side):
                        self._launch(group, step, ps)
                        self._late_done = torch.cuda.Event()
                        self._late_done.record(self._side)
                    for p in ps:
                        if p.grad is not None:
                            p.grad.record_stream(self._side)                      # zero_grad() may release it while the side stream still reads it
                else:
                    self._launch(group, step, ps)
        return loss

    def _launch(self, group, step, ps):
        """one mtl_adam_step launch (all tensors of `ps`, which share `step`) on the current stream"""
        arr = (N.AdamTensor * len(ps))()
        keep = []
        for i, p in enumerate(ps):
            st = self.state[p]
            for k in ("exp_avg", "exp_avg_sq"):
                m = st[k]
                if not (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.numel() == p.numel()):
                    raise N.MtlError(f"HipAdam: state['{k}'] must be a contiguous fp32 GPU tensor of the parameter's size (got {m.dtype}, {tuple(m.shape)})")
            g = p.grad
            if g.dtype != p.dtype or not g.is_contiguous():
                g = g.to(p.dtype).contiguous()
            keep.append(g)
            sh = self._shadows.get(id(p))
            arr[i].p, arr[i].g, arr[i].m, arr[i].v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            arr[i].n = p.numel()
            arr[i].param_dtype = N.MTL_BF16 if p.dtype == torch.bfloat16 else N.MTL_F32
            if sh is not None:
                arr[i].shadow, arr[i].cols, arr[i].ld_shadow = sh.tensor.data_ptr(), p.shape[1], sh.tensor.stride(0)
            else:
                arr[i].shadow, arr[i].cols, arr[i].ld_shadow = None, 0, 0
        b1, b2 = group["betas"]
        N.check(N.lib().mtl_adam_step(arr, len(ps), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                      float(group["weight_decay"]), 1 if group["decoupled_weight_decay"] else 0,
                                      int(step), N.stream()), "mtl_adam_step")
        for p in ps:   # the kernel wrote through raw pointers: bump the autograd version like an in-place op
            torch.autograd.graph.increment_version(p)
            sh = self._shadows.get(id(p))
            if sh is not None:
                sh.version = p._version
