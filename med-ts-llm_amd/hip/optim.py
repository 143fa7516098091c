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

    def register_shadow(self, shadow):
        self._shadows[id(shadow.param)] = shadow

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
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise N.MtlError("HipAdam needs contiguous fp32 parameters on the GPU (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                by_step.setdefault(st["step"], []).append(p)
            for step, ps in by_step.items():
                arr = (N.AdamTensor * len(ps))()
                keep = []
                for i, p in enumerate(ps):
                    st = self.state[p]
                    g = p.grad
                    if g.dtype != torch.float32 or not g.is_contiguous():
                        g = g.float().contiguous()
                    keep.append(g)
                    sh = self._shadows.get(id(p))
                    arr[i].p, arr[i].g, arr[i].m, arr[i].v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    arr[i].n = p.numel()
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
        return loss
