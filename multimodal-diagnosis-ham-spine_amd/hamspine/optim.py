"""Fused multi-tensor Adam / AdamW on the HIP kernel hs_adam_step_multi.

Drop-in for torch.optim.Adam / AdamW as the reference uses them (scripts/train.py:257-261,
mibf_net/train_resnet.py:136-139): same hyper-parameters, param_groups (LR schedulers keep working),
state_dict layout (step / exp_avg / exp_avg_sq).  One launch per 32 tensors instead of torch's foreach chain.
"""
import ctypes as C

import torch

from . import _lib as L
from . import rt


class _FusedAdamBase(torch.optim.Optimizer):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            rt.need_gpu(*ps)
            # all tensors of a group share the step count (they are created together)
            steps = set()
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                steps.add(st["step"])
            b1, b2 = group["betas"]
            for step in steps:
                sel = [p for p in ps if self.state[p]["step"] == step]
                n = len(sel)
                for p in sel:
                    if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                        raise L.HamspineError("FusedAdam expects f32 parameters and gradients")
                    # elementwise update: any dense layout works as long as p / grad / state share it
                    if p.grad.stride() != p.stride():
                        g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                        g2.copy_(p.grad)   # rare: a gradient produced outside our nodes in another layout
                        p.grad = g2
                arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
                cnt = (C.c_int64 * n)(*[p.numel() for p in sel])
                L.check(lib.hs_adam_step_multi(
                    n, arr(sel), arr([p.grad for p in sel]), arr([self.state[p]["exp_avg"] for p in sel]),
                    arr([self.state[p]["exp_avg_sq"] for p in sel]), cnt, float(group["lr"]), b1, b2, group["eps"],
                    group["weight_decay"], step, 1 if self._decoupled else 0, float(grad_scale), rt.stream()),
                    "hs_adam_step_multi")
        return loss


class FusedAdamW(_FusedAdamBase):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr, betas, eps, weight_decay)


class FusedAdam(_FusedAdamBase):
    _decoupled = False
