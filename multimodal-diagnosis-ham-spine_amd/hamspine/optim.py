"""Fused multi-tensor Adam / AdamW on the HIP kernel hs_adam_step_multi.

Drop-in for torch.optim.Adam / AdamW as the reference uses them (scripts/train.py:257-261,
mibf_net/train_resnet.py:136-139): same hyper-parameters, param_groups (LR schedulers keep working),
state_dict layout (step / exp_avg / exp_avg_sq).  One launch per 32 tensors instead of torch's foreach chain.
"""
import ctypes as C

import torch

from . import _lib as L
from . import rt


import os as _os
def _knock_adam():
    """measurement only (tools/knockout.sh): honoured only by a library built with -DHS_MEASURE, see csrc/blocks.hip knock()"""
    if not int(_os.environ.get("HAMSPINE_KNOCKOUT", "0")) & 32:
        return False
    return bool(L.lib().hs_measure_build())


_KNOCK_ADAM = None


class _FusedAdamBase(torch.optim.Optimizer):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, overlap_backward=None,
                 overlap_chunk=8 << 20):
        """overlap_backward: update parameters while backward is still running -- as soon as `overlap_chunk` elements
        worth of gradients are final (post-accumulate hooks) their fused update is enqueued on a side stream ordered
        behind the streams that produced them; step() flushes the rest and joins.  Same arithmetic as stepping after
        backward; valid whenever nothing between backward and step() reads all gradients at once (no global-norm
        clipping, no gradient accumulation over several backwards), which is how the reference trains
        (scripts/train.py:373-385, mibf_net/train_resnet.py:29-33)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if overlap_backward is None:       # (HAMSPINE_ADAM_OVERLAP=1: A/B measurements of the mode without touching the caller)
            overlap_backward = _os.environ.get("HAMSPINE_ADAM_OVERLAP", "0") == "1"
        self._overlap = bool(overlap_backward)
        self._chunk = int(_os.environ.get("HAMSPINE_ADAM_CHUNK", "0")) << 20 or int(overlap_chunk)
        self._pending, self._pending_n = [], 0
        self._pending_streams = {}      # streams the pending gradients became final on (id -> torch.cuda.Stream)
        self._stream = None
        self._done = set()
        if self._overlap:
            self._group_of = {}
            for g in self.param_groups:
                for p in g["params"]:
                    self._group_of[p] = g
                    p.register_post_accumulate_grad_hook(self._on_grad)

    # -- optimizer-in-backward -------------------------------------------------------------------------------------
    def _on_grad(self, p):
        """post-accumulate hook: runs on the autograd thread with the AccumulateGrad node's stream current; the engine
        has already ordered that stream behind the node that produced the gradient."""
        if p in self._done:
            raise RuntimeError(
                "FusedAdam(overlap_backward=True): a parameter received a second gradient after its update was enqueued "
                "(a second backward() before step(): gradient accumulation) -- step after backward instead")
        s = torch.cuda.current_stream(p.device)
        self._pending_streams[s.cuda_stream] = s
        if not any(q is p for q in self._pending):
            self._pending.append(p)
            self._pending_n += p.numel()
        if self._pending_n >= self._chunk or len(self._pending) >= 96:
            self._flush()

    @torch.no_grad()
    def _flush(self):
        if not self._pending:
            return
        ps, self._pending, self._pending_n = self._pending, [], 0
        dev = ps[0].device
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        # Order the update behind EVERY stream a pending gradient became final on -- not only the stream of the hook that
        # happens to trigger the flush.  (Round-1 bug: the image tower's backward runs first, on the ambient stream; the
        # stem's small gradients stayed pending below the chunk size and were flushed later from a BERT parameter's hook,
        # whose current stream is the text tower's: the update then waited for the tower stream only and could read the
        # stem's weight gradient before its GEMM had finished.)
        streams, self._pending_streams = self._pending_streams, {}
        cur = torch.cuda.current_stream(dev)
        streams[cur.cuda_stream] = cur
        for s in streams.values():
            self._stream.wait_stream(s)
        with torch.cuda.stream(self._stream):
            by_group = {}
            for p in ps:
                by_group.setdefault(id(self._group_of[p]), (self._group_of[p], []))[1].append(p)
            for group, sel in by_group.values():
                self._update(group, sel, 1.0)
        self._done.update(ps)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._overlap and closure is None and grad_scale == 1.0:
            self._flush()
            done, self._done = self._done, set()
            for group in self.param_groups:                 # parameters whose hook never fired this step (none, normally)
                rest = [p for p in group["params"] if p.grad is not None and p not in done]
                if rest:
                    self._update(group, rest, 1.0)
            if self._stream is not None:
                torch.cuda.current_stream(self._stream.device).wait_stream(self._stream)
            return loss
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if ps:
                self._update(group, ps, grad_scale)
        return loss

    def _update(self, group, ps, grad_scale):
        """one fused launch (per 32 tensors) for the parameters `ps` of `group`, on the current stream.

        Host cost matters here (a C2 step has ~400 parameter tensors; the first version spent 1.5 ms per step in this
        function): everything that does not move between steps -- parameter / moment pointer tables, sizes, strides, the
        per-tensor checks -- is built once per parameter set, the step counter is ONE shared 0-dim tensor per chunk
        (state[p]["step"] of every tensor in it), and the gradient pointer table is rebuilt only when a gradient pointer
        changed (the tower executors and hamspine.ddp hand out the same gradient memory every step)."""
        global _KNOCK_ADAM
        if _KNOCK_ADAM is None:
            _KNOCK_ADAM = _knock_adam()
        if not ps or _KNOCK_ADAM:
            return
        lib = L.lib()
        key = (id(group), len(ps), id(ps[0]), id(ps[-1]))
        cache = self.__dict__.setdefault("_tables", {})
        ent = cache.get(key)
        if ent is None or ent["ids"] != [id(p) for p in ps]:
            rt.need_gpu(*ps)
            by_step = {}
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if p.dtype != torch.float32:
                    raise L.HamspineError("FusedAdam expects f32 parameters and gradients")
                by_step.setdefault(int(st["step"]), []).append(p)
            ent = {"ids": [id(p) for p in ps], "chunks": []}
            for step0, sel in by_step.items():      # tensors of a group normally share the step count
                n = len(sel)
                arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
                shared = torch.tensor(float(step0), dtype=torch.float32)
                for p in sel:
                    self.state[p]["step"] = shared                       # one counter object for the whole chunk
                ent["chunks"].append({
                    "sel": sel, "n": n, "p": arr(sel), "m": arr([self.state[p]["exp_avg"] for p in sel]),
                    "v": arr([self.state[p]["exp_avg_sq"] for p in sel]), "cnt": (C.c_int64 * n)(*[p.numel() for p in sel]),
                    "ptrs": [p.data_ptr() for p in sel], "strides": [p.stride() for p in sel], "step": shared,
                    "step_host": int(step0), "gptrs": None, "garr": None, "shadow_epoch": -1, "h": None})
            cache[key] = ent
        b1, b2 = group["betas"]
        for ch in ent["chunks"]:
            sel, n = ch["sel"], ch["n"]
            if [p.data_ptr() for p in sel] != ch["ptrs"]:     # a parameter was re-allocated (.to(), load)
                cache.pop(key, None)
                return self._update(group, ps, grad_scale)
            gptrs = [p.grad.data_ptr() for p in sel]
            if gptrs != ch["gptrs"]:
                for p, st in zip(sel, ch["strides"]):
                    g = p.grad
                    if g.dtype != torch.float32:
                        raise L.HamspineError("FusedAdam expects f32 parameters and gradients")
                    if g.stride() != st:    # rare: a gradient produced outside our nodes in another layout
                        g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                        g2.copy_(g)
                        p.grad = g2
                gptrs = [p.grad.data_ptr() for p in sel]
                ch["gptrs"], ch["garr"] = gptrs, (C.c_void_p * n)(*gptrs)
            # Every tensor of the chunk must still share THIS chunk's counter object.  load_state_dict replaces the counters;
            # so does another table of the same group built for a different set of parameters-with-gradients (an unused
            # parameter, an expert without tokens, set_to_none): it rebinds state[p]["step"] of its own subset, and a check of
            # the first tensor alone would let this table's stale step_host through for the rest.  On a mismatch the table
            # is rebuilt from the per-parameter counters (by_step above), which keeps torch.optim's per-parameter step count.
            st_all = self.state
            if any(st_all[p]["step"] is not ch["step"] for p in sel):
                cache.pop(key, None)
                return self._update(group, ps, grad_scale)
            ch["step_host"] += 1
            ch["step"].fill_(ch["step_host"])                  # 0-dim CPU tensor: no device work
            # bf16 shadows the towers registered for these parameters (hamspine.rt): written by the same kernel, so the next
            # forward needs no cast.  The pointer table follows the registry's epoch.
            if ch["shadow_epoch"] != rt.shadow_epoch():
                hp = [rt.shadow_ptr_of(p) for p in sel]
                ch["h"] = (C.c_void_p * n)(*hp) if any(hp) else None
                ch["shadow_epoch"] = rt.shadow_epoch()
            L.check(lib.hs_adam_step_multi_shadow(
                n, ch["p"], ch["garr"], ch["m"], ch["v"], ch["h"], ch["cnt"], float(group["lr"]), b1, b2, group["eps"],
                group["weight_decay"], ch["step_host"], 1 if self._decoupled else 0, float(grad_scale), rt.stream()),
                "hs_adam_step_multi_shadow")


class FusedAdamW(_FusedAdamBase):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        super().__init__(params, lr, betas, eps, weight_decay, **kw)


class FusedAdam(_FusedAdamBase):
    _decoupled = False


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD (weight decay, momentum with dampening 0, Nesterov) on hs_sgd_step_multi: the reference's fallback
    optimizer (scripts/train.py:309).  Same hyper-parameters, param_groups and state_dict layout (momentum_buffer)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov))

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
            mom = float(group["momentum"])
            # tensors seeing their first momentum step are seeded with the gradient (torch.optim.SGD): partition first
            fresh = [p for p in ps if mom != 0 and "momentum_buffer" not in self.state[p]]
            seen = [p for p in ps if not (mom != 0 and "momentum_buffer" not in self.state[p])]
            for first, sel in ((True, fresh), (False, seen)):
                if not sel:
                    continue
                n = len(sel)
                for p in sel:
                    if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                        raise L.HamspineError("FusedSGD expects f32 parameters and gradients")
                    if p.grad.stride() != p.stride():
                        g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                        g2.copy_(p.grad)
                        p.grad = g2
                    if mom != 0 and first:
                        self.state[p]["momentum_buffer"] = torch.empty_like(p, memory_format=torch.preserve_format)
                arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
                bufs = arr([self.state[p]["momentum_buffer"] for p in sel]) if mom != 0 else None
                cnt = (C.c_int64 * n)(*[p.numel() for p in sel])
                L.check(lib.hs_sgd_step_multi(n, arr(sel), arr([p.grad for p in sel]), bufs, cnt, float(group["lr"]), mom,
                                              float(group["weight_decay"]), 1 if group["nesterov"] else 0, 1 if first else 0,
                                              float(grad_scale), rt.stream()), "hs_sgd_step_multi")
                rt.shadows_stale(sel)      # this kernel does not write the bf16 weight shadows: the next forward re-casts them
        return loss
