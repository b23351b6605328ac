"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).

Replaces torch.nn.parallel.DistributedDataParallel(find_unused_parameters=True) as used by reference
mibf_net/train_resnet.py:84-90,133-134 for the synchronous-SGD semantics that matter to the step:
  * constructor: parameters and buffers follow rank 0 (broadcast),
  * backward: gradients are averaged over ranks before optimizer.step(); parameters that took no part in
    the forward keep grad None,
  * per-rank BatchNorm statistics (no SyncBN), per-rank batch of 32 (weak scaling).

MI355X-first design
  * gradients live in a few large flat buckets (default 128 MiB, sized for 288 GB HBM and for xGMI's
    per-link bandwidth: few, large collectives).  The HIP backward nodes write parameter gradients
    straight into the bucket (hamspine.rt.grad_arena: zero-copy, no flatten pass),
  * buckets are laid out in reverse parameter order (~ the order backward produces them); when the last
    gradient of a bucket lands, its all-reduce is enqueued on a side HIP stream behind an event, so the
    collective overlaps the rest of backward,
  * one wait on the side stream before optimizer.step() (`finish()`, also run automatically at the end of
    each backward through an autograd-engine callback),
  * the set of parameters that never receive a gradient (find_unused_parameters semantics: the BERT pooler,
    MIBF's I2Iattention) is LEARNT on the first step; from the second step on a bucket counts only its used
    parameters, so the bucket that holds the pooler (the first one in reverse order) launches during backward
    like every other one instead of waiting for finish().  The graph is static per model configuration; if a
    learnt-unused parameter does receive a gradient later the wrapper raises instead of silently dropping it.
"""
import warnings

import torch
import torch.distributed as dist
import torch.nn as nn

from . import rt


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64          # 256-byte aligned slots
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views = [self.flat[o:o + p.numel()].as_strided(p.shape, p.stride()) for p, o in zip(params, self.offsets)]
        self.ready = 0
        self.have = [False] * len(params)
        self.unused = [False] * len(params)       # learnt after the first step: parameters that take no part in the step
        self.need = len(params)                   # gradients that make the bucket complete
        self.launched = False
        self.work = None
        self.streams = {}                         # streams this bucket's gradients became final on


class DataParallel(nn.Module):
    def __init__(self, module, process_group=None, bucket_mb=128, broadcast_buffers=True, zero_copy=None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        self.broadcast_buffers = broadcast_buffers
        params = [p for p in module.parameters() if p.requires_grad]
        self.device = params[0].device
        self.on_gpu = self.device.type == "cuda"
        # rank 0's parameters / buffers win (DDP constructor semantics)
        if self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.pg)
        # buckets in reverse parameter order
        cap = int(bucket_mb * (1 << 20) // 4)
        self.buckets, cur, cur_n = [], [], 0
        for p in reversed(params):
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(_Bucket(cur, self.device))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, self.device))
        # zero-copy bucketing (backward nodes write d(param) straight into the bucket slot) needs every parameter to be
        # used ONCE per step: a model that runs a tower twice (gate / global-local, reference model.py:257-281,334-337)
        # produces two gradients per parameter which autograd sums before AccumulateGrad runs -- both would land in the
        # same slot.  Those models take the copy path (the hook copies the summed gradient into the slot).
        if zero_copy is None:
            zero_copy = not (getattr(module, "gate_enabled", False) or getattr(module, "global_local_enabled", False))
        self.zero_copy = bool(zero_copy)
        self._where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                if self.zero_copy:
                    rt.grad_arena_register(p, b.views[i])
                p.register_post_accumulate_grad_hook(self._on_grad)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        # the text tower's gradients are produced on its own stream while the hooked AccumulateGrad nodes were created on
        # the ambient one; autograd orders the two correctly and says so on every backward
        warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")
        self._callback_queued = False
        self._dirty = False
        self._learnt_unused = False
        self.stats = {"launched_in_backward": 0, "launched_in_finish": 0}   # how many bucket collectives overlapped backward
        nccl = self.world > 1 and dist.get_backend(self.pg) == "nccl"     # AVG exists in RCCL only; gloo sums, then / world
        self._avg_op = dist.ReduceOp.AVG if (self.on_gpu and nccl) else dist.ReduceOp.SUM

    # ------------------------------------------------------------------------------------------
    def sync_buffers(self):
        """Buffers (BatchNorm running statistics) follow rank 0, as with DDP's broadcast_buffers=True.
        Train-mode BatchNorm never reads them, so instead of ~100 tiny broadcasts per forward the sync is
        one coalesced broadcast whenever they become observable: state_dict(), eval(), or on request."""
        if self.world > 1 and self.broadcast_buffers:
            bufs = [b.data for b in self.module.buffers() if b.is_floating_point()]
            if bufs:
                dist._broadcast_coalesced(self.pg or dist.group.WORLD, bufs, 256 << 20, 0)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def train(self, mode=True):
        if not mode and self.training:
            self.sync_buffers()
        return super().train(mode)

    def _on_grad(self, p):
        """post-accumulate hook (autograd thread; the AccumulateGrad node's stream is current and already ordered
        behind the node that produced the gradient)"""
        b, i = self._where[p]
        if b.unused[i]:
            raise RuntimeError("hamspine.ddp: a parameter that received no gradient on the first step received one now; "
                               "the unused-parameter set is learnt once (static graph per model configuration)")
        if b.launched:
            raise RuntimeError("hamspine.ddp: a gradient arrived after its bucket's all-reduce was launched (a second "
                               "backward() before finish()/optimizer.step(): gradient accumulation is not supported)")
        view = b.views[i]
        if p.grad.data_ptr() != view.data_ptr():             # gradient produced outside the arena: copy once
            view.copy_(p.grad)
            p.grad = view
        self._dirty = True
        if self.on_gpu:
            s = torch.cuda.current_stream(self.device)
            b.streams[s.cuda_stream] = s
        if not b.have[i]:
            b.have[i] = True
            b.ready += 1
        if not self._callback_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
            self._callback_queued = True
        # (a parameter used several times in one step still fires this hook once: autograd sums its gradients first)
        if b.ready == b.need and not b.launched:
            self.stats["launched_in_backward"] += 1
            self._launch(b)

    def _launch(self, b):
        b.launched = True
        if self.world == 1:
            return
        if self.on_gpu:
            # order the collective behind EVERY stream one of the bucket's gradients became final on (the two towers run
            # on different streams), not just the stream of the hook that completed the bucket
            streams, b.streams = b.streams, {}
            cur = torch.cuda.current_stream(self.device)
            streams[cur.cuda_stream] = cur
            for s in streams.values():
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                b.work = dist.all_reduce(b.flat, op=self._avg_op, group=self.pg, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def finish(self):
        """Reduce the buckets that never filled (unused parameters) and wait for all collectives.  Called by
        the autograd engine at the end of backward; safe to call again before optimizer.step()."""
        self._callback_queued = False
        if not self._dirty:
            return
        self._dirty = False
        for b in self.buckets:
            if not b.launched and b.ready > 0:
                for i, got in enumerate(b.have):
                    if not got:
                        b.views[i].zero_()                   # absent on every rank alike: contributes zeros
                self.stats["launched_in_finish"] += 1
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
                if self._avg_op == dist.ReduceOp.SUM and self.world > 1:
                    b.flat.div_(self.world)
        if self.on_gpu and self.world > 1:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        for b in self.buckets:
            for i, p in enumerate(b.params):
                if not b.have[i] and p.grad is not None and p.grad.data_ptr() == b.views[i].data_ptr():
                    p.grad = None                            # unused this step: keep DDP's grad=None contract
            if not self._learnt_unused:                      # first step: learn who takes part (see module docstring)
                b.unused = [not h for h in b.have]
                b.need = sum(b.have)
            b.ready = 0
            b.have = [False] * len(b.params)
            b.launched = False
            b.streams = {}
        self._learnt_unused = True

    def zero_grad(self, set_to_none=True):
        self.module.zero_grad(set_to_none=set_to_none)

    def state_dict(self, *a, **k):
        self.sync_buffers()
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.module.load_state_dict(*a, **k)
