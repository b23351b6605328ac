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
    gradient of a bucket lands, its exchange is enqueued on a side HIP stream, so the collective overlaps the
    rest of backward.  Buckets are always launched IN INDEX ORDER and EVERY bucket is launched every step, so all
    ranks issue the same sequence of collectives whatever their local gradient sets look like,
  * the whole-tower executors (hamspine.tower: one C call per tower and direction) stay on under data
    parallelism: the tower's backward records one HIP event per bucket at the point of its stream where that
    bucket's last gradient has been enqueued (`hs_grad_milestones`, csrc/blocks.hip), and the bucket's exchange
    waits for THAT event, not for the tower's end -- bucket-level overlap without per-block Python nodes,
  * one wait on the side stream before optimizer.step() (`finish()`, also run automatically at the end of
    each backward through an autograd-engine callback),
  * the set of parameters that never receive a gradient (find_unused_parameters semantics: the BERT pooler,
    MIBF's I2Iattention) is LEARNT on the first step and made identical on all ranks by one all-reduce of the
    used bitmap; from the second step on a bucket counts only its used parameters, so the bucket that holds the
    pooler launches during backward like every other one instead of waiting for finish().  If a learnt-unused
    parameter does receive a gradient later, the wrapper re-learns: the parameter counts as used from then on
    (this step included when its bucket has not been launched yet).  Only a gradient that arrives AFTER its
    bucket's exchange has started cannot be honoured; that raises, naming `static_unused=False`, the mode in
    which buckets holding not-always-used parameters wait for the end of backward (torch DDP's behaviour).
  * exchange algorithm (`algo=`, env HAMSPINE_DDP_ALGO): "allreduce" (default: one RCCL all-reduce(AVG) of the
    f32 bucket, the reference's semantics bit for bit in f32), "direct" (reduce-scatter as an all-to-all of
    shards + a local f32 sum, then all-gather: on xGMI's fully connected point-to-point links both phases use all
    7 links of a GPU at once instead of a ring) and "direct_bf16" (the same with a bf16 wire format:
    every rank sends bf16 shards, the owner of a shard sums them in f32, the reduced shard travels back in bf16
    -- half the bytes, f32 accumulation).
"""
import os
import warnings

import torch
import torch.distributed as dist
import torch.nn as nn

from . import rt


class _Bucket:
    def __init__(self, params, device, world):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64          # 256-byte aligned slots
        self.numel = n
        pad = (-n) % (64 * max(world, 1))             # the direct algorithms cut the bucket into `world` equal shards
        self.flat = torch.zeros(n + pad, dtype=torch.float32, device=device)
        self.views = [self.flat[o:o + p.numel()].as_strided(p.shape, p.stride()) for p, o in zip(params, self.offsets)]
        self.ready = 0
        self.have = [False] * len(params)
        self.unused = [False] * len(params)       # learnt after the first step: parameters that take no part in the step
        self.need = len(params)                   # gradients that make the bucket complete
        self.launched = False
        self.work = []
        self.streams = {}                         # streams this bucket's gradients became final on (hook time)
        self.events = {}                          # stream id -> event recorded INSIDE a tower backward (milestones)
        self.hook_events = {}                     # stream id -> event re-recorded by every hook on that stream
        self.hook_sids = set()                    # streams whose hook event was recorded in this step
        self.spare = {}                           # stream id -> reusable event objects
        self.post = None


class DataParallel(nn.Module):
    def __init__(self, module, process_group=None, bucket_mb=128, broadcast_buffers=True, zero_copy=None, algo=None,
                 static_unused=True, cut_mb=16):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        self.rank = dist.get_rank(self.pg) if dist.is_initialized() else 0
        self.broadcast_buffers = broadcast_buffers
        self.static_unused = bool(static_unused)
        params = [p for p in module.parameters() if p.requires_grad]
        self.device = params[0].device
        self.on_gpu = self.device.type == "cuda"
        # a collective group exists (possibly of one rank: bench.py --ddp-config runs the whole exchange path at N = 1)
        self.collective = dist.is_initialized() and (self.world > 1 or os.environ.get("HAMSPINE_DDP_SINGLE") == "1")
        # rank 0's parameters / buffers win (DDP constructor semantics)
        if self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.pg)
        # Buckets in reverse parameter order, at most bucket_mb each, and cut at module boundaries where a cut lets an exchange
        # start earlier: between top-level submodules (the towers finish their backward at different times -- a bucket that
        # mixes the text tower's embeddings with the image tower's layer4 waits for both), and between submodules up to
        # the third name level (image_encoder.*.layer4 | layer3 | ...: the ResNet executor releases its weight gradients stage by stage)
        # once the bucket holds cut_mb.  Every rank derives the same cuts from the same parameter names.
        cap = int(bucket_mb * (1 << 20) // 4)
        cut = int(cut_mb * (1 << 20) // 4)
        names = {id(p): n for n, p in module.named_parameters()}
        self.buckets, cur, cur_n, cur_key = [], [], 0, None
        for p in reversed(params):
            key = tuple(names.get(id(p), "").split(".")[:3])   # e.g. (image_encoder, backbone, layer4) / (text_encoder, bert, encoder)
            boundary = cur and cur_key is not None and ((key[:1] != cur_key[:1] and cur_n * 4 >= (1 << 20)) or (key != cur_key and cur_n >= cut))
            if cur and (cur_n + p.numel() > cap or boundary):
                self.buckets.append(_Bucket(cur, self.device, self.world))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
            cur_key = key
        if cur:
            self.buckets.append(_Bucket(cur, self.device, self.world))
        # zero-copy bucketing (backward nodes write d(param) straight into the bucket slot) needs every parameter to be
        # used ONCE per step: a model that runs a tower twice (gate / global-local, reference model.py:257-281,334-337)
        # produces two gradients per parameter which autograd sums before AccumulateGrad runs -- both would land in the
        # same slot.  Those models take the copy path (the hook copies the summed gradient into the slot).
        if zero_copy is None:
            zero_copy = not (getattr(module, "gate_enabled", False) or getattr(module, "global_local_enabled", False))
        self.zero_copy = bool(zero_copy)
        self._where = {}
        self._hook_handles = []
        self._order = []                              # (bucket, index) in bucket order: the layout of the used bitmap
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
                self._order.append((b, i))
                if self.zero_copy:
                    rt.grad_arena_register(p, b.views[i])
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        # (Normal priority on purpose.  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES = 4) and a queue whose
        # head is a pending event wait holds back every stream sharing it; moving this stream and RCCL's internal one
        # (TORCH_NCCL_HIGH_PRIORITY=1) to the high-priority queues was measured: the one-rank exchange kernels then starve
        # the towers, 25 ms per step.  GPU_MAX_HW_QUEUES=8: 16.7 ms.  profiles/round3_ddp_queue_experiments.txt)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        # the text tower's gradients are produced on its own stream while the hooked AccumulateGrad nodes were created on
        # the ambient one; autograd orders the two correctly and says so on every backward
        warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")
        self._callback_queued = False
        self._dirty = False
        self._learnt_unused = False
        self._next = 0                                # position in _launch_order of the next bucket to launch (launches are in order)
        # Launch order: bucket index order until it is learnt.  On the third step every bucket's events carry timestamps; the
        # buckets are then sorted by the device time at which their last gradient was final (rank 0's order, broadcast), so
        # that on the one exchange stream no bucket waits behind one that becomes ready later -- index order does that
        # whenever the module's parameter order is not its backward's completion order (two towers on two streams; the image
        # tower's layer4 is final long before the text tower's embeddings, whatever the order of the submodules).
        self._launch_order = list(range(len(self.buckets)))
        # OFF unless HAMSPINE_DDP_LEARN_ORDER=1 (CPU / gloo runs: always index order).  Measured on one rank: C3's exposed tail
        # drops from 0.75 to 0.09 ms, C2 is unchanged -- but with the learnt order the first ResNet bucket's exchange is issued
        # between the two towers' backward calls, its event wait reaches a hardware queue early, and when the text tower's
        # stream shares that queue (it did after ten wrapper-less training steps, it did not in a fresh process) BERT's
        # backward starts 1-2 ms late: 13.1 vs 12.1 ms per step.  Index order has the same exposure in principle and has not
        # shown it in any context measured.
        self._order_learnt = not (self.on_gpu and self.collective) or os.environ.get("HAMSPINE_DDP_LEARN_ORDER", "0") != "1"
        self._steps = 0
        self._ref_ev = None
        self._covered = set()                         # ids of parameters whose gradients a tower milestone covers (this step)
        self._bulk_done = set()                       # ... and that were booked in bulk when the tower reported its events
        self._bulk_next = None
        self._ms_cache = {}                           # per tower parameter list: (list, bucket-ending members, bulk list, covered ids)
        self._unused_epoch = 0                        # bumped whenever a parameter's unused flag changes
        self.stats = {"launched_in_backward": 0, "launched_in_finish": 0}   # how many bucket collectives overlapped backward
        backend = dist.get_backend(self.pg) if dist.is_initialized() else None
        nccl = backend == "nccl"                      # AVG exists in RCCL only; gloo sums, then / world
        self._avg_op = dist.ReduceOp.AVG if (self.on_gpu and nccl) else dist.ReduceOp.SUM
        self.algo = algo or os.environ.get("HAMSPINE_DDP_ALGO", "allreduce")
        if self.algo not in ("allreduce", "direct", "direct_bf16"):
            raise ValueError(f"hamspine.ddp: unknown exchange algorithm {self.algo!r}")
        if self.on_gpu and self.zero_copy:
            rt.add_milestone_provider(self)

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

    # ---- milestones: called by hamspine.tower around a whole-tower backward --------------------------------------------
    def _event(self, b, sid):
        if not self._order_learnt:                    # timestamps wanted: fresh timing events (the pool holds plain ones)
            return torch.cuda.Event(enable_timing=True)
        pool = b.spare.setdefault(sid, [])
        return pool.pop() if pool else torch.cuda.Event()

    def _mark_step_start(self):
        if not self._order_learnt and self._ref_ev is None and self.on_gpu:
            self._ref_ev = torch.cuda.Event(enable_timing=True)
            self._ref_ev.record(torch.cuda.current_stream(self.device))

    def _milestones(self, params, stream):
        """`params`: a tower's parameters in the order its backward FINISHES their gradients last-to-first reversed, i.e.
        params[0]'s gradient is produced last (hamspine.tower passes bert_params: embeddings first, layer 0, 1, ...; the
        backward walks the layers from the top down).  Returns [(index into params, event handle, key)]: for every bucket
        holding some of these parameters, the member whose gradient is enqueued LAST and the event the executor records
        right after it."""
        # what does not change from step to step (which bucket ends where, who is booked) is worked out once per tower and
        # unused-set: this runs on the autograd thread in front of the tower's one C call
        self._mark_step_start()
        ckey = (id(params), len(params), sum(1 for p in params if p is None), self._unused_epoch)
        ent = self._ms_cache.get(ckey)
        if ent is None or ent[0] is not params:
            first = {}
            bulk = []
            cov = []
            for i, p in enumerate(params):
                w = None if p is None else self._where.get(p)
                if w is None:
                    continue
                cov.append(id(p))                     # its position in the stream is covered by a milestone, not by a hook event
                b, k = w
                if b.unused[k]:
                    continue
                bulk.append((b, k, id(p)))
                if id(b) in first:
                    continue
                first[id(b)] = (i, b)
            ent = self._ms_cache[ckey] = (params, list(first.values()), bulk, cov)
        _, firsts, bulk, cov = ent
        self._covered.update(cov)
        self._bulk_next = (stream, bulk)              # booked in one go when the tower reports its events (_milestones_recorded)
        out = []
        sid = stream.cuda_stream
        for i, b in firsts:
            ev = self._event(b, sid)
            ev.record(stream)                         # creates the HIP event (torch creates it lazily); re-recorded by the executor
            out.append((i, ev.cuda_event, (b, sid, ev)))
        return out

    def _milestones_recorded(self, keys):
        for b, sid, ev in keys:
            old = b.events.get(sid)
            if old is not None:
                b.spare[sid].append(old)
            b.events[sid] = ev
        # The tower's backward has enqueued every gradient it owns (straight into the bucket slots): book them all here instead
        # of in ~200 post-accumulate hooks (10 us of Python each on the autograd thread: with two towers the host ended a C2
        # backward 0.3 ms after the GPU, a C3 one 0.9 ms after).  The hooks of these parameters return at once (_bulk_done).
        pend, self._bulk_next = self._bulk_next, None
        if pend:
            stream, bulk = pend
            sid = stream.cuda_stream
            touched = False
            for b, k, pid in bulk:
                if b.launched:
                    continue                          # (left to the hook, which reports the late gradient)
                if not b.have[k]:
                    b.have[k] = True
                    b.ready += 1
                b.streams[sid] = stream
                self._bulk_done.add(pid)
                touched = True
            if touched:
                self._dirty = True
                if not self._callback_queued:
                    torch.autograd.Variable._execution_engine.queue_callback(self.finish)
                    self._callback_queued = True
                self._launch_ready(in_backward=True)

    # ---- hooks ---------------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        """post-accumulate hook (autograd thread; the AccumulateGrad node's stream is current and already ordered
        behind the node that produced the gradient)"""
        if id(p) in self._bulk_done:                         # a tower gradient, already booked with its milestone
            return
        self._mark_step_start()
        b, i = self._where[p]
        if b.launched:
            if b.unused[i]:
                raise RuntimeError(
                    "hamspine.ddp: a parameter that took no part in the first step received a gradient after its bucket's "
                    "exchange had started in this step.  The set of used parameters changes from step to step in this model: "
                    "construct DataParallel(..., static_unused=False) so that buckets holding such parameters wait for the end "
                    "of backward")
            raise RuntimeError("hamspine.ddp: a gradient arrived after its bucket's all-reduce was launched (a second "
                               "backward() before finish()/optimizer.step(): gradient accumulation is not supported)")
        if b.unused[i]:                                      # the used set grew: count it from now on (re-learn)
            b.unused[i] = False
            b.need += 1
            self._unused_epoch += 1
        view = b.views[i]
        if p.grad.data_ptr() != view.data_ptr():             # gradient produced outside the arena: copy once
            view.copy_(p.grad)
            p.grad = view
        self._dirty = True
        if self.on_gpu:
            s = torch.cuda.current_stream(self.device)
            sid = s.cuda_stream
            b.streams[sid] = s
            if id(p) not in self._covered:                   # not a tower-milestone gradient: its stream needs an event --
                b.hook_sids.add(sid)                         # recorded ONCE, when the bucket launches (_launch), not per hook
        if not b.have[i]:
            b.have[i] = True
            b.ready += 1
        if not self._callback_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
            self._callback_queued = True
        # (a parameter used several times in one step still fires this hook once: autograd sums its gradients first)
        self._launch_ready(in_backward=True)

    def _complete(self, b):
        if b.ready < b.need:
            return False
        # static_unused=False: a bucket that holds parameters which are not used in every step cannot know during backward
        # whether they will still arrive
        return self.static_unused or b.need == len(b.params)

    def _launch_ready(self, in_backward):
        """launch, in the agreed order, every bucket that is complete (in finish(): every remaining bucket)"""
        while self._next < len(self.buckets):
            b = self.buckets[self._launch_order[self._next]]
            if in_backward:
                # before the unused set is known a bucket is complete when ALL its parameters have arrived
                ok = self._complete(b) if self._learnt_unused else b.ready == len(b.params)
                if not ok:
                    return
            self.stats["launched_in_backward" if in_backward else "launched_in_finish"] += 1
            self._launch(b)
            self._next += 1

    def _launch(self, b):
        b.launched = True
        if not self.collective:
            return
        absent = [i for i, got in enumerate(b.have) if not got]
        if self.on_gpu:
            # order the exchange behind the point of EVERY stream where this bucket's last gradient was enqueued: the tower
            # milestone event where a tower's backward recorded one, and the event the hooks of other parameters left
            for sid, ev in b.events.items():
                self.comm_stream.wait_event(ev)
            for sid in b.hook_sids:
                # the position of that stream NOW is behind every gradient its hooks reported (a bucket launches from its last
                # hook, or from finish()); one record per bucket and stream instead of one per parameter (~5 us each on the
                # autograd thread, in front of the towers' backward calls)
                ev = b.hook_events.get(sid)
                if ev is None:
                    ev = b.hook_events[sid] = self._event(b, sid)
                ev.record(b.streams[sid])
                self.comm_stream.wait_event(ev)
            for sid, st in b.streams.items():
                if sid not in b.events and sid not in b.hook_sids:
                    self.comm_stream.wait_stream(st)
            if not b.events and not b.hook_sids:             # no gradient here at all: still behind the last step's readers
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                for i in absent:
                    b.views[i].zero_()                       # absent here (unused, or used on another rank only): contributes zeros
                self._exchange(b)
        else:
            for i in absent:
                b.views[i].zero_()
            self._exchange(b)

    def _exchange(self, b):
        """average b.flat over the ranks (asynchronously where the backend allows)"""
        W = self.world
        if self.algo == "allreduce":
            b.work.append(dist.all_reduce(b.flat, op=self._avg_op, group=self.pg, async_op=True))
            b.post = "div" if self._avg_op == dist.ReduceOp.SUM and W > 1 else None
            return
        # "direct": every rank sends shard r of its bucket to rank r (all-to-all: on xGMI's fully connected point-to-point
        # links a GPU drives its 7 links at once), the owner sums the W copies of its shard in f32, and the reduced shards
        # travel back (all-gather).  "direct_bf16": the same with bf16 on the wire -- half the bytes, f32 accumulation.
        wire = torch.bfloat16 if self.algo == "direct_bf16" else torch.float32
        shard = b.flat.numel() // W
        send = b.flat.to(wire) if wire != torch.float32 else b.flat
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.pg)            # recv[r * shard:(r + 1) * shard] = rank r's copy of MY shard
        red = recv.view(W, shard).float().sum(0).div_(W).to(wire)
        if wire == torch.float32:
            dist.all_gather_into_tensor(b.flat, red, group=self.pg)
        else:
            out = torch.empty(W * shard, dtype=wire, device=b.flat.device)
            dist.all_gather_into_tensor(out, red, group=self.pg)
            b.flat.copy_(out)
        b.post = None

    def _learn(self):
        """first step: who takes part in the step -- the union over ranks (one all-reduce of the used bitmap), so that every
        rank counts the same parameters and launches its buckets at matching points"""
        flags = torch.tensor([1 if b.have[i] else 0 for b, i in self._order], dtype=torch.int32)
        if self.world > 1:
            dev = flags.to(self.device) if (self.on_gpu and dist.get_backend(self.pg) == "nccl") else flags
            dist.all_reduce(dev, op=dist.ReduceOp.MAX, group=self.pg)
            flags = dev.cpu()
        used = flags.tolist()
        for (b, i), u in zip(self._order, used):
            b.unused[i] = not u
        for b in self.buckets:
            b.need = sum(1 for u in b.unused if not u)
        self._learnt_unused = True
        self._unused_epoch += 1

    def _learn_order(self):
        """third step: sort the buckets by the device time at which their last gradient was final (rank 0 decides)"""
        torch.cuda.synchronize(self.device)
        ref = self._ref_ev
        times = []
        for k, b in enumerate(self.buckets):
            t = None
            evs = list(b.events.values()) + [b.hook_events[sid] for sid in b.hook_sids if sid in b.hook_events]
            for ev in evs:
                try:
                    dt = ref.elapsed_time(ev) if ref is not None else 0.0
                except Exception:          # a plain event left from before the learning steps: no timestamp
                    dt = None
                if dt is not None:
                    t = dt if t is None else max(t, dt)
            times.append(float("inf") if t is None else t)
        order = sorted(range(len(self.buckets)), key=lambda k: (times[k], k))
        if self.world > 1 and self.collective:
            o = torch.tensor(order, dtype=torch.int32, device=self.device if dist.get_backend(self.pg) == "nccl" else "cpu")
            dist.broadcast(o, src=0, group=self.pg)
            order = [int(v) for v in o.cpu().tolist()]
        self._launch_order = order
        self._order_learnt = True
        self.stats["launch_order"] = order
        self.stats["ready_ms"] = [round(t, 3) if t != float("inf") else None for t in times]

    def finish(self):
        """Launch the buckets that did not complete during backward (unused parameters) and wait for all exchanges.
        Called by the autograd engine at the end of backward; safe to call again before optimizer.step()."""
        self._callback_queued = False
        if not self._dirty:
            return
        self._dirty = False
        self._launch_ready(in_backward=False)
        for b in self.buckets:
            for w in b.work:
                w.wait()
            if b.work and b.post == "div":
                b.flat.div_(self.world)
            b.work = []
        if self.on_gpu and self.collective:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        first = not self._learnt_unused
        if first:
            self._learn()
        self._steps += 1
        drop_timed = False
        if not self._order_learnt and self._steps >= 3:
            self._learn_order()
            drop_timed = True
        for b in self.buckets:
            for i, p in enumerate(b.params):
                if not b.have[i] and p.grad is not None and p.grad.data_ptr() == b.views[i].data_ptr():
                    p.grad = None                            # unused this step: keep DDP's grad=None contract
            b.ready = 0
            b.have = [False] * len(b.params)
            b.launched = False
            b.streams = {}
            b.hook_sids = set()
            for sid, ev in b.events.items():
                b.spare.setdefault(sid, []).append(ev)
            b.events = {}
        if drop_timed:                                       # the timestamped events of the learning steps are not reused
            for b in self.buckets:
                b.spare = {}
                b.hook_events = {}
        self._covered = set()
        self._bulk_done = set()
        self._bulk_next = None
        self._ref_ev = None
        self._next = 0

    def detach(self):
        """undo the wrapping: remove the gradient hooks and the bucket slots (the module trains stand-alone again)"""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        if self.zero_copy:
            rt.grad_arena_clear()
        for p in self._where:
            p.grad = None
        self._where = {}

    def zero_grad(self, set_to_none=True):
        self.module.zero_grad(set_to_none=set_to_none)

    def state_dict(self, *a, **k):
        self.sync_buffers()
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.module.load_state_dict(*a, **k)
