"""Whole-tower autograd nodes: one C call per tower per direction (hs_resnet_fwd/bwd, hs_bert_fwd/bwd).

The reference loop calls the model once per step (scripts/train.py:373-385, mibf_net/train_resnet.py:30-32); with one
autograd node per residual block / BertLayer a C2 step cost ~70 Python -> C round trips (descriptor rebuild, three layout
passes and several tensor allocations each): 11 ms of host time against 12.6 ms of GPU time.  Here the descriptors of a
whole tower are built ONCE per (module, shape, mode) and reused; a step pays one Function.apply, one saved-arena
allocation and one C call per tower and direction.

Parameter gradients of a tower live in one flat f32 buffer owned by the cache entry (fresh view objects are handed to
autograd every step, so AccumulateGrad adopts them without a copy); with hamspine.ddp the bucket slots are used instead.
The per-block modules stay the hookable path: a tower falls back to them whenever any of its submodules carries a hook
(Grad-CAM: reference scripts/run_analysis.py:126-133).
"""
import ctypes as C

import torch
from torch.autograd import Function
from torch.nn.modules import module as _nn_module

import hamspine

from . import _lib as L
from . import rt

_CACHE_ATTR = "_hamspine_tower_cache"
DEBUG_KEEP = None      # tests set this to a list: every ResNet tower forward appends (cache entry, saved arena)


def _has_hooks(root):
    """any forward / backward hook on `root` or below it, or registered globally"""
    if (_nn_module._global_forward_hooks or _nn_module._global_forward_pre_hooks or _nn_module._global_backward_hooks or
            getattr(_nn_module, "_global_backward_pre_hooks", None)):
        return True
    # the walk over root.modules() costs 0.27 ms on ResNet50 / BERT-base (three towers a step): the module list is kept
    # on the root, keyed by the identity of its direct children (a tower whose inner structure is edited in place after its
    # first forward also invalidates the descriptor cache below, which is rebuilt from the same walk)
    key = tuple(map(id, root._modules.values()))
    cached = root.__dict__.get("_hamspine_module_list")
    if cached is None or cached[0] != key:
        cached = (key, list(root.modules()))
        root.__dict__["_hamspine_module_list"] = cached
    for m in cached[1]:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return True
    return False


def towers_enabled():
    """The whole-tower nodes are the default everywhere, data parallelism included: hamspine.ddp gets one HIP event per
    gradient bucket from inside the tower's backward (hs_grad_milestones), so a bucket's exchange starts behind its own last
    gradient while the rest of the tower still runs; under torch's DistributedDataParallel (reference
    mibf_net/train_resnet.py:134) the tower's gradients become ready together at its end, which costs overlap, not
    correctness.  HAMSPINE_TOWER_EXEC=0 selects the per-block nodes."""
    import os
    return os.environ.get("HAMSPINE_TOWER_EXEC", "1") != "0"


def _with_milestones(ent, params, call):
    """run `call()` (one hs_*_bwd) with the data-parallel wrappers' bucket milestones registered: each gets the event of
    every bucket recorded at the point of this stream where that bucket's last gradient has been enqueued"""
    stream = torch.cuda.current_stream()
    # (parameters that get no gradient from this backward are passed as None: the wrapper books only what is produced)
    mp = getattr(ent, "_ms_params", None)              # (the same list object every step: the wrapper caches its bookkeeping on it)
    if mp is None or mp[0] is not ent.grad_ptrs:
        mp = ent._ms_params = (ent.grad_ptrs, [p if ent.grad_ptrs[i] else None for i, p in enumerate(params)])
    asked = rt.grad_milestones(mp[1], stream) if ent.store is None else []
    flat = [(ent.grad_ptrs[i], ev) for _, ms in asked for i, ev, _ in ms if ent.grad_ptrs[i]]
    lib = L.lib()
    if flat:
        n = len(flat)
        L.check(lib.hs_grad_milestones(n, (C.c_void_p * n)(*[q for q, _ in flat]), (C.c_void_p * n)(*[e for _, e in flat])),
                "hs_grad_milestones")
    try:
        call()
    finally:
        if flat:
            lib.hs_grad_milestones(0, None, None)
    for o, ms in asked:
        o._milestones_recorded([k for _, _, k in ms])


class _GradStore:
    """flat f32 gradient buffer of a tower + the slot (offset, shape, strides) of every parameter"""

    def __init__(self, params, needs, device):
        self.slots = []
        n = 0
        for p, need in zip(params, needs):
            if need:
                self.slots.append((n, tuple(p.shape), tuple(p.stride())))
                n += (p.numel() + 63) // 64 * 64
            else:
                self.slots.append(None)
        self.flat = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        self.base = self.flat.data_ptr()

    def ptr(self, i):
        s = self.slots[i]
        return None if s is None else self.base + 4 * s[0]

    def views(self):
        f = self.flat
        return [None if s is None else f.as_strided(s[1], s[2], s[0]) for s in self.slots]


def _param_grad_ptrs(params, needs, device):
    """where the backward writes d(param): DDP bucket slots when hamspine.ddp registered them, else a private flat buffer.
    Returns (store or None, pointer list, view factory)."""
    arena = [rt._grad_arena.get(p.data_ptr()) if need else None for p, need in zip(params, needs)]
    if any(a is not None for a in arena):
        if not all((a is not None and a.shape == p.shape) or not need for a, p, need in zip(arena, params, needs)):
            raise L.HamspineError("hamspine.tower: only some parameters of a tower have DDP bucket slots")
        return None, [None if a is None else a.data_ptr() for a in arena], (lambda: [None if a is None else a.detach() for a in arena])
    store = _GradStore(params, needs, device)
    return store, [store.ptr(i) for i in range(len(params))], store.views


# =====================================================================================================================
# ResNet
# =====================================================================================================================
def _fill_cb(cb, conv, bn, dw=None, dg=None, db=None):
    cb.Cin, cb.Cout, cb.R, cb.stride, cb.pad = conv.geo()
    w = conv.weight
    if not w.is_contiguous(memory_format=torch.channels_last):
        raise L.HamspineError("hamspine.tower: conv filters must be channels_last (KRSC) in memory")
    cb.w = w.data_ptr()
    cb.gamma, cb.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
    cb.running_mean = bn.running_mean.data_ptr() if bn.running_mean is not None else None
    cb.running_var = bn.running_var.data_ptr() if bn.running_var is not None else None
    cb.dw, cb.dgamma, cb.dbeta = dw, dg, db


def resnet_blocks(model):
    return [b for layer in (model.layer1, model.layer2, model.layer3, model.layer4) for b in layer]


def resnet_params(model):
    """stem (w, gamma, beta), then per block its main stages and the optional downsample pair, each (w, gamma, beta)"""
    ps = [model.conv1.weight, model.bn1.weight, model.bn1.bias]
    bns = [model.bn1]
    for b in resnet_blocks(model):
        pairs = b._pairs() + ([(b.downsample[0], b.downsample[1])] if b.downsample is not None else [])
        for c, n in pairs:
            ps += [c.weight, n.weight, n.bias]
            bns.append(n)
    return ps, bns


class _ResnetEntry:
    def __init__(self, model, x_shape, dtype, training, inference, taps, params, needs, device):
        blocks = resnet_blocks(model)
        if len(blocks) > L.RESNET_MAX_BLOCKS:
            raise L.HamspineError(f"hamspine.tower: {len(blocks)} residual blocks (max {L.RESNET_MAX_BLOCKS})")
        self.store, gp, self.make_views = _param_grad_ptrs(params, needs, device)
        self.grad_ptrs = gp
        self.param_ptrs = [p.data_ptr() for p in params]
        d = L.ResnetDesc()
        N, _, H, W = x_shape
        hd = rt.hs_dtype(dtype)
        bn0 = model.bn1
        eps, mom = bn0.eps, (bn0.momentum if bn0.momentum is not None else 0.1)
        st = d.stem
        st.dtype, st.N, st.H, st.W, st.training, st.eps, st.momentum, st.inference = hd, N, H, W, int(training), eps, mom, int(inference)
        _fill_cb(st.cb, model.conv1, model.bn1, gp[0], gp[1], gp[2])
        gi = 3
        H, W = ((H + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1, ((W + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
        d.n_blocks = len(blocks)
        for i, b in enumerate(blocks):
            bd = d.blocks[i]
            pairs = b._pairs()
            bd.dtype, bd.N, bd.H, bd.W, bd.training, bd.eps, bd.momentum, bd.inference = hd, N, H, W, int(training), eps, mom, int(inference)
            bd.n_main = len(pairs)
            for j, (c, n) in enumerate(pairs):
                if n.eps != eps or (n.momentum if n.momentum is not None else 0.1) != mom:
                    raise L.HamspineError("hamspine.tower: BatchNorm layers of a tower must share eps / momentum")
                _fill_cb(bd.main[j], c, n, gp[gi], gp[gi + 1], gp[gi + 2])
                gi += 3
                H = (H + 2 * c.padding[0] - c.kernel_size[0]) // c.stride[0] + 1
                W = (W + 2 * c.padding[0] - c.kernel_size[0]) // c.stride[0] + 1
            bd.has_ds = 1 if b.downsample is not None else 0
            if b.downsample is not None:
                _fill_cb(bd.ds, b.downsample[0], b.downsample[1], gp[gi], gp[gi + 1], gp[gi + 2])
                gi += 3
        d.n_taps = len(taps)
        for t, bi in enumerate(taps):
            d.tap_block[t] = bi
        self.desc = d
        self.plan = L.ResnetPlan()
        L.check(L.lib().hs_resnet_query(C.byref(d), C.byref(self.plan)), "hs_resnet_query")
        es = 2 if dtype == torch.bfloat16 else 4
        self.taps = [(int(self.plan.tap_offset[t]), (N, int(self.plan.tap_H[t]), int(self.plan.tap_W[t]), int(self.plan.tap_C[t])), es)
                     for t in range(len(taps))]
        self.dtype = dtype
        self.arena_version = rt.arena_version()
        self.outstanding = self.peak = 0
        # bf16 shadows of the block convolutions' filters (the stem's filter is re-packed, not cast): rt.ensure_shadows
        self.shadow_groups = [[params[i]] for i in range(3, len(params), 3)] if dtype == torch.bfloat16 else []

    def valid_for(self, params):
        if self.arena_version != rt.arena_version():
            return False
        return all(p.data_ptr() == q for p, q in zip(params, self.param_ptrs))


class _Pending:
    """one differentiable forward of a tower whose backward has not run yet (released by the backward, or when the graph
    is dropped without one)"""

    def __init__(self, ent):
        self.ent = ent
        ent.outstanding += 1
        ent.peak = max(ent.peak, ent.outstanding)

    def finish(self):
        ent, self.ent = self.ent, None
        if ent is not None:
            ent.outstanding = max(0, ent.outstanding - 1)
            if ent.outstanding == 0:
                ent.peak = 0

    __del__ = finish


def _forward_begins(ent, needs):
    """a forward that will be backpropagated starts: count the forwards of this entry whose backward is still pending"""
    return _Pending(ent) if any(needs) else None


def _shared_buffer_unsafe(ent, params):
    """may this backward write into the entry's cached gradient buffer?  Not when the tower ran more than once in the graph
    being differentiated (gate / global-local models: reference model.py:257-281,334-337 -- autograd sums the gradients of
    the passes AFTER all of them were produced, so each pass needs memory of its own), and not when a parameter still holds
    the buffer from an earlier backward (accumulation without zero_grad)."""
    if ent.store is None:
        return ent.peak > 1            # DDP bucket slots: zero-copy is only registered for single-pass models (hamspine.ddp)
    if ent.peak > 1:
        return True
    return any(p.grad is not None and p.grad.data_ptr() == ent.store.ptr(i) for i, p in enumerate(params))


class ResNetTowerFn(Function):
    """image (N,3,H,W) f32 -> the requested block outputs (NCHW-shaped, NHWC memory, compute dtype)"""

    @staticmethod
    def forward(ctx, image, holder, *params):
        model, taps, training = holder
        rt.need_gpu(image, *params)
        image = image.contiguous()
        if image.dtype != torch.float32:
            image = image.float()
        dtype = hamspine.compute_dtype()
        needs = tuple(ctx.needs_input_grad[2:])
        inference = not (training or any(needs))
        key = ("resnet", tuple(image.shape), dtype, training, inference, taps, needs)
        cache = model.__dict__.setdefault(_CACHE_ATTR, {})
        ent = cache.get(key)
        if ent is None or not ent.valid_for(params):
            ent = cache[key] = _ResnetEntry(model, image.shape, dtype, training, inference, taps, params, needs, image.device)
        rt.ensure_shadows(ent.shadow_groups)
        saved = torch.empty(int(ent.plan.saved_bytes), dtype=torch.uint8, device=image.device)
        ws = rt.workspace(int(ent.plan.ws_bytes), image.device)
        L.check(L.lib().hs_resnet_fwd(C.byref(ent.desc), image.data_ptr(), saved.data_ptr(), saved.numel(), ws.data_ptr(),
                                      ws.numel(), rt.stream()), "hs_resnet_fwd")
        outs = []
        for off, (n, h, w, c), es in ent.taps:
            t = saved[off:off + n * h * w * c * es].view(dtype).view(n, h, w, c).permute(0, 3, 1, 2)
            outs.append(t)
        ctx.ent, ctx.saved_buf = ent, saved
        ctx.params = params
        ctx.counted = _forward_begins(ent, needs)
        if DEBUG_KEEP is not None:
            DEBUG_KEEP.append((ent, saved))
        # the outputs are views into the arena the backward reads (ReLU masks, BatchNorm inputs): registering them makes
        # autograd's version counters catch a consumer that modifies one in place (it would corrupt the backward silently)
        ctx.save_for_backward(*outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        ent, saved, params = ctx.ent, ctx.saved_buf, ctx.params
        if saved is None:
            raise RuntimeError("hamspine.tower: backward through a tower a second time: its saved activations are freed by the "
                               "first backward (retain_graph=True is not supported on the tower nodes)")
        _ = ctx.saved_tensors                        # raises if an output was modified in place since the forward
        n = len(ent.taps)
        ptrs = (C.c_void_p * n)()
        keep = []
        for t, dy in enumerate(dys):
            if dy is None:
                ptrs[t] = None
                continue
            dy = rt.as_cl(dy, ent.dtype)
            keep.append(dy)
            ptrs[t] = dy.data_ptr()
        desc = ent.desc
        views = ent.make_views
        if _shared_buffer_unsafe(ent, params):      # one-off private gradient buffers for this backward
            desc, views = _resnet_desc_with_fresh_grads(ent, params, tuple(ctx.needs_input_grad[2:]), saved.device)
        ws = rt.workspace(int(ent.plan.ws_bytes), saved.device)

        def call():
            L.check(L.lib().hs_resnet_bwd(C.byref(desc), ptrs, saved.data_ptr(), saved.numel(), ws.data_ptr(), ws.numel(),
                                          rt.stream()), "hs_resnet_bwd")
        if desc is ent.desc:
            _with_milestones(ent, params, call)
        else:
            call()
        ctx.saved_buf = None
        if ctx.counted is not None:
            ctx.counted.finish()
        return (None, None, *views())


def _resnet_desc_with_fresh_grads(ent, params, needs, device):
    store = _GradStore(params, needs, device)
    d = L.ResnetDesc()
    C.memmove(C.byref(d), C.byref(ent.desc), C.sizeof(L.ResnetDesc))
    gi = 0

    def put(cb):
        nonlocal gi
        cb.dw, cb.dgamma, cb.dbeta = store.ptr(gi), store.ptr(gi + 1), store.ptr(gi + 2)
        gi += 3
    put(d.stem.cb)
    for i in range(d.n_blocks):
        for j in range(d.blocks[i].n_main):
            put(d.blocks[i].main[j])
        if d.blocks[i].has_ds:
            put(d.blocks[i].ds)
    return d, store.views


def resnet_taps(model, x, tap_layers):
    """run `model` (hamspine.nn.ResNet) up to layer4 through the tower executor; tap_layers: which of
    ("layer2", "layer3", "layer4") to return.  Returns None when the tower path does not apply (hooks, CPU)."""
    if not (towers_enabled() and x.is_cuda and x.dim() == 4 and x.shape[1] == 3):
        return None
    if _has_hooks(model):
        return None
    counts = [len(model.layer1), len(model.layer2), len(model.layer3), len(model.layer4)]
    ends = {"layer1": counts[0] - 1, "layer2": sum(counts[:2]) - 1, "layer3": sum(counts[:3]) - 1, "layer4": sum(counts) - 1}
    taps = tuple(ends[t] for t in tap_layers)
    params, bns = resnet_params(model)
    training = model.training
    if training:
        for bn in bns:
            bn.bump()
    outs = ResNetTowerFn.apply(x, (model, taps, training), *params)
    return outs


# =====================================================================================================================
# BERT
# =====================================================================================================================
def bert_params(model):
    e = model.embeddings
    ps = [e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight, e.LayerNorm.weight,
          e.LayerNorm.bias]
    for layer in model.encoder.layer:
        a, so, o = layer.attention.self, layer.attention.output, layer.output
        ps += [a.query.weight, a.query.bias, a.key.weight, a.key.bias, a.value.weight, a.value.bias,
               so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias,
               layer.intermediate.dense.weight, layer.intermediate.dense.bias,
               o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias]
    return ps


def _lin(l, in_f, out_f, w, b, dw, db):
    l.in_f, l.out_f = in_f, out_f
    l.w, l.b, l.dw, l.db = w.data_ptr(), b.data_ptr(), dw, db


class _BertEntry:
    def __init__(self, model, B, Lq, dtype, training, params, needs, device):
        c = model.config
        if len(model.encoder.layer) > L.BERT_MAX_LAYERS:
            raise L.HamspineError(f"hamspine.tower: {len(model.encoder.layer)} BertLayers (max {L.BERT_MAX_LAYERS})")
        self.store, gp, self.make_views = _param_grad_ptrs(params, needs, device)
        self.grad_ptrs = gp
        self.param_ptrs = [p.data_ptr() for p in params]
        d = L.BertDesc()
        hd = rt.hs_dtype(dtype)
        e = model.embeddings
        H = c.hidden_size
        d.dtype, d.B, d.L, d.hidden = hd, B, Lq, H
        d.vocab, d.max_pos, d.n_types = e.word_embeddings.weight.shape[0], e.position_embeddings.weight.shape[0], e.token_type_embeddings.weight.shape[0]
        d.pad_id = -1 if e.word_embeddings.padding_idx is None else int(e.word_embeddings.padding_idx)
        d.ln_eps = float(e.LayerNorm.eps)
        d.embed_dropout = float(e.dropout.p) if training else 0.0
        d.word, d.pos, d.type0, d.gamma, d.beta = (p.data_ptr() for p in params[:5])
        d.dword, d.dpos, d.dtype0, d.dgamma, d.dbeta = gp[:5]
        d.n_layers = len(model.encoder.layer)
        gi = 5
        for i, layer in enumerate(model.encoder.layer):
            ld = d.layers[i]
            a, so, o = layer.attention.self, layer.attention.output, layer.output
            I = layer._inter
            ld.dtype, ld.B, ld.L, ld.hidden, ld.heads, ld.inter = hd, B, Lq, H, layer._heads, I
            ld.ln_eps = float(so.LayerNorm.eps)
            ld.hidden_dropout = float(so.dropout.p) if training else 0.0
            ld.attn_dropout = float(a.dropout.p) if training else 0.0
            p = params[gi:gi + 16]
            g = gp[gi:gi + 16]
            _lin(ld.q, H, H, p[0], p[1], g[0], g[1])
            _lin(ld.k, H, H, p[2], p[3], g[2], g[3])
            _lin(ld.v, H, H, p[4], p[5], g[4], g[5])
            _lin(ld.ao, H, H, p[6], p[7], g[6], g[7])
            ld.ln1.gamma, ld.ln1.beta, ld.ln1.dgamma, ld.ln1.dbeta = p[8].data_ptr(), p[9].data_ptr(), g[8], g[9]
            _lin(ld.inter_l, H, I, p[10], p[11], g[10], g[11])
            _lin(ld.out_l, I, H, p[12], p[13], g[12], g[13])
            ld.ln2.gamma, ld.ln2.beta, ld.ln2.dgamma, ld.ln2.dbeta = p[14].data_ptr(), p[15].data_ptr(), g[14], g[15]
            gi += 16
        self.desc = d
        sv, ws, off = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(L.lib().hs_bert_query(C.byref(d), C.byref(sv), C.byref(ws), C.byref(off)), "hs_bert_query")
        self.saved_bytes, self.ws_bytes, self.out_off = sv.value, ws.value, off.value
        self.dtype = dtype
        self.shape = (B, Lq, H)
        self.arena_version = rt.arena_version()
        self.outstanding = self.peak = 0
        # bf16 shadows: q, k, v of a layer as the three segments of one buffer (read as the fused [3H][H] weight), the other
        # three Linear weights on their own
        self.shadow_groups = []
        if dtype == torch.bfloat16:
            for i in range(len(model.encoder.layer)):
                b = 5 + 16 * i
                self.shadow_groups += [[params[b], params[b + 2], params[b + 4]], [params[b + 6]], [params[b + 10]], [params[b + 12]]]

    def valid_for(self, params):
        if self.arena_version != rt.arena_version():
            return False
        return all(p.data_ptr() == q for p, q in zip(params, self.param_ptrs))


class BertTowerFn(Function):
    """(input_ids, attention_mask) -> last_hidden_state (B, L, hidden) in the compute dtype"""

    @staticmethod
    def forward(ctx, ids, mask, holder, *params):
        model, training = holder
        rt.need_gpu(ids, mask, *params)
        ids = ids.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.long()
        B, Lq = ids.shape
        if Lq > model.embeddings.position_embeddings.weight.shape[0]:
            raise ValueError(f"sequence length {Lq} exceeds max_position_embeddings")
        dtype = hamspine.compute_dtype()
        needs = tuple(ctx.needs_input_grad[3:])
        key = ("bert", B, Lq, dtype, training, needs)
        cache = model.__dict__.setdefault(_CACHE_ATTR, {})
        ent = cache.get(key)
        if ent is None or not ent.valid_for(params):
            ent = cache[key] = _BertEntry(model, B, Lq, dtype, training, params, needs, ids.device)
        rt.ensure_shadows(ent.shadow_groups)
        saved = torch.empty(ent.saved_bytes, dtype=torch.uint8, device=ids.device)
        ws = rt.workspace(ent.ws_bytes, ids.device)
        seed = (rt.next_seed() * 64) & 0xFFFFFFFFFFFFFFFF if training else 0
        ent.desc.seed = seed
        L.check(L.lib().hs_bert_fwd(C.byref(ent.desc), ids.data_ptr(), rt.p(mask), saved.data_ptr(), saved.numel(),
                                    ws.data_ptr(), ws.numel(), rt.stream()), "hs_bert_fwd")
        Bq, Lq2, H = ent.shape
        es = 2 if dtype == torch.bfloat16 else 4
        out = saved[ent.out_off:ent.out_off + Bq * Lq2 * H * es].view(dtype).view(Bq, Lq2, H)
        ctx.ent, ctx.saved_buf, ctx.seed = ent, saved, seed
        ctx.ids, ctx.mask, ctx.params = ids, mask, params
        ctx.counted = _forward_begins(ent, needs)
        ctx.save_for_backward(out)                   # a view into the arena: in-place edits by a consumer are caught at backward
        return out

    @staticmethod
    def backward(ctx, dy):
        ent, saved, params = ctx.ent, ctx.saved_buf, ctx.params
        if saved is None:
            raise RuntimeError("hamspine.tower: backward through a tower a second time: its saved activations are freed by the "
                               "first backward (retain_graph=True is not supported on the tower nodes)")
        _ = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != ent.dtype:
            dy = dy.to(ent.dtype)
        desc, views = ent.desc, ent.make_views
        if _shared_buffer_unsafe(ent, params):
            desc, views = _bert_desc_with_fresh_grads(ent, params, tuple(ctx.needs_input_grad[3:]), saved.device)
        desc.seed = ctx.seed
        ws = rt.workspace(ent.ws_bytes, saved.device)

        def call():
            L.check(L.lib().hs_bert_bwd(C.byref(desc), ctx.ids.data_ptr(), rt.p(ctx.mask), dy.data_ptr(), saved.data_ptr(),
                                        saved.numel(), ws.data_ptr(), ws.numel(), rt.stream()), "hs_bert_bwd")
        if desc is ent.desc:
            _with_milestones(ent, params, call)       # bert_params order: the backward finishes params[0] (embeddings) last
        else:
            call()
        ctx.saved_buf = None
        if ctx.counted is not None:
            ctx.counted.finish()
        return (None, None, None, *views())


def _bert_desc_with_fresh_grads(ent, params, needs, device):
    store = _GradStore(params, needs, device)
    d = L.BertDesc()
    C.memmove(C.byref(d), C.byref(ent.desc), C.sizeof(L.BertDesc))
    d.dword, d.dpos, d.dtype0, d.dgamma, d.dbeta = (store.ptr(i) for i in range(5))
    gi = 5
    for i in range(d.n_layers):
        ld = d.layers[i]
        g = [store.ptr(gi + k) for k in range(16)]
        ld.q.dw, ld.q.db, ld.k.dw, ld.k.db, ld.v.dw, ld.v.db, ld.ao.dw, ld.ao.db = g[:8]
        ld.ln1.dgamma, ld.ln1.dbeta = g[8], g[9]
        ld.inter_l.dw, ld.inter_l.db, ld.out_l.dw, ld.out_l.db = g[10:14]
        ld.ln2.dgamma, ld.ln2.dbeta = g[14], g[15]
        gi += 16
    return d, store.views


def bert_hidden(model, input_ids, attention_mask):
    """last_hidden_state of `model` (hamspine.nn.BertModel) through the tower executor, or None when it does not apply"""
    if not (towers_enabled() and input_ids.is_cuda and input_ids.dim() == 2):
        return None
    if _has_hooks(model):
        return None
    return BertTowerFn.apply(input_ids, attention_mask, (model, model.training), *bert_params(model))
