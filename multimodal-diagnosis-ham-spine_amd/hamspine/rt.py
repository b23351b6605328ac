"""Runtime plumbing shared by the autograd functions: scratch arenas, dropout seeds, pointer helpers."""
import ctypes as C
import threading

import torch

from . import _lib as L

_tls = threading.local()
_ws = {}
_seed_lock = threading.Lock()
_seed_state = {"base": None, "ctr": 0}


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    """One growing scratch buffer per (device, thread, stream).  All kernels launched by one thread on one stream
    are ordered, so reuse between consecutive calls is safe; the autograd engine thread gets its own buffer so a
    forward running ahead on the main thread never shares scratch with a backward, and towers running concurrently
    on different streams never share scratch either."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), threading.get_ident(),
           torch.cuda.current_stream().cuda_stream)
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


def next_seed():
    """Dropout seeds: a counter offset by torch's seed, so torch.manual_seed makes runs repeatable."""
    with _seed_lock:
        if _seed_state["base"] is None:
            _seed_state["base"] = (torch.initial_seed() * 0x9E3779B1) & 0xFFFFFFFFFFFF
        _seed_state["ctr"] += 1
        return (_seed_state["base"] + _seed_state["ctr"] * 16) & 0xFFFFFFFFFFFFFFF


def reset_seed(base=None):
    with _seed_lock:
        _seed_state["base"] = base
        _seed_state["ctr"] = 0


# ------------------------------------------------------------------------------------------------------------------
# tower-level concurrency: the text tower runs on a side HIP stream next to the image tower (they are independent
# until the fusion operator), so the small ResNet kernels and the BERT GEMMs fill each other's gaps.  autograd runs
# each tower's backward on the stream its forward used and orders the junctions itself.
# ------------------------------------------------------------------------------------------------------------------
_tower_streams = {}


def towers_overlap_enabled():
    import os
    return os.environ.get("HAMSPINE_TOWER_OVERLAP", "1") != "0"


def tower_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _tower_streams.get(idx)
    if s is None:
        import os
        # HAMSPINE_TOWER_PRIORITY: queue priority of the text tower's stream (-1 = high: its workgroups are dispatched ahead of
        # the image tower's when both streams have work pending)
        s = torch.cuda.Stream(device=idx, priority=int(os.environ.get("HAMSPINE_TOWER_PRIORITY", "0")))
        _tower_streams[idx] = s
    return s


def side_streams():
    """streams hamspine may have work on besides the ambient one (collectives must order behind them too)"""
    return list(_tower_streams.values())


def run_on_tower_stream(fn, *inputs):
    """fn() on the side stream, ordered after everything enqueued so far on the ambient stream.  Returns (result, join);
    call join(*outputs) on the ambient stream before the first consumer of the outputs."""
    cur = torch.cuda.current_stream()
    side = tower_stream(cur.device)
    side.wait_stream(cur)
    for t in inputs:                       # inputs were allocated on the ambient stream: keep them alive for `side`
        if torch.is_tensor(t):
            t.record_stream(side)
    with torch.cuda.stream(side):
        out = fn()

    def join(*outs):
        cur.wait_stream(side)
        for t in outs:                     # allocated on `side`, consumed on the ambient stream
            if torch.is_tensor(t):
                t.record_stream(cur)
    return out, join


_grad_arena = {}
_arena_version = [0]


def grad_arena_register(param, view):
    """hamspine.ddp: parameter -> preallocated slot of a flat gradient bucket (same shape / strides)."""
    _grad_arena[param.data_ptr()] = view
    _arena_version[0] += 1


def grad_arena_clear():
    _grad_arena.clear()
    _arena_version[0] += 1


def arena_version():
    """changes whenever bucket slots are registered / cleared (cached tower descriptors hold gradient pointers)"""
    return _arena_version[0]


# Gradient milestones (hamspine.ddp <-> hamspine.tower).  A data-parallel wrapper that owns bucket slots registers itself
# here; a whole-tower backward asks, before its one C call, which of its parameters is the LAST one of each bucket to be
# produced, hands (gradient pointer, HIP event) pairs to the executor (hs_grad_milestones) and reports afterwards that
# the events have been recorded on its stream.
import weakref as _weakref

_milestone_providers = []


def add_milestone_provider(obj):
    """obj: has _milestones(params, stream) -> [(index, raw event handle, key)] and _milestones_recorded(keys)"""
    _milestone_providers.append(_weakref.ref(obj))


def grad_milestones(params, stream):
    """-> [(provider, [(index into params, raw HIP event handle, key)])] over the live data-parallel wrappers"""
    out = []
    live = []
    for r in _milestone_providers:
        o = r()
        if o is None:
            continue
        live.append(r)
        ms = o._milestones(params, stream)
        if ms:
            out.append((o, ms))
    _milestone_providers[:] = live
    return out


def grad_buffer_like(param):
    """Where a backward node writes d(param): the bucket slot when data-parallel training registered one
    (zero-copy bucketing), else a fresh tensor laid out like the parameter."""
    v = _grad_arena.get(param.data_ptr())
    if v is not None and v.shape == param.shape:
        return v.detach()      # fresh alias: autograd may adopt it as .grad without cloning
    return torch.empty_like(param)


def hs_dtype(t_or_dtype):
    dt = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if dt == torch.bfloat16:
        return L.HS_BF16
    if dt == torch.float32:
        return L.HS_F32
    raise L.HamspineError(f"unsupported dtype {dt}")


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise L.HamspineError(
                f"hamspine kernels only run on the HIP device (got a tensor on {t.device}); there is no CPU fallback")


def p(t, byte_offset=0):
    """raw device pointer (int) of a tensor, or None."""
    if t is None:
        return None
    return t.data_ptr() + byte_offset


def query(fn, desc):
    sv, ws = C.c_int64(0), C.c_int64(0)
    L.check(fn(C.byref(desc), C.byref(sv), C.byref(ws)), fn.__name__)
    return sv.value, ws.value


def as_cl(t, dtype):
    """NCHW-shaped tensor with NHWC memory of the compute dtype (no copy when it already is)."""
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous(memory_format=torch.channels_last)


def empty_cl(shape, dtype, device):
    return torch.empty(shape, dtype=dtype, device=device, memory_format=torch.channels_last)


def cast_weights(tensors, device):
    """f32 parameter tensors -> bf16 copies in one launch (memory order preserved)."""
    outs = [torch.empty(t.numel(), dtype=torch.bfloat16, device=device) for t in tensors]
    n = len(tensors)
    src = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    dst = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    cnt = (C.c_int64 * n)(*[t.numel() for t in tensors])
    L.check(L.lib().hs_cast_f32_to_bf16_multi(n, src, dst, cnt, stream()), "hs_cast_f32_to_bf16_multi")
    return outs


# ------------------------------------------------------------------------------------------------------------------
# bf16 weight shadows (bf16 mode): persistent bf16 copies of f32 weights that the towers read instead of casting every
# weight in every forward (33 cast launches, 0.8 GB per C2 step).  Who keeps a shadow equal to bf16(weight):
#   * the fused Adam / AdamW step writes it together with the f32 update (hs_adam_step_multi_shadow);
#   * every tower forward checks its weights' `_version` and address (ensure_shadows) and re-casts, in one launch, the
#     shadows of weights that anything else touched (load_state_dict, torch.optim, manual init: all bump `_version`);
#   * our other raw-pointer writer (FusedSGD) calls shadows_stale().
# The registry on the C side (hs_weight_shadow_set) is keyed by the weight's address.
# ON by default since round 3 (HAMSPINE_WEIGHT_SHADOWS=0 switches it off).  Measured on C2: round 2 12.05-12.11 ms with,
# 12.02-12.05 ms without (the casts hid inside an HBM-bound step tail); round 3, same box, three interleaved runs each:
# 10.93-10.99 ms with, 11.05-11.08 ms without; kernel time 13.93 vs 14.09 ms per step (33 -> 5 cast launches).
# ------------------------------------------------------------------------------------------------------------------
import weakref as _weakref

_shadows = {}          # id(param) -> _Shadow
_shadow_epoch = [0]    # bumped whenever an entry appears, moves or goes (optimizers cache shadow pointers against it)


class _Shadow:
    __slots__ = ("ref", "ptr", "version", "view", "stale", "__weakref__")


def shadows_enabled():
    import os
    return os.environ.get("HAMSPINE_WEIGHT_SHADOWS", "1") != "0"


def shadow_epoch():
    return _shadow_epoch[0]


def shadow_ptr_of(param):
    """device address of the registered bf16 shadow of `param`, or None"""
    rec = _shadows.get(id(param))
    if rec is None or rec.ref() is not param or rec.ptr != param.data_ptr():
        return None
    return rec.view.data_ptr()


def shadows_stale(params):
    """a raw-pointer writer changed these parameters without updating their shadows"""
    for p in params:
        rec = _shadows.get(id(p))
        if rec is not None and rec.ref() is p:
            rec.stale = True


_shadow_by_ptr = {}    # weight address -> id(param) of the entry registered for it


def _forget_shadow(key, ptr):
    """drop the entry `key` registered at weight address `ptr` (no-op when that address has since been re-registered)"""
    if _shadow_by_ptr.get(ptr) != key:
        return
    del _shadow_by_ptr[ptr]
    rec = _shadows.get(key)
    if rec is not None and rec.ptr == ptr:
        del _shadows[key]
    _shadow_epoch[0] += 1
    try:
        L.lib().hs_weight_shadow_set(ptr, None)
    except Exception:      # interpreter shutdown
        pass


def ensure_shadows(groups):
    """groups: lists of f32 parameters; the shadows of one group are consecutive segments of one bf16 buffer (the fused QKV
    weight of a BertLayer is read as one [3H][H] matrix).  Registers what is missing and re-casts what is out of date."""
    if not shadows_enabled():
        return
    stale_src, stale_dst = [], []
    for group in groups:
        recs = [_shadows.get(id(p)) for p in group]
        ok = all(r is not None and r.ref() is p and r.ptr == p.data_ptr() for r, p in zip(recs, group))
        if not ok:
            for p, r in zip(group, recs):
                if r is not None:
                    _forget_shadow(id(p), r.ptr)
            buf = torch.empty(sum(p.numel() for p in group), dtype=torch.bfloat16, device=group[0].device)
            off = 0
            for p in group:
                rec = _Shadow()
                rec.ref, rec.ptr, rec.version, rec.stale = _weakref.ref(p), p.data_ptr(), p._version, True
                rec.view = buf[off:off + p.numel()]
                off += p.numel()
                _shadows[id(p)] = rec
                _shadow_by_ptr[rec.ptr] = id(p)
                L.check(L.lib().hs_weight_shadow_set(rec.ptr, rec.view.data_ptr()), "hs_weight_shadow_set")
                _weakref.finalize(p, _forget_shadow, id(p), rec.ptr)
            _shadow_epoch[0] += 1
            recs = [_shadows[id(p)] for p in group]
        for p, r in zip(group, recs):
            if r.stale or r.version != p._version:
                stale_src.append(p)
                stale_dst.append(r.view)
                r.version, r.stale = p._version, False
    if stale_src:
        n = len(stale_src)
        src = (C.c_void_p * n)(*[t.data_ptr() for t in stale_src])
        dst = (C.c_void_p * n)(*[t.data_ptr() for t in stale_dst])
        cnt = (C.c_int64 * n)(*[t.numel() for t in stale_src])
        L.check(L.lib().hs_cast_f32_to_bf16_multi(n, src, dst, cnt, stream()), "hs_cast_f32_to_bf16_multi")


def clear_shadows():
    for key, rec in list(_shadows.items()):
        _forget_shadow(key, rec.ptr)
