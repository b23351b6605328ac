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
        s = torch.cuda.Stream(device=idx)
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
