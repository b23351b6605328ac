"""Autograd nodes of the ConvNeXt tower and of the ConNeXT fusion (reference ConNexT/models/ourmodel.py):
depthwise 7x7 convolution, layer scale + residual, stride==kernel "patchify" convolutions (space-to-depth + GEMM)
and the attention core on projected q/k/v.  Activations are NHWC tensors of shape (N, H, W, C)."""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib as L
from . import rt
from .functional import _grad_like, linear


def _lib():
    return L.lib()


class DwConvFn(Function):
    """nn.Conv2d(C, C, k, padding=k//2, groups=C) on an NHWC activation; weight (C,1,k,k), bias (C,) f32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        rt.need_gpu(x, weight, bias)
        x = x.contiguous()
        N, H, W, Cc = x.shape
        k = weight.shape[-1]
        if weight.shape[0] != Cc or weight.shape[1] != 1:
            raise L.HamspineError(f"DwConvFn: weight {tuple(weight.shape)} is not a depthwise filter for C={Cc}")
        w = weight.contiguous()
        y = torch.empty_like(x)
        wsb = _lib().hs_dwconv_ws_bytes(N, H, W, Cc, k)
        ws = rt.workspace(wsb, x.device)
        L.check(_lib().hs_dwconv_fwd(rt.hs_dtype(x), rt.p(x), rt.p(w), rt.p(bias), rt.p(y), N, H, W, Cc, k, rt.p(ws),
                                     ws.numel(), rt.stream()), "hs_dwconv_fwd")
        ctx.save_for_backward(x, w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias = ctx.saved_tensors
        N, H, W, Cc = x.shape
        k = w.shape[-1]
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = _grad_like(w, ctx.needs_input_grad[1])
        db = _grad_like(bias, bias is not None and ctx.needs_input_grad[2] and dw is not None)
        wsb = _lib().hs_dwconv_ws_bytes(N, H, W, Cc, k)
        ws = rt.workspace(wsb, x.device)
        L.check(_lib().hs_dwconv_bwd(rt.hs_dtype(x), rt.p(x), rt.p(w), rt.p(dy), rt.p(dx), rt.p(dw), rt.p(db), N, H, W, Cc,
                                     k, rt.p(ws), ws.numel(), rt.stream()), "hs_dwconv_bwd")
        return dx, dw, db


def dwconv(x, weight, bias):
    return DwConvFn.apply(x, weight, bias)


class LayerScaleFn(Function):
    """res + gamma * rowscale[sample] * u  (ConvNextLayer tail; rowscale = stochastic-depth keep / (1-p) or None)."""

    @staticmethod
    def forward(ctx, u, gamma, res, rowscale):
        rt.need_gpu(u, gamma, res, rowscale)
        u = u.contiguous()
        res = res.contiguous()
        Cc = u.shape[-1]
        M = u.numel() // Cc
        rps = M // u.shape[0]
        g = gamma.reshape(-1)
        out = torch.empty_like(u)
        L.check(_lib().hs_layerscale_fwd(rt.hs_dtype(u), rt.p(u), rt.p(g), rt.p(rowscale), rps, rt.p(res), rt.p(out), M, Cc,
                                         rt.stream()), "hs_layerscale_fwd")
        ctx.save_for_backward(u, gamma, rowscale)
        return out

    @staticmethod
    def backward(ctx, dy):
        u, gamma, rowscale = ctx.saved_tensors
        Cc = u.shape[-1]
        M = u.numel() // Cc
        rps = M // u.shape[0]
        dy = dy.contiguous()
        if dy.dtype != u.dtype:
            dy = dy.to(u.dtype)
        du = torch.empty_like(u) if ctx.needs_input_grad[0] else None
        dg = _grad_like(gamma, True)
        wsb = _lib().hs_layerscale_ws_bytes(M, Cc)
        ws = rt.workspace(wsb, u.device)
        L.check(_lib().hs_layerscale_bwd(rt.hs_dtype(u), rt.p(dy), rt.p(u), rt.p(gamma.reshape(-1)), rt.p(rowscale), rps,
                                         rt.p(du), rt.p(dg), rt.p(ws), ws.numel(), M, Cc, rt.stream()), "hs_layerscale_bwd")
        return du, dg, (dy if ctx.needs_input_grad[2] else None), None


def layer_scale_residual(u, gamma, res, rowscale=None):
    return LayerScaleFn.apply(u, gamma, res, rowscale)


class PatchifyFn(Function):
    """(N,H,W,C) -> (N, H//k, W//k, k*k*C): the im2col of a stride==kernel convolution is a pure permutation."""

    @staticmethod
    def forward(ctx, x, k):
        rt.need_gpu(x)
        x = x.contiguous()
        N, H, W, Cc = x.shape
        P, Q = H // k, W // k
        out = torch.empty((N, P, Q, k * k * Cc), dtype=x.dtype, device=x.device)
        L.check(_lib().hs_patchify_fwd(rt.hs_dtype(x), rt.p(x), rt.p(out), N, H, W, Cc, k, k * k * Cc, rt.stream()),
                "hs_patchify_fwd")
        ctx.meta = (N, H, W, Cc, k)
        return out

    @staticmethod
    def backward(ctx, dp):
        N, H, W, Cc, k = ctx.meta
        dp = dp.contiguous()
        dx = torch.empty((N, H, W, Cc), dtype=dp.dtype, device=dp.device)
        L.check(_lib().hs_patchify_bwd(rt.hs_dtype(dp), rt.p(dp), rt.p(dx), N, H, W, Cc, k, k * k * Cc, rt.stream()),
                "hs_patchify_bwd")
        return dx, None


def patch_conv(x, weight, bias, k):
    """nn.Conv2d(Cin, Cout, k, stride=k) on NHWC x; weight (Cout, Cin, k, k) held channels_last, i.e. (Cout, k, k, Cin)
    in memory, so its GEMM view (Cout, k*k*Cin) is free."""
    w2 = weight.permute(0, 2, 3, 1)
    if not w2.is_contiguous():
        w2 = w2.contiguous()
    w2 = w2.reshape(weight.shape[0], -1)
    if (w2.shape[1] * (2 if x.dtype == torch.bfloat16 else 4)) % 16:
        raise NotImplementedError(f"patch_conv: k*k*Cin={w2.shape[1]} rows are not 16-byte multiples")
    return linear(PatchifyFn.apply(x, k), w2, bias)


class AttnCoreFn(Function):
    """softmax(scale * q k^T [+ key mask]) v on projected (B, L, heads*hd) tensors (no dropout)."""

    @staticmethod
    def _desc(q, k, heads, scale, key_mask=None, dropout_p=0.0, seed=0):
        B, Lq, D = q.shape
        d = L.AttnDesc()
        d.dtype = rt.hs_dtype(q)
        d.B, d.H, d.Lq, d.Lk, d.hd = B, heads, Lq, k.shape[1], D // heads
        d.q_bs, d.k_bs, d.v_bs, d.o_bs = Lq * D, k.shape[1] * D, k.shape[1] * D, Lq * D
        d.q_ld, d.k_ld, d.v_ld, d.o_ld = D, D, D, D
        d.scale = scale
        d.dropout_p = dropout_p
        d.seed = seed
        d.key_mask = rt.p(key_mask)
        return d

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, key_mask=None, dropout_p=0.0):
        rt.need_gpu(q, k, v, key_mask)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        if key_mask is not None:
            key_mask = key_mask.contiguous().long()
        seed = rt.next_seed() if dropout_p > 0 else 0
        d = AttnCoreFn._desc(q, k, heads, scale, key_mask, dropout_p, seed)
        sv_b, ws_b = rt.query(_lib().hs_attention_query, d)
        saved = torch.empty(sv_b, dtype=torch.uint8, device=q.device)
        ws = rt.workspace(ws_b, q.device)
        o = torch.empty_like(q)
        L.check(_lib().hs_attention_fwd(C.byref(d), rt.p(q), rt.p(k), rt.p(v), rt.p(o), rt.p(saved), sv_b, rt.p(ws),
                                        ws.numel(), rt.stream()), "hs_attention_fwd")
        ctx.save_for_backward(q, k, v, saved, key_mask)
        ctx.meta = (heads, scale, dropout_p, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, saved, key_mask = ctx.saved_tensors
        heads, scale, dropout_p, seed = ctx.meta
        do = do.contiguous()
        if do.dtype != q.dtype:
            do = do.to(q.dtype)
        d = AttnCoreFn._desc(q, k, heads, scale, key_mask, dropout_p, seed)
        sv_b, ws_b = rt.query(_lib().hs_attention_query, d)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = rt.workspace(ws_b, q.device)
        L.check(_lib().hs_attention_bwd(C.byref(d), rt.p(q), rt.p(k), rt.p(v), rt.p(do), rt.p(dq), rt.p(dk), rt.p(dv),
                                        rt.p(saved), saved.numel(), rt.p(ws), ws.numel(), rt.stream()), "hs_attention_bwd")
        return dq, dk, dv, None, None, None, None


def attention_core(q, k, v, heads=1, scale=1.0, key_mask=None, dropout_p=0.0):
    """key_mask: optional (B, Lk) integer tensor, 0 = masked key; dropout_p: attention-probability dropout"""
    return AttnCoreFn.apply(q, k, v, heads, float(scale), key_mask, float(dropout_p))
