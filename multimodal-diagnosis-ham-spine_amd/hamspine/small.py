"""autograd nodes for the small f32 operators (csrc/small.hip): token select, concat, products,
sigmoid, softmax entropy, MP-Loss, focal loss, centre-crop resize.  All tensors here are f32 and of
size O(batch x hidden); each node is one kernel launch per direction.
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib as L
from . import rt

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
_declared = False


def _l():
    global _declared
    l = L.lib()
    if not _declared:
        l.hs_select_token_fwd.argtypes = [i32, vp, vp, i32, i32, i32, i32, vp]
        l.hs_select_token_bwd.argtypes = [i32, vp, vp, i32, i32, i32, i32, vp]
        l.hs_concat2.argtypes = [vp, i32, vp, i32, vp, i64, vp]
        l.hs_split2.argtypes = [vp, vp, i32, vp, i32, i64, vp]
        l.hs_mul.argtypes = [vp, vp, vp, i64, i32, i32, vp]
        l.hs_rowdot.argtypes = [vp, vp, vp, i32, i32, vp]
        l.hs_dot.argtypes = [vp, vp, vp, i64, vp]
        l.hs_sigmoid_fwd.argtypes = [vp, vp, i64, vp]
        l.hs_sigmoid_bwd.argtypes = [vp, vp, vp, i64, vp]
        l.hs_softmax_entropy.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        l.hs_mp_loss.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
        l.hs_focal_loss.argtypes = [vp, vp, vp, f32, i32, i32, vp, vp, vp]
        l.hs_center_crop_resize.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        l.hs_gelu_bwd.argtypes = [i32, vp, vp, vp, i64, vp]
        l.hs_mul_dev_scalar.argtypes = [vp, vp, vp, i64, vp]
        _declared = True
    return l


def _f32c(t):
    rt.need_gpu(t)
    if t.dtype != torch.float32:
        raise L.HamspineError(f"expected an f32 tensor, got {t.dtype}")
    return t.contiguous()


class SelectTokenFn(Function):
    """(B, Nt, H) any compute dtype -> (B, H) f32 = x[:, t, :]"""

    @staticmethod
    def forward(ctx, x, t):
        rt.need_gpu(x)
        x = x.contiguous()
        B, Nt, H = x.shape
        o = torch.empty((B, H), dtype=torch.float32, device=x.device)
        L.check(_l().hs_select_token_fwd(rt.hs_dtype(x), rt.p(x), rt.p(o), B, Nt, H, t, rt.stream()), "hs_select_token_fwd")
        ctx.meta = (x.shape, x.dtype, t)
        return o

    @staticmethod
    def backward(ctx, g):
        (B, Nt, H), dt, t = ctx.meta
        g = _f32c(g)
        dx = torch.empty((B, Nt, H), dtype=dt, device=g.device)
        L.check(_l().hs_select_token_bwd(rt.hs_dtype(dt), rt.p(g), rt.p(dx), B, Nt, H, t, rt.stream()), "hs_select_token_bwd")
        return dx, None


class Concat2Fn(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        rows = a.numel() // a.shape[-1]
        Ha, Hb = a.shape[-1], b.shape[-1]
        o = torch.empty(a.shape[:-1] + (Ha + Hb,), dtype=torch.float32, device=a.device)
        L.check(_l().hs_concat2(rt.p(a), Ha, rt.p(b), Hb, rt.p(o), rows, rt.stream()), "hs_concat2")
        ctx.meta = (a.shape, b.shape)
        return o

    @staticmethod
    def backward(ctx, g):
        sa, sb = ctx.meta
        g = _f32c(g)
        da = torch.empty(sa, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(sb, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        rows = g.numel() // g.shape[-1]
        L.check(_l().hs_split2(rt.p(g), rt.p(da), sa[-1], rt.p(db), sb[-1], rows, rt.stream()), "hs_split2")
        return da, db


def _mul_raw(a, b, mode, rows, cols):
    o = torch.empty_like(a)
    L.check(_l().hs_mul(rt.p(a), rt.p(b), rt.p(o), rows, cols, mode, rt.stream()), "hs_mul")
    return o


class MulFn(Function):
    """a * b for f32 tensors of the same shape."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        ctx.save_for_backward(a, b)
        return _mul_raw(a, b, 0, a.numel(), 1)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32c(g)
        return (_mul_raw(g, b, 0, g.numel(), 1) if ctx.needs_input_grad[0] else None,
                _mul_raw(g, a, 0, g.numel(), 1) if ctx.needs_input_grad[1] else None)


class ScaleBySigmoidFn(Function):
    """x * sigmoid(w) with a one-element parameter w (reference modules/fusion_blocks.py:199-201)."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _f32c(x), _f32c(w)
        s = torch.empty_like(w)
        L.check(_l().hs_sigmoid_fwd(rt.p(w), rt.p(s), 1, rt.stream()), "hs_sigmoid_fwd")
        ctx.save_for_backward(x, s)
        return _mul_raw(x, s, 2, x.numel(), 1)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = _f32c(g)
        dx = _mul_raw(g, s, 2, g.numel(), 1) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            d = torch.empty_like(s)
            L.check(_l().hs_dot(rt.p(g), rt.p(x), rt.p(d), g.numel(), rt.stream()), "hs_dot")
            dw = torch.empty_like(s)
            L.check(_l().hs_sigmoid_bwd(rt.p(d), rt.p(s), rt.p(dw), 1, rt.stream()), "hs_sigmoid_bwd")
        return dx, dw


class SigmoidFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        L.check(_l().hs_sigmoid_fwd(rt.p(x), rt.p(y), x.numel(), rt.stream()), "hs_sigmoid_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty_like(y)
        L.check(_l().hs_sigmoid_bwd(rt.p(g), rt.p(y), rt.p(dx), y.numel(), rt.stream()), "hs_sigmoid_bwd")
        return dx


class EntropyFn(Function):
    """(B, C) logits -> (B, 1) entropy of softmax (reference model.py:276-278)."""

    @staticmethod
    def forward(ctx, z):
        z = _f32c(z)
        B, Cn = z.shape
        e = torch.empty((B, 1), dtype=torch.float32, device=z.device)
        L.check(_l().hs_softmax_entropy(rt.p(z), None, rt.p(e), None, B, Cn, rt.stream()), "hs_softmax_entropy")
        ctx.save_for_backward(z)
        return e

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        g = _f32c(g)
        B, Cn = z.shape
        dz = torch.empty_like(z)
        L.check(_l().hs_softmax_entropy(rt.p(z), rt.p(g), None, rt.p(dz), B, Cn, rt.stream()), "hs_softmax_entropy")
        return dz


class GateMixFn(Function):
    """alpha * l_local + (1 - alpha) * l_context, alpha (B,1) (reference model.py:280-281)."""

    @staticmethod
    def forward(ctx, alpha, ll, lc):
        alpha, ll, lc = _f32c(alpha), _f32c(ll), _f32c(lc)
        B, Cn = ll.shape
        lib = _l()
        # out = lc + alpha*(ll - lc)
        diff = torch.empty_like(ll)
        L.check(lib.hs_axpby(L.HS_F32, L.HS_F32, rt.p(ll), rt.p(lc), rt.p(diff), ll.numel(), 1.0, -1.0, rt.stream()), "hs_axpby")
        t = _mul_raw(diff, alpha, 1, B, Cn)
        out = torch.empty_like(ll)
        L.check(lib.hs_axpby(L.HS_F32, L.HS_F32, rt.p(t), rt.p(lc), rt.p(out), ll.numel(), 1.0, 1.0, rt.stream()), "hs_axpby")
        ctx.save_for_backward(alpha, diff)
        return out

    @staticmethod
    def backward(ctx, g):
        alpha, diff = ctx.saved_tensors
        g = _f32c(g)
        B, Cn = g.shape
        lib = _l()
        dalpha = torch.empty_like(alpha)
        L.check(lib.hs_rowdot(rt.p(g), rt.p(diff), rt.p(dalpha), B, Cn, rt.stream()), "hs_rowdot")
        dll = _mul_raw(g, alpha, 1, B, Cn)
        dlc = torch.empty_like(g)
        L.check(lib.hs_axpby(L.HS_F32, L.HS_F32, rt.p(g), rt.p(dll), rt.p(dlc), g.numel(), 1.0, -1.0, rt.stream()), "hs_axpby")
        return dalpha, dll, dlc


class _LossWithGrads(Function):
    """shared tail: loss scalar saved with precomputed logit gradients, scaled by the incoming grad."""

    @staticmethod
    def _scale(grads, g):
        g = g.contiguous().float()
        outs = []
        for d in grads:
            o = torch.empty_like(d)
            L.check(_l().hs_mul_dev_scalar(rt.p(d), rt.p(g), rt.p(o), d.numel(), rt.stream()), "hs_mul_dev_scalar")
            outs.append(o)
        return outs


class MPLossFn(_LossWithGrads):
    @staticmethod
    def forward(ctx, zi, zt, zf, labels):
        zi, zt, zf = _f32c(zi), _f32c(zt), _f32c(zf)
        labels = labels.contiguous().long()
        B, Cn = zi.shape
        loss = torch.empty((), dtype=torch.float32, device=zi.device)
        di, dt_, df = torch.empty_like(zi), torch.empty_like(zt), torch.empty_like(zf)
        scratch = torch.empty(B, dtype=torch.float32, device=zi.device)
        L.check(_l().hs_mp_loss(rt.p(zi), rt.p(zt), rt.p(zf), rt.p(labels), B, Cn, rt.p(loss), rt.p(di), rt.p(dt_), rt.p(df),
                                rt.p(scratch), rt.stream()), "hs_mp_loss")
        ctx.save_for_backward(di, dt_, df)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b, c = _LossWithGrads._scale(ctx.saved_tensors, g)
        return a, b, c, None


class FocalLossFn(_LossWithGrads):
    @staticmethod
    def forward(ctx, z, labels, weight, gamma):
        z = _f32c(z)
        labels = labels.contiguous().long()
        B, Cn = z.shape
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        L.check(_l().hs_focal_loss(rt.p(z), rt.p(labels), rt.p(weight), gamma, B, Cn, rt.p(loss), rt.p(dz), rt.stream()),
                "hs_focal_loss")
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (a,) = _LossWithGrads._scale(ctx.saved_tensors, g)
        return a, None, None, None


class KLRowsFn(Function):
    """KL(p || q) over the last dim on probabilities clamped to [eps, 1] (reference mibf_net/attention.py:25-28)"""

    @staticmethod
    def forward(ctx, p, q, eps):
        p, q = _f32c(p), _f32c(q)
        cols = p.shape[-1]
        rows = p.numel() // cols
        o = torch.empty(p.shape[:-1], dtype=torch.float32, device=p.device)
        L.check(_l().hs_kl_rows(rt.p(p), rt.p(q), rt.p(o), None, None, None, rows, cols, eps, rt.stream()), "hs_kl_rows")
        ctx.save_for_backward(p, q)
        ctx.eps = eps
        return o

    @staticmethod
    def backward(ctx, g):
        p, q = ctx.saved_tensors
        cols = p.shape[-1]
        rows = p.numel() // cols
        g = _f32c(g)
        dp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        dq = torch.empty_like(q) if ctx.needs_input_grad[1] else None
        L.check(_l().hs_kl_rows(rt.p(p), rt.p(q), None, rt.p(g), rt.p(dp), rt.p(dq), rows, cols, ctx.eps, rt.stream()),
                "hs_kl_rows")
        return dp, dq, None


def kl_divergence(p, q, eps=1e-8):
    return KLRowsFn.apply(p, q, float(eps))


def select_token(x, t=0):
    return SelectTokenFn.apply(x, int(t))


class ConcatTokensFn(Function):
    """[a | b] along the last dim for activations of the compute dtype (bf16 or f32)"""

    @staticmethod
    def forward(ctx, a, b):
        rt.need_gpu(a, b)
        a, b = a.contiguous(), b.contiguous()
        if a.dtype != b.dtype:
            b = b.to(a.dtype)
        rows = a.numel() // a.shape[-1]
        Ha, Hb = a.shape[-1], b.shape[-1]
        o = torch.empty(a.shape[:-1] + (Ha + Hb,), dtype=a.dtype, device=a.device)
        L.check(_l().hs_concat2_t(rt.hs_dtype(a), rt.p(a), Ha, rt.p(b), Hb, rt.p(o), rows, rt.stream()), "hs_concat2_t")
        ctx.meta = (a.shape, b.shape, a.dtype)
        return o

    @staticmethod
    def backward(ctx, g):
        sa, sb, dt = ctx.meta
        g = g.contiguous()
        if g.dtype != dt:
            g = g.to(dt)
        da = torch.empty(sa, dtype=dt, device=g.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(sb, dtype=dt, device=g.device) if ctx.needs_input_grad[1] else None
        rows = g.numel() // g.shape[-1]
        L.check(_l().hs_split2_t(rt.hs_dtype(dt), rt.p(g), rt.p(da), sa[-1], rt.p(db), sb[-1], rows, rt.stream()), "hs_split2_t")
        return da, db


def concat_tokens(a, b):
    return ConcatTokensFn.apply(a, b)


def concat2(a, b):
    return Concat2Fn.apply(a, b)


def mul(a, b):
    return MulFn.apply(a, b)


def scale_by_sigmoid(x, w):
    return ScaleBySigmoidFn.apply(x, w)


def sigmoid(x):
    return SigmoidFn.apply(x)


def softmax_entropy(logits):
    return EntropyFn.apply(logits)


def gate_mix(alpha, logits_local, logits_context):
    return GateMixFn.apply(alpha, logits_local, logits_context)


def mp_loss(image_logits, text_logits, fused_logits, labels):
    return MPLossFn.apply(image_logits, text_logits, fused_logits, labels)


def focal_loss(logits, labels, weight=None, gamma=2.0):
    return FocalLossFn.apply(logits, labels, weight, float(gamma))


def center_crop_resize(x, ratio):
    """reference model.py:292-301: centre crop by `ratio`, bilinear resize back to (H, W); no grad."""
    rt.need_gpu(x)
    x = x.contiguous().float()
    N, Cc, H, W = x.shape
    ch, cw = max(1, int(H * ratio)), max(1, int(W * ratio))
    y0, x0 = max(0, (H - ch) // 2), max(0, (W - cw) // 2)
    if (ch, cw) == (H, W):
        return x
    o = torch.empty_like(x)
    L.check(_l().hs_center_crop_resize(rt.p(x), rt.p(o), N, Cc, H, W, y0, x0, ch, cw, rt.stream()), "hs_center_crop_resize")
    return o
