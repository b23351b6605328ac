"""torch.autograd.Function wrappers around the composite executors and ops of the C ABI.

Each Function is one node of the autograd graph == one call into libhamspine_hip.so per direction.
Nothing in here computes with torch ops: tensors are only allocated (torch.empty) and handed over as
raw pointers.  Shapes at module boundaries follow the reference (NCHW-shaped image activations whose
memory is NHWC; (B, L, H) token tensors).
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib as L
from . import rt

CL = torch.channels_last


def _lib():
    return L.lib()


def _cl_weight(w):
    return w if w.is_contiguous(memory_format=CL) else w.contiguous(memory_format=CL)


def _fill_cb(cb, geo, w, gamma, beta, rm, rv, dw=None, dgamma=None, dbeta=None):
    cb.Cin, cb.Cout, cb.R, cb.stride, cb.pad = geo
    cb.w = rt.p(w)
    cb.gamma, cb.beta = rt.p(gamma), rt.p(beta)
    cb.running_mean, cb.running_var = rt.p(rm), rt.p(rv)
    cb.dw, cb.dgamma, cb.dbeta = rt.p(dw), rt.p(dgamma), rt.p(dbeta)


def _lin(in_f, out_f, w, b, dw=None, db=None, w_off=0, b_off=0):
    l = L.Linear()
    l.in_f, l.out_f = in_f, out_f
    l.w = rt.p(w, w_off * 4)
    l.b = rt.p(b, b_off * 4) if b is not None else None
    l.dw = rt.p(dw, w_off * 4) if dw is not None else None
    l.db = rt.p(db, b_off * 4) if db is not None else None
    return l


def _grad_like(t, needed):
    return rt.grad_buffer_like(t) if needed else None


# =================================================================================================
# ResNet stem / residual blocks
# =================================================================================================
class StemFn(Function):
    """conv7x7/2 + BN + ReLU + maxpool3x3/2 (reference encoder.py:63-68)."""

    @staticmethod
    def forward(ctx, image, weight, gamma, beta, cfg):
        rt.need_gpu(image, weight, gamma, beta)
        image = image.contiguous()
        if image.dtype != torch.float32:
            image = image.float()
        w = _cl_weight(weight)
        N, Cin, H, W = image.shape
        d = L.StemDesc()
        d.dtype = rt.hs_dtype(cfg["dtype"])
        d.N, d.H, d.W = N, H, W
        d.training = 1 if cfg["training"] else 0
        d.eps, d.momentum = cfg["eps"], cfg["momentum"]
        # no gradient will be asked for (inference under no_grad): BatchNorm folds into the convolution epilogue
        d.inference = 0 if (cfg["training"] or any(ctx.needs_input_grad)) else 1
        geo = (Cin, weight.shape[0], 7, 2, 3)
        _fill_cb(d.cb, geo, w, gamma, beta, cfg["running_mean"], cfg["running_var"])
        sv_b, ws_b = rt.query(_lib().hs_stem_query, d)
        saved = torch.empty(sv_b, dtype=torch.uint8, device=image.device)
        ws = rt.workspace(ws_b, image.device)
        P = (H + 6 - 7) // 2 + 1
        Q = (W + 6 - 7) // 2 + 1
        P2, Q2 = (P + 2 - 3) // 2 + 1, (Q + 2 - 3) // 2 + 1
        y = rt.empty_cl((N, weight.shape[0], P2, Q2), cfg["dtype"], image.device)
        L.check(_lib().hs_stem_fwd(C.byref(d), rt.p(image), rt.p(y), rt.p(saved), sv_b, rt.p(ws), ws.numel(), rt.stream()),
                "hs_stem_fwd")
        ctx.save_for_backward(image, w, gamma, beta, saved, y)
        ctx.cfg = cfg
        ctx.geo = geo
        return y

    @staticmethod
    def backward(ctx, dy):
        image, w, gamma, beta, saved, y = ctx.saved_tensors
        cfg = ctx.cfg
        dy = rt.as_cl(dy, cfg["dtype"])
        N, Cin, H, W = image.shape
        d = L.StemDesc()
        d.dtype = rt.hs_dtype(cfg["dtype"])
        d.N, d.H, d.W = N, H, W
        d.training = 1 if cfg["training"] else 0
        d.eps, d.momentum = cfg["eps"], cfg["momentum"]
        dw = _grad_like(w, ctx.needs_input_grad[1])
        dg = _grad_like(gamma, ctx.needs_input_grad[2])
        db = _grad_like(beta, ctx.needs_input_grad[3])
        _fill_cb(d.cb, ctx.geo, w, gamma, beta, None, None, dw, dg, db)
        sv_b, ws_b = rt.query(_lib().hs_stem_query, d)
        ws = rt.workspace(ws_b, image.device)
        L.check(_lib().hs_stem_bwd(C.byref(d), rt.p(y), rt.p(dy), rt.p(saved), saved.numel(), rt.p(ws), ws.numel(),
                                   rt.stream()), "hs_stem_bwd")
        return None, dw, dg, db, None


class ResBlockFn(Function):
    """torchvision BasicBlock / Bottleneck as one node (reference encoder.py:69-72, model_resnet.py:15).

    params = [w, gamma, beta] per main conv (2 or 3) followed by the optional downsample triple."""

    @staticmethod
    def _desc(cfg, x_shape, params, grads=None):
        d = L.ResblockDesc()
        d.dtype = rt.hs_dtype(cfg["dtype"])
        d.N, d.H, d.W = x_shape[0], x_shape[2], x_shape[3]
        d.training = 1 if cfg["training"] else 0
        d.eps, d.momentum = cfg["eps"], cfg["momentum"]
        stages = cfg["stages"]
        n_main = len(stages) - (1 if cfg["has_ds"] else 0)
        d.n_main = n_main
        d.has_ds = 1 if cfg["has_ds"] else 0
        for i, st in enumerate(stages):
            w, g, b = params[3 * i:3 * i + 3]
            gr = grads[3 * i:3 * i + 3] if grads is not None else (None, None, None)
            cb = d.main[i] if i < n_main else d.ds
            _fill_cb(cb, st["geo"], w, g, b, st["rm"] if grads is None else None, st["rv"] if grads is None else None, *gr)
        return d

    @staticmethod
    def forward(ctx, x, cfg, *params):
        rt.need_gpu(x, *params)
        x = rt.as_cl(x, cfg["dtype"])
        params = list(params)
        for i in range(0, len(params), 3):
            params[i] = _cl_weight(params[i])
        d = ResBlockFn._desc(cfg, x.shape, params)
        d.inference = 0 if (cfg["training"] or any(ctx.needs_input_grad)) else 1
        sv_b, ws_b = rt.query(_lib().hs_resblock_query, d)
        saved = torch.empty(sv_b, dtype=torch.uint8, device=x.device)
        ws = rt.workspace(ws_b, x.device)
        stages = cfg["stages"]
        n_main = len(stages) - (1 if cfg["has_ds"] else 0)
        H, W = x.shape[2], x.shape[3]
        for st in stages[:n_main]:
            _, _, R, s, pd = st["geo"]
            H, W = (H + 2 * pd - R) // s + 1, (W + 2 * pd - R) // s + 1
        y = rt.empty_cl((x.shape[0], stages[n_main - 1]["geo"][1], H, W), cfg["dtype"], x.device)
        L.check(_lib().hs_resblock_fwd(C.byref(d), rt.p(x), rt.p(y), rt.p(saved), sv_b, rt.p(ws), ws.numel(), rt.stream()),
                "hs_resblock_fwd")
        ctx.save_for_backward(x, y, saved, *params)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, saved, *params = ctx.saved_tensors
        cfg = ctx.cfg
        dy = rt.as_cl(dy, cfg["dtype"])
        grads = [_grad_like(t, ctx.needs_input_grad[2 + i]) for i, t in enumerate(params)]
        d = ResBlockFn._desc(cfg, x.shape, params, grads)
        sv_b, ws_b = rt.query(_lib().hs_resblock_query, d)
        ws = rt.workspace(ws_b, x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        L.check(_lib().hs_resblock_bwd(C.byref(d), rt.p(x), rt.p(y), rt.p(dy), rt.p(dx), rt.p(saved), saved.numel(),
                                       rt.p(ws), ws.numel(), rt.stream()), "hs_resblock_bwd")
        return (dx, None, *grads)


# =================================================================================================
# BERT
# =================================================================================================
class BertEmbedFn(Function):
    """BertEmbeddings (token_type_ids = 0): gather + LayerNorm + dropout."""

    @staticmethod
    def forward(ctx, ids, cfg, word, pos, typ, gamma, beta):
        rt.need_gpu(ids, word, pos, typ, gamma, beta)
        ids = ids.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.long()
        B, Lq = ids.shape
        H = word.shape[1]
        if Lq > pos.shape[0]:
            raise ValueError(f"sequence length {Lq} exceeds max_position_embeddings {pos.shape[0]}")
        dt = cfg["dtype"]
        dev = ids.device
        ssum = torch.empty((B, Lq, H), dtype=dt, device=dev)
        y = torch.empty((B, Lq, H), dtype=dt, device=dev)
        stats = torch.empty((2, B * Lq), dtype=torch.float32, device=dev)
        p = cfg["dropout"] if cfg["training"] else 0.0
        seed = rt.next_seed() if p > 0 else 0
        L.check(_lib().hs_bert_embed_fwd(rt.hs_dtype(dt), rt.p(ids), rt.p(word), rt.p(pos), rt.p(typ), rt.p(gamma),
                                         rt.p(beta), rt.p(ssum), rt.p(y), rt.p(stats), rt.p(stats, B * Lq * 4), B * Lq, Lq,
                                         H, word.shape[0], cfg["eps"], p, seed, rt.stream()), "hs_bert_embed_fwd")
        ctx.save_for_backward(ids, ssum, stats, gamma, word, pos, typ)
        ctx.meta = (p, seed, dt)
        ctx.pad_id = int(cfg.get("pad_id", -1))
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, ssum, stats, gamma, word, pos, typ = ctx.saved_tensors
        p, seed, dt = ctx.meta
        B, Lq, H = ssum.shape
        M = B * Lq
        dev = ids.device
        hdt = rt.hs_dtype(dt)
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        if p > 0:
            g = torch.empty_like(dy)
            L.check(_lib().hs_dropout(hdt, rt.p(dy), rt.p(g), M * H, p, seed, rt.stream()), "hs_dropout")
        else:
            g = dy
        dsum = torch.empty_like(ssum)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        wsb = _lib().hs_layernorm_bwd_ws_bytes(M, H)
        ws = rt.workspace(wsb + _lib().hs_colsum_ws_bytes(M, H) + 512, dev)
        L.check(_lib().hs_layernorm_bwd(hdt, rt.p(g), rt.p(ssum), rt.p(gamma), rt.p(stats), rt.p(stats, M * 4), rt.p(dsum),
                                        rt.p(dgamma), rt.p(dbeta), rt.p(ws), wsb, M, H, rt.stream()), "hs_layernorm_bwd")
        need_w, need_p, need_t = ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        dword = rt.grad_buffer_like(word).zero_() if need_w else None
        dpos = rt.grad_buffer_like(pos).zero_() if need_p else None
        dtyp = rt.grad_buffer_like(typ).zero_() if need_t else None
        if need_w or need_p:
            L.check(_lib().hs_bert_embed_bwd(hdt, rt.p(ids), rt.p(dsum), rt.p(dword), rt.p(dpos), B, Lq, H, word.shape[0],
                                             ctx.pad_id, rt.stream()), "hs_bert_embed_bwd")
        if need_t:
            csb = _lib().hs_colsum_ws_bytes(M, H)
            L.check(_lib().hs_colsum(hdt, rt.p(dsum), M, H, H, rt.p(dtyp), rt.p(ws), csb, 0, rt.stream()), "hs_colsum")
        return None, None, dword, dpos, dtyp, dgamma, dbeta


_BERT_LINS = ("q", "k", "v", "ao", "inter_l", "out_l")


class BertLayerFn(Function):
    """transformers BertLayer as one node.  params order:
    q.w q.b k.w k.b v.w v.b ao.w ao.b ln1.g ln1.b inter.w inter.b out.w out.b ln2.g ln2.b"""

    @staticmethod
    def _desc(cfg, x, mask, params, seed, grads=None):
        d = L.BertLayerDesc()
        B, Lq, H = x.shape
        d.dtype = rt.hs_dtype(cfg["dtype"])
        d.B, d.L, d.hidden, d.heads, d.inter = B, Lq, H, cfg["heads"], cfg["inter"]
        d.ln_eps = cfg["eps"]
        tr = cfg["training"]
        d.hidden_dropout = cfg["hidden_dropout"] if tr else 0.0
        d.attn_dropout = cfg["attn_dropout"] if tr else 0.0
        d.seed = seed
        d.attention_mask = rt.p(mask)
        g = grads if grads is not None else [None] * 16
        I = cfg["inter"]
        dims = {"q": (H, H), "k": (H, H), "v": (H, H), "ao": (H, H), "inter_l": (H, I), "out_l": (I, H)}
        idx = {"q": 0, "k": 2, "v": 4, "ao": 6, "inter_l": 10, "out_l": 12}
        for name in _BERT_LINS:
            i = idx[name]
            setattr(d, name, _lin(dims[name][0], dims[name][1], params[i], params[i + 1], g[i], g[i + 1]))
        for name, i in (("ln1", 8), ("ln2", 14)):
            n = L.Norm()
            n.gamma, n.beta = rt.p(params[i]), rt.p(params[i + 1])
            n.dgamma, n.dbeta = rt.p(g[i]), rt.p(g[i + 1])
            setattr(d, name, n)
        return d

    @staticmethod
    def forward(ctx, x, mask, cfg, *params):
        rt.need_gpu(x, mask, *params)
        x = x.contiguous()
        if x.dtype != cfg["dtype"]:
            x = x.to(cfg["dtype"])
        seed = rt.next_seed()
        d = BertLayerFn._desc(cfg, x, mask, params, seed)
        sv_b, ws_b = rt.query(_lib().hs_bert_layer_query, d)
        saved = torch.empty(sv_b, dtype=torch.uint8, device=x.device)
        ws = rt.workspace(ws_b, x.device)
        y = torch.empty_like(x)
        L.check(_lib().hs_bert_layer_fwd(C.byref(d), rt.p(x), rt.p(y), rt.p(saved), sv_b, rt.p(ws), ws.numel(), rt.stream()),
                "hs_bert_layer_fwd")
        ctx.save_for_backward(x, mask, saved, *params)
        ctx.cfg = dict(cfg)   # freeze training flag / dropout as used in the forward
        ctx.seed = seed
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, saved, *params = ctx.saved_tensors
        cfg = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != cfg["dtype"]:
            dy = dy.to(cfg["dtype"])
        grads = [_grad_like(t, ctx.needs_input_grad[3 + i]) for i, t in enumerate(params)]
        d = BertLayerFn._desc(cfg, x, mask, params, ctx.seed, grads)
        sv_b, ws_b = rt.query(_lib().hs_bert_layer_query, d)
        ws = rt.workspace(ws_b, x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        L.check(_lib().hs_bert_layer_bwd(C.byref(d), rt.p(x), rt.p(dy), rt.p(dx), rt.p(saved), saved.numel(), rt.p(ws),
                                         ws.numel(), rt.stream()), "hs_bert_layer_bwd")
        return (dx, None, None, *grads)


# =================================================================================================
# generic ops used by the fusion operators and heads
# =================================================================================================
_ACT = {None: L.ACT_NONE, "none": L.ACT_NONE, "relu": L.ACT_RELU, "gelu": L.ACT_GELU}


def _rows(x):
    return x.numel() // x.shape[-1]


class LinearFn(Function):
    """y = dropout(act(x W^T + b)) (+ residual).  x: (..., in); dtype of x selects the MFMA path;
    out_dtype may be f32 for a bf16 input (pooling / head boundary)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, dropout_p, residual, out_dtype):
        rt.need_gpu(x, weight, bias, residual)
        x = x.contiguous()
        out_f, in_f = weight.shape
        M = _rows(x)
        dt = x.dtype
        hdt = rt.hs_dtype(dt)
        out_dtype = out_dtype or dt
        w_lp = rt.cast_weights([weight], x.device)[0] if dt == torch.bfloat16 else None
        y = torch.empty(x.shape[:-1] + (out_f,), dtype=out_dtype, device=x.device)
        pre = torch.empty_like(y) if act == "gelu" else None
        if residual is not None:
            residual = residual.contiguous()
            if residual.dtype != out_dtype:
                residual = residual.to(out_dtype)
        seed = rt.next_seed() if dropout_p > 0 else 0
        lin = _lin(in_f, out_f, weight, bias)
        L.check(_lib().hs_linear_fwd(hdt, rt.p(x), M, in_f, C.byref(lin), rt.p(w_lp), rt.p(y), out_f, rt.hs_dtype(out_dtype),
                                     _ACT[act], rt.p(pre), rt.p(residual), out_f, dropout_p, seed, rt.stream()),
                "hs_linear_fwd")
        ctx.save_for_backward(x, weight, bias, w_lp, pre, y if act == "relu" else None)
        ctx.meta = (act, dropout_p, seed, out_dtype, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, w_lp, pre, yrelu = ctx.saved_tensors
        act, p, seed, out_dtype, has_res = ctx.meta
        out_f, in_f = weight.shape
        M = _rows(x)
        dt = x.dtype
        hdt = rt.hs_dtype(dt)
        dy = dy.contiguous()
        dres = dy if has_res and ctx.needs_input_grad[5] else None
        # gradient wrt the GEMM output, in the compute dtype of x
        g = dy
        if g.dtype != out_dtype:
            g = g.to(out_dtype)
        if out_dtype != dt:   # f32 boundary output of a bf16 GEMM: bring the gradient back to bf16
            g2 = torch.empty(g.shape, dtype=dt, device=g.device)
            L.check(_lib().hs_axpby(L.HS_F32, hdt, rt.p(g), None, rt.p(g2), g.numel(), 1.0, 0.0, rt.stream()), "hs_axpby")
            g = g2
        n = g.numel()
        if p > 0:
            g2 = torch.empty_like(g)
            L.check(_lib().hs_dropout(hdt, rt.p(g), rt.p(g2), n, p, seed, rt.stream()), "hs_dropout")
            g = g2
        if act == "relu":
            yr = yrelu if yrelu.dtype == dt else yrelu.to(dt)
            g2 = torch.empty_like(g)
            L.check(_lib().hs_relu_bwd(hdt, rt.p(g), rt.p(yr), rt.p(g2), n, rt.stream()), "hs_relu_bwd")
            g = g2
        elif act == "gelu":
            pr = pre if pre.dtype == dt else pre.to(dt)
            g2 = torch.empty_like(g)
            L.check(_lib().hs_gelu_bwd(hdt, rt.p(g), rt.p(pr), rt.p(g2), n, rt.stream()), "hs_gelu_bwd")
            g = g2
        dw = _grad_like(weight, ctx.needs_input_grad[1])
        db = _grad_like(bias, bias is not None and ctx.needs_input_grad[2])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        lin = _lin(in_f, out_f, weight, bias, dw, db)
        wsb = _lib().hs_linear_bwd_ws_bytes(M, in_f, out_f, hdt)
        ws = rt.workspace(wsb, x.device)
        L.check(_lib().hs_linear_bwd(hdt, rt.p(x), M, in_f, C.byref(lin), rt.p(w_lp), rt.p(g), out_f, rt.p(dx), in_f, hdt,
                                     L.MUL_NONE, None, 0, None, rt.p(ws), ws.numel(), rt.stream()), "hs_linear_bwd")
        return dx, dw, db, None, None, dres, None


def linear(x, weight, bias=None, act=None, dropout_p=0.0, residual=None, out_dtype=None):
    return LinearFn.apply(x, weight, bias, act, float(dropout_p), residual, out_dtype)


class MlpGeluFn(Function):
    """y = gelu(x W1^T + b1) W2^T + b2 as ONE node (the ConvNeXt block's pwconv1 -> GELU -> pwconv2, transformers
    ConvNextLayer / reference ConNexT/models/ourmodel.py:41-62).  Two Linear nodes computed the same values but ran GELU'
    as its own pass over the 4x-wide hidden gradient; here the second Linear's data-gradient GEMM applies GELU'(pre) in its
    epilogue (the BertLayer executor does the same), and both weight copies are cast in one launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        rt.need_gpu(x, w1, b1, w2, b2)
        x = x.contiguous()
        hid, in_f = w1.shape
        out_f = w2.shape[0]
        M = _rows(x)
        dt = x.dtype
        hdt = rt.hs_dtype(dt)
        w1_lp, w2_lp = rt.cast_weights([w1, w2], x.device) if dt == torch.bfloat16 else (None, None)
        pre = torch.empty(x.shape[:-1] + (hid,), dtype=dt, device=x.device)
        h = torch.empty_like(pre)
        y = torch.empty(x.shape[:-1] + (out_f,), dtype=dt, device=x.device)
        lin1, lin2 = _lin(in_f, hid, w1, b1), _lin(hid, out_f, w2, b2)
        L.check(_lib().hs_linear_fwd(hdt, rt.p(x), M, in_f, C.byref(lin1), rt.p(w1_lp), rt.p(h), hid, hdt, L.ACT_GELU,
                                     rt.p(pre), None, hid, 0.0, 0, rt.stream()), "hs_linear_fwd")
        L.check(_lib().hs_linear_fwd(hdt, rt.p(h), M, hid, C.byref(lin2), rt.p(w2_lp), rt.p(y), out_f, hdt, L.ACT_NONE,
                                     None, None, out_f, 0.0, 0, rt.stream()), "hs_linear_fwd")
        ctx.save_for_backward(x, w1, b1, w2, b2, w1_lp, w2_lp, pre, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, w1_lp, w2_lp, pre, h = ctx.saved_tensors
        hid, in_f = w1.shape
        out_f = w2.shape[0]
        M = _rows(x)
        dt = x.dtype
        hdt = rt.hs_dtype(dt)
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        need = ctx.needs_input_grad
        dw1, db1 = _grad_like(w1, need[1]), _grad_like(b1, b1 is not None and need[2])
        dw2, db2 = _grad_like(w2, need[3]), _grad_like(b2, b2 is not None and need[4])
        dpre = torch.empty_like(pre)                      # gradient wrt the hidden pre-activation
        dx = torch.empty_like(x) if need[0] else None
        lin2 = _lin(hid, out_f, w2, b2, dw2, db2)
        lin1 = _lin(in_f, hid, w1, b1, dw1, db1)
        # the two weight-gradient GEMMs run as one grouped grid (hs_wgrad_group_*): each call gets its own half of the workspace,
        # which stays alive until the group has been launched
        n2 = (int(_lib().hs_linear_bwd_ws_bytes(M, hid, out_f, hdt)) + 255) // 256 * 256
        n1 = int(_lib().hs_linear_bwd_ws_bytes(M, in_f, hid, hdt))
        ws = rt.workspace(n2 + n1, x.device)
        st = rt.stream()
        # grouped 256x128 tiles only pay when the two outputs give the chip enough of them (ConvNeXt-base's 512 <-> 2048 pair is
        # 64 tiles: 178 us grouped vs 2 x 62 us apart, measured); otherwise the two launches stay separate
        tiles = (-(-hid // 256)) * (-(-in_f // 128)) + (-(-out_f // 256)) * (-(-hid // 128))
        grouped = dt == torch.bfloat16 and tiles >= 192
        if grouped:
            L.check(_lib().hs_wgrad_group_begin(st), "hs_wgrad_group_begin")
        try:
            L.check(_lib().hs_linear_bwd(hdt, rt.p(h), M, hid, C.byref(lin2), rt.p(w2_lp), rt.p(dy), out_f, rt.p(dpre), hid, hdt,
                                         L.MUL_GELU_GRAD, rt.p(pre), hid, None, rt.p(ws), n2, st), "hs_linear_bwd")
            L.check(_lib().hs_linear_bwd(hdt, rt.p(x), M, in_f, C.byref(lin1), rt.p(w1_lp), rt.p(dpre), hid, rt.p(dx), in_f, hdt,
                                         L.MUL_NONE, None, 0, None, rt.p(ws, n2), n1, st), "hs_linear_bwd")
        finally:
            if grouped:
                L.check(_lib().hs_wgrad_group_end(st), "hs_wgrad_group_end")
        return dx, dw1, db1, dw2, db2


def mlp_gelu(x, w1, b1, w2, b2):
    return MlpGeluFn.apply(x, w1, b1, w2, b2)


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        rt.need_gpu(x, gamma, beta)
        x = x.contiguous()
        H = x.shape[-1]
        M = _rows(x)
        y = torch.empty_like(x)
        stats = torch.empty((2, M), dtype=torch.float32, device=x.device)
        L.check(_lib().hs_layernorm_fwd(rt.hs_dtype(x), rt.p(x), rt.p(gamma), rt.p(beta), rt.p(y), rt.p(stats),
                                        rt.p(stats, M * 4), M, H, eps, rt.stream()), "hs_layernorm_fwd")
        ctx.save_for_backward(x, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        H = x.shape[-1]
        M = _rows(x)
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dg = torch.empty_like(gamma)
        db = torch.empty_like(gamma)
        wsb = _lib().hs_layernorm_bwd_ws_bytes(M, H)
        ws = rt.workspace(wsb, x.device)
        L.check(_lib().hs_layernorm_bwd(rt.hs_dtype(x), rt.p(dy), rt.p(x), rt.p(gamma), rt.p(stats), rt.p(stats, M * 4),
                                        rt.p(dx), rt.p(dg), rt.p(db), rt.p(ws), wsb, M, H, rt.stream()), "hs_layernorm_bwd")
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, float(eps))


class MeanTokensFn(Function):
    """(B, Nt, H) -> (B, H) mean over tokens; output f32 (fusion/head boundary) or the input dtype."""

    @staticmethod
    def forward(ctx, x, out_f32):
        rt.need_gpu(x)
        x = x.contiguous()
        B, Nt, H = x.shape
        y = torch.empty((B, H), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
        L.check(_lib().hs_mean_tokens_fwd(rt.hs_dtype(x), rt.p(x), rt.p(y), B, Nt, H, 1 if out_f32 else 0, rt.stream()),
                "hs_mean_tokens_fwd")
        ctx.meta = (x.shape, x.dtype, out_f32)
        return y

    @staticmethod
    def backward(ctx, dy):
        (B, Nt, H), dt, out_f32 = ctx.meta
        dy = dy.contiguous()
        want = torch.float32 if out_f32 else dt
        if dy.dtype != want:
            dy = dy.to(want)
        dx = torch.empty((B, Nt, H), dtype=dt, device=dy.device)
        L.check(_lib().hs_mean_tokens_bwd(rt.hs_dtype(dt), rt.p(dy), rt.p(dx), B, Nt, H, 1 if out_f32 else 0, rt.stream()),
                "hs_mean_tokens_bwd")
        return dx, None


def mean_tokens(x, out_f32=True):
    return MeanTokensFn.apply(x, out_f32)


class AxpbyFn(Function):
    """out = a*x + b*y (same shapes); used for 0.5*(global+local) token mixing and small sums."""

    @staticmethod
    def forward(ctx, x, y, a, b, out_dtype):
        rt.need_gpu(x, y)
        x = x.contiguous()
        y = y.contiguous() if y is not None else None
        if y is not None and y.dtype != x.dtype:
            y = y.to(x.dtype)
        out_dtype = out_dtype or x.dtype
        o = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        L.check(_lib().hs_axpby(rt.hs_dtype(x), rt.hs_dtype(out_dtype), rt.p(x), rt.p(y), rt.p(o), x.numel(), a, b,
                                rt.stream()), "hs_axpby")
        ctx.meta = (a, b, x.dtype, y is not None)
        return o

    @staticmethod
    def backward(ctx, do):
        a, b, dt, has_y = ctx.meta
        do = do.contiguous()

        def scaled(s):
            g = torch.empty(do.shape, dtype=dt, device=do.device)
            L.check(_lib().hs_axpby(rt.hs_dtype(do), rt.hs_dtype(dt), rt.p(do), None, rt.p(g), do.numel(), s, 0.0, rt.stream()),
                    "hs_axpby")
            return g
        dx = scaled(a) if ctx.needs_input_grad[0] else None
        dyy = scaled(b) if has_y and ctx.needs_input_grad[1] else None
        return dx, dyy, None, None, None


def axpby(x, y, a=1.0, b=1.0, out_dtype=None):
    return AxpbyFn.apply(x, y, float(a), float(b), out_dtype)


class MHAFn(Function):
    """torch.nn.MultiheadAttention(batch_first=True) forward/backward as one node: input projections,
    attention core (key-padding mask, dropout on the probabilities), output projection, optional fused
    residual add.  Handles both the packed (in_proj_weight) and the split (q/k/v_proj_weight) layouts.
    (reference modules/fusion_blocks.py:18-31,48,62,107-125; modules/heads.py:86-103)"""

    @staticmethod
    def forward(ctx, query, key, mask, residual, meta, in_w, in_b, q_w, k_w, v_w, out_w, out_b):
        rt.need_gpu(query, key, mask, residual, in_w, in_b, q_w, k_w, v_w, out_w, out_b)
        heads, drop_p, self_attn = meta["heads"], meta["dropout"], meta["self_attn"]
        query = query.contiguous()
        key = query if self_attn else key.contiguous()
        dt = query.dtype
        hdt = rt.hs_dtype(dt)
        dev = query.device
        B, Lq, E = query.shape
        Lk, Ek = key.shape[1], key.shape[2]
        hd = E // heads
        packed = in_w is not None
        lp = {}
        if dt == torch.bfloat16:
            ws_list = [in_w, out_w] if packed else [q_w, k_w, v_w, out_w]
            outs = rt.cast_weights(ws_list, dev)
            lp = dict(zip(("in", "out") if packed else ("q", "k", "v", "out"), outs))

        def wlp(name, off_rows=0, cols=E):
            t = lp.get(name)
            return rt.p(t, off_rows * cols * 2) if t is not None else None

        lib = _lib()
        st = rt.stream()
        if packed and self_attn:
            # one GEMM: [B*L, E] x [3E, E]^T
            qkv = torch.empty((B, Lq, 3 * E), dtype=dt, device=dev)
            lin = _lin(E, 3 * E, in_w, in_b)
            L.check(lib.hs_linear_fwd(hdt, rt.p(query), B * Lq, E, C.byref(lin), wlp("in"), rt.p(qkv), 3 * E, hdt, 0, None,
                                      None, 0, 0.0, 0, st), "hs_linear_fwd")
            q_t, k_t, v_t = qkv, qkv, qkv
            q_off, k_off, v_off = 0, E, 2 * E
            q_ld = k_ld = v_ld = 3 * E
        else:
            q_t = torch.empty((B, Lq, E), dtype=dt, device=dev)
            kv = torch.empty((B, Lk, 2 * E), dtype=dt, device=dev)
            if packed:
                lq = _lin(E, E, in_w, in_b)
                L.check(lib.hs_linear_fwd(hdt, rt.p(query), B * Lq, E, C.byref(lq), wlp("in"), rt.p(q_t), E, hdt, 0, None, None,
                                          0, 0.0, 0, st), "hs_linear_fwd")
                lkv = _lin(E, 2 * E, in_w, in_b, w_off=E * E, b_off=E)
                L.check(lib.hs_linear_fwd(hdt, rt.p(key), B * Lk, Ek, C.byref(lkv), wlp("in", E, E), rt.p(kv), 2 * E, hdt, 0,
                                          None, None, 0, 0.0, 0, st), "hs_linear_fwd")
            else:
                lq = _lin(E, E, q_w, in_b)
                L.check(lib.hs_linear_fwd(hdt, rt.p(query), B * Lq, E, C.byref(lq), wlp("q"), rt.p(q_t), E, hdt, 0, None, None,
                                          0, 0.0, 0, st), "hs_linear_fwd")
                lk = _lin(Ek, E, k_w, in_b, b_off=E)
                L.check(lib.hs_linear_fwd(hdt, rt.p(key), B * Lk, Ek, C.byref(lk), wlp("k"), rt.p(kv), 2 * E, hdt, 0, None,
                                          None, 0, 0.0, 0, st), "hs_linear_fwd")
                lv = _lin(Ek, E, v_w, in_b, b_off=2 * E)
                L.check(lib.hs_linear_fwd(hdt, rt.p(key), B * Lk, Ek, C.byref(lv), wlp("v"), rt.p(kv, E * query.element_size()),
                                          2 * E, hdt, 0, None, None, 0, 0.0, 0, st), "hs_linear_fwd")
            k_t = v_t = kv
            q_off, k_off, v_off = 0, 0, E
            q_ld, k_ld, v_ld = E, 2 * E, 2 * E
        es = query.element_size()
        d = L.AttnDesc()
        d.dtype = hdt
        d.B, d.H, d.Lq, d.Lk, d.hd = B, heads, Lq, Lk, hd
        d.q_bs, d.k_bs, d.v_bs, d.o_bs = Lq * q_ld, Lk * k_ld, Lk * v_ld, Lq * E
        d.q_ld, d.k_ld, d.v_ld, d.o_ld = q_ld, k_ld, v_ld, E
        d.scale = 1.0 / (hd ** 0.5)
        d.dropout_p = drop_p
        seed = rt.next_seed() if drop_p > 0 else 0
        d.seed = seed
        d.key_mask = rt.p(mask)
        sv_b, ws_b = rt.query(lib.hs_attention_query, d)
        saved = torch.empty(sv_b, dtype=torch.uint8, device=dev)
        ws = rt.workspace(ws_b, dev)
        ctxv = torch.empty((B, Lq, E), dtype=dt, device=dev)
        L.check(lib.hs_attention_fwd(C.byref(d), rt.p(q_t, q_off * es), rt.p(k_t, k_off * es), rt.p(v_t, v_off * es),
                                     rt.p(ctxv), rt.p(saved), sv_b, rt.p(ws), ws.numel(), st), "hs_attention_fwd")
        y = torch.empty((B, Lq, E), dtype=dt, device=dev)
        if residual is not None:
            residual = residual.contiguous()
        lo = _lin(E, E, out_w, out_b)
        L.check(lib.hs_linear_fwd(hdt, rt.p(ctxv), B * Lq, E, C.byref(lo), wlp("out"), rt.p(y), E, hdt, 0, None,
                                  rt.p(residual), E, 0.0, 0, st), "hs_linear_fwd")
        keep = [t for t in (query, key if not self_attn else None, mask, q_t, k_t if k_t is not q_t else None, saved, ctxv)]
        ctx.save_for_backward(*keep, in_w, in_b, q_w, k_w, v_w, out_w, out_b, *[lp.get(n) for n in ("in", "q", "k", "v", "out")])
        ctx.meta = (meta, seed, packed, (q_off, k_off, v_off), (q_ld, k_ld, v_ld), residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (query, key, mask, q_t, k_t, saved, ctxv, in_w, in_b, q_w, k_w, v_w, out_w, out_b,
         lp_in, lp_q, lp_k, lp_v, lp_out) = ctx.saved_tensors
        meta, seed, packed, (q_off, k_off, v_off), (q_ld, k_ld, v_ld), has_res = ctx.meta
        heads, drop_p, self_attn = meta["heads"], meta["dropout"], meta["self_attn"]
        if key is None:
            key = query
        if k_t is None:
            k_t = q_t
        v_t = k_t
        dt = query.dtype
        hdt = rt.hs_dtype(dt)
        dev = query.device
        B, Lq, E = query.shape
        Lk, Ek = key.shape[1], key.shape[2]
        hd = E // heads
        es = query.element_size()
        lib = _lib()
        st = rt.stream()
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        nig = ctx.needs_input_grad
        d_in_w = _grad_like(in_w, packed and nig[5])
        d_in_b = _grad_like(in_b, in_b is not None and nig[6])
        d_q_w = _grad_like(q_w, (not packed) and nig[7])
        d_k_w = _grad_like(k_w, (not packed) and nig[8])
        d_v_w = _grad_like(v_w, (not packed) and nig[9])
        d_out_w = _grad_like(out_w, nig[10])
        d_out_b = _grad_like(out_b, out_b is not None and nig[11])

        def run_lin_bwd(x, M, ldx, lin, w_lp_ptr, g, ldg, dx, lddx, dx_res=None):
            wsb = lib.hs_linear_bwd_ws_bytes(M, lin.in_f, lin.out_f, hdt)
            ws = rt.workspace(wsb, dev)
            L.check(lib.hs_linear_bwd(hdt, rt.p(x), M, ldx, C.byref(lin), w_lp_ptr, g, ldg, rt.p(dx), lddx, hdt, L.MUL_NONE,
                                      None, 0, rt.p(dx_res), rt.p(ws), ws.numel(), st), "hs_linear_bwd")

        # output projection
        dctx = torch.empty_like(ctxv)
        lo = _lin(E, E, out_w, out_b, d_out_w, d_out_b)
        run_lin_bwd(ctxv, B * Lq, E, lo, rt.p(lp_out), rt.p(dy), E, dctx, E)
        # attention core
        d = L.AttnDesc()
        d.dtype = hdt
        d.B, d.H, d.Lq, d.Lk, d.hd = B, heads, Lq, Lk, hd
        d.q_bs, d.k_bs, d.v_bs, d.o_bs = Lq * q_ld, Lk * k_ld, Lk * v_ld, Lq * E
        d.q_ld, d.k_ld, d.v_ld, d.o_ld = q_ld, k_ld, v_ld, E
        d.scale = 1.0 / (hd ** 0.5)
        d.dropout_p = drop_p
        d.seed = seed
        d.key_mask = rt.p(mask)
        sv_b, ws_b = rt.query(lib.hs_attention_query, d)
        dq_t = torch.empty_like(q_t)
        dk_t = dq_t if k_t is q_t else torch.empty_like(k_t)
        ws = rt.workspace(ws_b, dev)
        L.check(lib.hs_attention_bwd(C.byref(d), rt.p(q_t, q_off * es), rt.p(k_t, k_off * es), rt.p(v_t, v_off * es),
                                     rt.p(dctx), rt.p(dq_t, q_off * es), rt.p(dk_t, k_off * es), rt.p(dk_t, v_off * es),
                                     rt.p(saved), saved.numel(), rt.p(ws), ws.numel(), st), "hs_attention_bwd")
        need_dq = nig[0]
        need_dk = (not self_attn) and nig[1]
        dquery = torch.empty_like(query) if need_dq else None
        dkey = torch.empty_like(key) if need_dk else None

        def lpp(t, off_rows=0, cols=E):
            return rt.p(t, off_rows * cols * 2) if t is not None else None

        if packed and self_attn:
            lin = _lin(E, 3 * E, in_w, in_b, d_in_w, d_in_b)
            run_lin_bwd(query, B * Lq, E, lin, lpp(lp_in), rt.p(dq_t), 3 * E, dquery, E)
        elif packed:
            lq = _lin(E, E, in_w, in_b, d_in_w, d_in_b)
            run_lin_bwd(query, B * Lq, E, lq, lpp(lp_in), rt.p(dq_t), E, dquery, E)
            lkv = _lin(E, 2 * E, in_w, in_b, d_in_w, d_in_b, w_off=E * E, b_off=E)
            run_lin_bwd(key, B * Lk, Ek, lkv, lpp(lp_in, E, E), rt.p(dk_t), 2 * E, dkey, Ek)
        else:
            lq = _lin(E, E, q_w, in_b, d_q_w, d_in_b)
            run_lin_bwd(query, B * Lq, E, lq, lpp(lp_q), rt.p(dq_t), E, dquery, E)
            lk = _lin(Ek, E, k_w, in_b, d_k_w, d_in_b, b_off=E)
            run_lin_bwd(key, B * Lk, Ek, lk, lpp(lp_k), rt.p(dk_t), 2 * E, dkey, Ek)
            # v projection: dkey accumulates (k and v both read `key`)
            lv = _lin(Ek, E, v_w, in_b, d_v_w, d_in_b, b_off=2 * E)
            if dkey is not None:
                dkey2 = torch.empty_like(dkey)
                wsb = lib.hs_linear_bwd_ws_bytes(B * Lk, Ek, E, hdt)
                ws2 = rt.workspace(wsb, dev)
                L.check(lib.hs_linear_bwd(hdt, rt.p(key), B * Lk, Ek, C.byref(lv), lpp(lp_v), rt.p(dk_t, E * es), 2 * E,
                                          rt.p(dkey2), Ek, hdt, L.MUL_NONE, None, 0, rt.p(dkey), rt.p(ws2), ws2.numel(), st),
                        "hs_linear_bwd")
                dkey = dkey2
            else:
                run_lin_bwd(key, B * Lk, Ek, lv, lpp(lp_v), rt.p(dk_t, E * es), 2 * E, None, Ek)
        dres = dy if has_res and nig[3] else None
        return dquery, dkey, None, dres, None, d_in_w, d_in_b, d_q_w, d_k_w, d_v_w, d_out_w, d_out_b


class CrossEntropyFn(Function):
    """mean CrossEntropy with class weights and label smoothing; loss and d(loss)/d(logits) in one kernel."""

    @staticmethod
    def forward(ctx, logits, labels, weight, smoothing):
        rt.need_gpu(logits, labels, weight)
        if logits.dtype != torch.float32:
            raise L.HamspineError("cross_entropy expects f32 logits")
        logits = logits.contiguous()
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        B, Cn = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dlog = torch.empty_like(logits)
        L.check(_lib().hs_cross_entropy(rt.p(logits), rt.p(labels), rt.p(weight), smoothing, B, Cn, rt.p(loss), rt.p(dlog),
                                        None, rt.stream()), "hs_cross_entropy")
        ctx.save_for_backward(dlog)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty_like(dlog)
        L.check(_lib().hs_mul_dev_scalar(rt.p(dlog), rt.p(g), rt.p(out), dlog.numel(), rt.stream()), "hs_mul_dev_scalar")
        return out, None, None, None


def cross_entropy(logits, labels, weight=None, label_smoothing=0.0):
    return CrossEntropyFn.apply(logits, labels, weight, float(label_smoothing))


class CrossAttnV2Fn(Function):
    """MIBF "IBFA" block, MultiHeadCrossAttention_v2 (reference mibf_net/attention.py:31-70) for the
    shapes the model uses (one token per modality): Q from x, keys/values = [x-token, y-token], softmax
    over the two keys, output projection.  f32 throughout.  params: wKx bKx wQx bQx wVx bVx wKy bKy wVy bVy wO bO"""

    @staticmethod
    def forward(ctx, x, y, heads, *params):
        rt.need_gpu(x, y, *params)
        x = x.contiguous()
        y = y.contiguous()
        if x.dtype != torch.float32 or y.dtype != torch.float32:
            raise L.HamspineError("CrossAttnV2Fn runs in f32 (post-tower features)")
        B, Sx, D = x.shape
        Sy = y.shape[1]
        if Sx != 1 or Sy != 1:
            raise NotImplementedError("MultiHeadCrossAttention_v2: only one token per modality is implemented "
                                      "(the only shape the reference model produces, model_resnet.py:40-56)")
        wKx, bKx, wQx, bQx, wVx, bVx, wKy, bKy, wVy, bVy, wO, bO = params
        dev = x.device
        lib = _lib()
        st = rt.stream()
        f32 = L.HS_F32
        q = torch.empty((B, 1, D), dtype=torch.float32, device=dev)
        kcat = torch.empty((B, 2, D), dtype=torch.float32, device=dev)
        vcat = torch.empty((B, 2, D), dtype=torch.float32, device=dev)

        def lin(inp, w, b, out, off, ld):
            l_ = _lin(D, D, w, b)
            L.check(lib.hs_linear_fwd(f32, rt.p(inp), B, D, C.byref(l_), None, rt.p(out, off * 4), ld, f32, 0, None, None, 0,
                                      0.0, 0, st), "hs_linear_fwd")
        lin(x, wQx, bQx, q, 0, D)
        lin(x, wKx, bKx, kcat, 0, 2 * D)
        lin(y, wKy, bKy, kcat, D, 2 * D)
        lin(x, wVx, bVx, vcat, 0, 2 * D)
        lin(y, wVy, bVy, vcat, D, 2 * D)
        hd = D // heads
        d = CrossAttnV2Fn._desc(B, heads, hd, D)
        sv_b, ws_b = rt.query(lib.hs_attention_query, d)
        saved = torch.empty(sv_b, dtype=torch.uint8, device=dev)
        ws = rt.workspace(ws_b, dev)
        o = torch.empty((B, 1, D), dtype=torch.float32, device=dev)
        L.check(lib.hs_attention_fwd(C.byref(d), rt.p(q), rt.p(kcat), rt.p(vcat), rt.p(o), rt.p(saved), sv_b, rt.p(ws),
                                     ws.numel(), st), "hs_attention_fwd")
        out = torch.empty((B, 1, D), dtype=torch.float32, device=dev)
        lin(o, wO, bO, out, 0, D)
        ctx.save_for_backward(x, y, q, kcat, vcat, o, saved, *params)
        ctx.heads = heads
        return out

    @staticmethod
    def _desc(B, heads, hd, D):
        d = L.AttnDesc()
        d.dtype = L.HS_F32
        d.B, d.H, d.Lq, d.Lk, d.hd = B, heads, 1, 2, hd
        d.q_bs, d.k_bs, d.v_bs, d.o_bs = D, 2 * D, 2 * D, D
        d.q_ld, d.k_ld, d.v_ld, d.o_ld = D, D, D, D
        d.scale = 1.0 / (hd ** 0.5)
        d.dropout_p = 0.0
        d.seed = 0
        d.key_mask = None
        return d

    @staticmethod
    def backward(ctx, g):
        x, y, q, kcat, vcat, o, saved, *params = ctx.saved_tensors
        wKx, bKx, wQx, bQx, wVx, bVx, wKy, bKy, wVy, bVy, wO, bO = params
        heads = ctx.heads
        B, _, D = x.shape
        dev = x.device
        lib = _lib()
        st = rt.stream()
        f32 = L.HS_F32
        g = g.contiguous().float()
        nig = ctx.needs_input_grad
        grads = [_grad_like(t, nig[3 + i]) for i, t in enumerate(params)]
        gKx, gbKx, gQx, gbQx, gVx, gbVx, gKy, gbKy, gVy, gbVy, gO, gbO = grads

        def lin_bwd(inp, w, b, dw, db, gout, off, ld, dx, res):
            l_ = _lin(D, D, w, b, dw, db)
            wsb = lib.hs_linear_bwd_ws_bytes(B, D, D, f32)
            ws = rt.workspace(wsb, dev)
            L.check(lib.hs_linear_bwd(f32, rt.p(inp), B, D, C.byref(l_), None, rt.p(gout, off * 4), ld, rt.p(dx), D, f32,
                                      L.MUL_NONE, None, 0, rt.p(res), rt.p(ws), ws.numel(), st), "hs_linear_bwd")
        do = torch.empty_like(o)
        lin_bwd(o, wO, bO, gO, gbO, g, 0, D, do, None)
        hd = D // heads
        d = CrossAttnV2Fn._desc(B, heads, hd, D)
        sv_b, ws_b = rt.query(lib.hs_attention_query, d)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(kcat), torch.empty_like(vcat)
        ws = rt.workspace(ws_b, dev)
        L.check(lib.hs_attention_bwd(C.byref(d), rt.p(q), rt.p(kcat), rt.p(vcat), rt.p(do), rt.p(dq), rt.p(dk), rt.p(dv),
                                     rt.p(saved), saved.numel(), rt.p(ws), ws.numel(), st), "hs_attention_bwd")
        need_x, need_y = nig[0], nig[1]
        dx1 = torch.empty_like(x) if need_x else None
        dx2 = torch.empty_like(x) if need_x else None
        dx3 = torch.empty_like(x) if need_x else None
        lin_bwd(x, wQx, bQx, gQx, gbQx, dq, 0, D, dx1, None)
        lin_bwd(x, wKx, bKx, gKx, gbKx, dk, 0, 2 * D, dx2, dx1)
        lin_bwd(x, wVx, bVx, gVx, gbVx, dv, 0, 2 * D, dx3, dx2)
        dy1 = torch.empty_like(y) if need_y else None
        dy2 = torch.empty_like(y) if need_y else None
        lin_bwd(y, wKy, bKy, gKy, gbKy, dk, D, 2 * D, dy1, None)
        lin_bwd(y, wVy, bVy, gVy, gbVy, dv, D, 2 * D, dy2, dy1)
        return (dx3, dy2, None, *grads)
