"""KAN layer, mixture-of-experts gating and SupCon loss on the HIP kernels (csrc/kan_moe.hip + hs_gemm, f32).

reference: ConNexT/models/block/kan1.py:77-165 (KANLinear.forward / b_splines), moe.py:171-291 (MoE),
scripts/train.py:23-44 (SupConLoss).
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib as L
from . import raw, rt

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
_declared = False


def _l():
    global _declared
    l = L.lib()
    if not _declared:
        l.hs_kan_features_fwd.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp]
        l.hs_kan_features_bwd.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
        l.hs_kan_pack_weight.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
        l.hs_kan_unpack_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
        l.hs_moe_gate_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, u64, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.hs_moe_dispatch_index.argtypes = [vp, i32, i32, vp, vp, vp]
        l.hs_rows_gather.argtypes = [vp, vp, vp, i32, i32, vp]
        l.hs_rows_scatter_add.argtypes = [vp, vp, vp, i32, i32, vp, i32, i32, vp]
        l.hs_rows_scatter_add_bwd.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp]
        l.hs_moe_gate_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
        l.hs_moe_combine_fwd.argtypes = [vp, C.POINTER(vp), vp, i32, i32, i32, vp]
        l.hs_moe_combine_bwd.argtypes = [vp, C.POINTER(vp), vp, C.POINTER(vp), vp, i32, i32, i32, vp]
        l.hs_supcon_loss.argtypes = [vp, vp, i32, i32, f32, vp, vp, vp, vp]
        l.hs_supcon_ws_bytes.argtypes = [i32, i32]
        l.hs_kan_regularization.argtypes = [vp, i64, i32, f32, f32, vp, vp, vp]
        l.hs_supcon_ws_bytes.restype = i64
        l.hs_mul_dev_scalar.argtypes = [vp, vp, vp, i64, vp]
        _declared = True
    return l


def _f32c(t):
    rt.need_gpu(t)
    if t.dtype != torch.float32:
        raise L.HamspineError(f"expected f32, got {t.dtype}")
    return t.contiguous()


BASE_ACTS = {"silu": 0, "swish": 0, "gelu": 1, "relu": 2, "identity": 3}   # hs_kan_features_* base_act codes


class KANLinearFn(Function):
    @staticmethod
    def forward(ctx, x, grid, base_w, spline_w, scaler, grid_size, order, act=0):
        x, base_w, spline_w = _f32c(x), _f32c(base_w), _f32c(spline_w)
        grid = _f32c(grid)
        scaler = _f32c(scaler) if scaler is not None else None
        B, in_f = x.shape
        out_f = base_w.shape[0]
        nb = grid_size + order
        Kc = in_f * (1 + nb)
        dev = x.device
        lib = _l()
        feat = torch.empty((B, Kc), dtype=torch.float32, device=dev)
        wcat = torch.empty((out_f, Kc), dtype=torch.float32, device=dev)
        L.check(lib.hs_kan_features_fwd(rt.p(x), rt.p(grid), rt.p(feat), B, in_f, grid_size, order, act, rt.stream()),
                "kan_features")
        L.check(lib.hs_kan_pack_weight(rt.p(base_w), rt.p(spline_w), rt.p(scaler), rt.p(wcat), out_f, in_f, nb, rt.stream()),
                "kan_pack_weight")
        y = torch.empty((B, out_f), dtype=torch.float32, device=dev)
        raw.gemm(feat, wcat, y, B, out_f, Kc, lda=Kc, ldb=Kc)
        ctx.save_for_backward(x, grid, spline_w, scaler, feat, wcat)
        ctx.meta = (grid_size, order, base_w.shape, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, grid, spline_w, scaler, feat, wcat = ctx.saved_tensors
        grid_size, order, base_shape, act = ctx.meta
        dy = _f32c(dy)
        B, in_f = x.shape
        out_f = base_shape[0]
        nb = grid_size + order
        Kc = in_f * (1 + nb)
        dev = x.device
        lib = _l()
        need_w = any(ctx.needs_input_grad[2:5])
        d_base = d_spline = d_scaler = None
        if need_w:
            dwcat = torch.empty((out_f, Kc), dtype=torch.float32, device=dev)
            split = raw.suggest_split(out_f, Kc, B, L.HS_F32)
            raw.gemm(dy, feat, dwcat, out_f, Kc, B, a_kind=L.A_RC, b_kind=L.B_RC, lda=out_f, ldb=Kc, split_k=split)
            d_base = torch.empty(base_shape, dtype=torch.float32, device=dev)
            d_spline = torch.empty_like(spline_w)
            d_scaler = torch.empty_like(scaler) if scaler is not None else None
            L.check(lib.hs_kan_unpack_wgrad(rt.p(dwcat), rt.p(spline_w), rt.p(scaler), rt.p(d_base), rt.p(d_spline),
                                            rt.p(d_scaler), out_f, in_f, nb, rt.stream()), "kan_unpack_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            dfeat = torch.empty((B, Kc), dtype=torch.float32, device=dev)
            raw.gemm(dy, wcat, dfeat, B, Kc, out_f, a_kind=L.A_KC, b_kind=L.B_RC, lda=out_f, ldb=Kc)
            dx = torch.empty_like(x)
            L.check(lib.hs_kan_features_bwd(rt.p(x), rt.p(grid), rt.p(dfeat), rt.p(dx), B, in_f, grid_size, order, act,
                                            rt.stream()), "kan_features_bwd")
        return dx, None, d_base, d_spline, d_scaler, None, None, None


def kan_linear(x, grid, base_weight, spline_weight, spline_scaler, grid_size, spline_order, base_act="silu"):
    return KANLinearFn.apply(x, grid, base_weight, spline_weight, spline_scaler, int(grid_size), int(spline_order),
                             BASE_ACTS[base_act])


class MoEGateFn(Function):
    """x -> (gates (B,E), aux loss) with w_gate / w_noise; noisy top-k in training, plain top-k in eval.  `noise`: optional
    (B,E) standard-normal draw (the reference's torch.randn_like(clean_logits), moe.py:247); None draws from the counter RNG."""

    @staticmethod
    def forward(ctx, x, w_gate, w_noise, k, noisy, coef, noise=None):
        x, w_gate, w_noise = _f32c(x), _f32c(w_gate), _f32c(w_noise)
        if noise is not None:
            noise = _f32c(noise)
            if tuple(noise.shape) != (x.shape[0], w_gate.shape[1]):
                raise ValueError(f"MoE gating noise must be (batch, experts) = {(x.shape[0], w_gate.shape[1])}, got {tuple(noise.shape)}")
        B, in_f = x.shape
        E = w_gate.shape[1]
        dev = x.device
        lib = _l()
        clean = torch.empty((B, E), dtype=torch.float32, device=dev)
        raw.gemm(x, w_gate, clean, B, E, in_f, a_kind=L.A_KC, b_kind=L.B_RC, lda=in_f, ldb=E)
        rawn = None
        if noisy:
            rawn = torch.empty((B, E), dtype=torch.float32, device=dev)
            raw.gemm(x, w_noise, rawn, B, E, in_f, a_kind=L.A_KC, b_kind=L.B_RC, lda=in_f, ldb=E)
        gates = torch.empty((B, E), dtype=torch.float32, device=dev)
        p = torch.empty_like(gates)
        z = torch.empty_like(gates)
        sigma = torch.empty_like(gates)
        loadrow = torch.empty_like(gates)
        top = torch.empty((B, 17), dtype=torch.int32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        d_imp = torch.empty(E, dtype=torch.float32, device=dev)
        d_load = torch.empty(E, dtype=torch.float32, device=dev)
        seed = rt.next_seed() if noisy else 0
        L.check(lib.hs_moe_gate_fwd(rt.p(clean), rt.p(rawn), rt.p(noise) if noisy else None, B, E, k, 1 if noisy else 0, 1e-2, seed, coef, rt.p(gates), rt.p(p),
                                    rt.p(top), rt.p(z), rt.p(sigma), rt.p(loadrow), rt.p(loss), rt.p(d_imp), rt.p(d_load),
                                    rt.stream()), "hs_moe_gate_fwd")
        ctx.save_for_backward(x, w_gate, w_noise, clean, rawn, p, top, z, sigma, d_imp, d_load)
        ctx.meta = (k, noisy)
        return gates, loss

    @staticmethod
    def backward(ctx, dgates, dloss):
        x, w_gate, w_noise, clean, rawn, p, top, z, sigma, d_imp, d_load = ctx.saved_tensors
        k, noisy = ctx.meta
        B, in_f = x.shape
        E = w_gate.shape[1]
        dev = x.device
        lib = _l()
        dgates = _f32c(dgates) if dgates is not None else torch.zeros((B, E), dtype=torch.float32, device=dev)
        gl = dloss.contiguous().float() if dloss is not None else None
        d_clean = torch.empty((B, E), dtype=torch.float32, device=dev)
        d_raw = torch.empty((B, E), dtype=torch.float32, device=dev) if noisy else None
        L.check(lib.hs_moe_gate_bwd(rt.p(clean), rt.p(rawn), rt.p(p), rt.p(top), rt.p(z), rt.p(sigma), rt.p(dgates), rt.p(gl),
                                    rt.p(d_imp), rt.p(d_load), B, E, k, 1 if noisy else 0, rt.p(d_clean), rt.p(d_raw),
                                    rt.stream()), "hs_moe_gate_bwd")
        # w_gate (in,E): d = x^T d_clean ; dx = d_clean w_gate^T (+ d_raw w_noise^T)
        d_wg = torch.empty_like(w_gate)
        raw.gemm(x, d_clean, d_wg, in_f, E, B, a_kind=L.A_RC, b_kind=L.B_RC, lda=in_f, ldb=E)
        d_wn = None
        if noisy:
            d_wn = torch.empty_like(w_noise)
            raw.gemm(x, d_raw, d_wn, in_f, E, B, a_kind=L.A_RC, b_kind=L.B_RC, lda=in_f, ldb=E)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            raw.gemm(d_clean, w_gate, dx, B, in_f, E, a_kind=L.A_KC, b_kind=L.B_KC, lda=E, ldb=E)
            if noisy:
                raw.gemm(d_raw, w_noise, dx, B, in_f, E, a_kind=L.A_KC, b_kind=L.B_KC, lda=E, ldb=E, accumulate=True)
        return dx, d_wg, d_wn, None, None, None, None


def moe_dispatch_index(gates):
    """-> (idx (E,B) int32: idx[e, :count[e]] = rows with gates[:, e] > 0 ascending, counts as a Python list).  The counts
    come back to the host -- the reference's SparseDispatcher does the same (`.tolist()`, moe.py:60): the experts' batch
    sizes are data-dependent launch shapes."""
    gates = _f32c(gates.detach())
    B, E = gates.shape
    idx = torch.empty((E, B), dtype=torch.int32, device=gates.device)
    cnt = torch.empty(E, dtype=torch.int32, device=gates.device)
    L.check(_l().hs_moe_dispatch_index(rt.p(gates), B, E, rt.p(idx), rt.p(cnt), rt.stream()), "hs_moe_dispatch_index")
    return idx, cnt.tolist()


class RowsGatherFn(Function):
    """x (B,D), idx (n,) int32 unique -> x[idx] (n,D); backward adds the rows' gradients back into a (B,D) zero tensor."""

    @staticmethod
    def forward(ctx, x, idx, n):
        x = _f32c(x)
        out = torch.empty((n, x.shape[1]), dtype=torch.float32, device=x.device)
        L.check(_l().hs_rows_gather(rt.p(x), rt.p(idx), rt.p(out), n, x.shape[1], rt.stream()), "hs_rows_gather")
        ctx.save_for_backward(idx)
        ctx.meta = (x.shape[0], n)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, n = ctx.meta
        dout = _f32c(dout)
        dx = torch.zeros((B, dout.shape[1]), dtype=torch.float32, device=dout.device)
        L.check(_l().hs_rows_scatter_add(rt.p(dx), rt.p(idx), None, 0, 0, rt.p(dout), n, dout.shape[1], rt.stream()),
                "hs_rows_scatter_add")
        return dx, None, None


class MoESparseCombineFn(Function):
    """y[b] = sum over the experts e that row b was dispatched to of gates[b, e] * out_e[position of b in e's batch]
    (SparseDispatcher.combine, moe.py:86-103: stitched * nonzero gates, index_add).  Experts are added in index order and a
    row occurs once per expert, so the sum is deterministic."""

    @staticmethod
    def forward(ctx, gates, idx, counts, B, *outs):
        gates = _f32c(gates)
        outs = [_f32c(o) for o in outs]
        O = outs[0].shape[1] if outs else 0
        for o in outs:
            if o.shape[0]:
                O = o.shape[1]
        y = torch.zeros((B, O), dtype=torch.float32, device=gates.device)
        E = gates.shape[1]
        lib = _l()
        for e, (o, n) in enumerate(zip(outs, counts)):
            if n:
                L.check(lib.hs_rows_scatter_add(rt.p(y), rt.p(idx[e]), rt.p(gates), E, e, rt.p(o), n, O, rt.stream()),
                        "hs_rows_scatter_add")
        ctx.save_for_backward(gates, idx, *outs)
        ctx.counts = counts
        return y

    @staticmethod
    def backward(ctx, dy):
        gates, idx, *outs = ctx.saved_tensors
        counts = ctx.counts
        dy = _f32c(dy)
        E = gates.shape[1]
        dgates = torch.zeros_like(gates)
        douts = []
        lib = _l()
        for e, (o, n) in enumerate(zip(outs, counts)):
            d = torch.empty_like(o)
            if n:
                L.check(lib.hs_rows_scatter_add_bwd(rt.p(dy), rt.p(idx[e]), rt.p(gates), E, e, rt.p(o), rt.p(d), rt.p(dgates), n,
                                                    o.shape[1], rt.stream()), "hs_rows_scatter_add_bwd")
            douts.append(d)
        return (dgates, None, None, None, *douts)


class MoECombineFn(Function):
    """y = sum_e gates[:, e, None] * out_e  (dense SparseDispatcher.combine)."""

    @staticmethod
    def forward(ctx, gates, *outs):
        gates = _f32c(gates)
        outs = [_f32c(o) for o in outs]
        B, E = gates.shape
        O = outs[0].shape[1]
        y = torch.empty((B, O), dtype=torch.float32, device=gates.device)
        ptrs = (vp * E)(*[o.data_ptr() for o in outs])
        L.check(_l().hs_moe_combine_fwd(rt.p(gates), ptrs, rt.p(y), B, E, O, rt.stream()), "hs_moe_combine_fwd")
        ctx.save_for_backward(gates, *outs)
        return y

    @staticmethod
    def backward(ctx, dy):
        gates, *outs = ctx.saved_tensors
        dy = _f32c(dy)
        B, E = gates.shape
        O = outs[0].shape[1]
        douts = [torch.empty_like(o) for o in outs]
        dgates = torch.empty_like(gates)
        ptrs = (vp * E)(*[o.data_ptr() for o in outs])
        dptrs = (vp * E)(*[o.data_ptr() for o in douts])
        L.check(_l().hs_moe_combine_bwd(rt.p(gates), ptrs, rt.p(dy), dptrs, rt.p(dgates), B, E, O, rt.stream()),
                "hs_moe_combine_bwd")
        return (dgates, *douts)


class SupConFn(Function):
    @staticmethod
    def forward(ctx, feat, labels, temperature):
        feat = _f32c(feat)
        labels = labels.contiguous().long()
        B, D = feat.shape
        lib = _l()
        ws = torch.empty(lib.hs_supcon_ws_bytes(B, D) // 4, dtype=torch.float32, device=feat.device)
        loss = torch.empty((), dtype=torch.float32, device=feat.device)
        df = torch.empty_like(feat)
        L.check(lib.hs_supcon_loss(rt.p(feat), rt.p(labels), B, D, temperature, rt.p(loss), rt.p(df), rt.p(ws), rt.stream()),
                "hs_supcon_loss")
        ctx.save_for_backward(df)
        return loss

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty_like(df)
        L.check(_l().hs_mul_dev_scalar(rt.p(df), rt.p(g), rt.p(out), df.numel(), rt.stream()), "hs_mul_dev_scalar")
        return out, None, None


def supcon_loss(features, labels, temperature=0.07):
    return SupConFn.apply(features, labels, float(temperature))


class KANRegularizationFn(Function):
    """KANLinear.regularization_loss on spline_weight (out, in, coeffs): scalar loss with its gradient (one kernel each)"""

    @staticmethod
    def forward(ctx, w, ra, re):
        rt.need_gpu(w)
        w = w.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=w.device)
        L.check(_l().hs_kan_regularization(rt.p(w), w.numel() // w.shape[-1], w.shape[-1], ra, re, rt.p(loss), None, rt.stream()),
                "hs_kan_regularization")
        ctx.save_for_backward(w)
        ctx.meta = (ra, re)
        return loss

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        ra, re = ctx.meta
        dw = torch.empty_like(w)
        L.check(_l().hs_kan_regularization(rt.p(w), w.numel() // w.shape[-1], w.shape[-1], ra, re, None, rt.p(dw), rt.stream()),
                "hs_kan_regularization")
        g = g.contiguous().float()
        out = torch.empty_like(dw)
        L.check(_l().hs_mul_dev_scalar(rt.p(dw), rt.p(g), rt.p(out), dw.numel(), rt.stream()), "hs_mul_dev_scalar")
        return out, None, None


def kan_regularization(spline_weight, regularize_activation=1.0, regularize_entropy=1.0):
    return KANRegularizationFn.apply(spline_weight, float(regularize_activation), float(regularize_entropy))
