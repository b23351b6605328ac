"""nn.Module shells with torch.nn's parameter names / init, whose forward runs the HIP kernels.

They subclass the torch.nn classes only to inherit construction, initialisation and state-dict
layout (the reference's checkpoints must load with strict=True); every forward is overridden.
"""
import torch
import torch.nn as nn

from .. import functional as F
from .. import rt


class Linear(nn.Linear):
    """torch.nn.Linear on the MFMA GEMM. `act`/`dropout_p`/`residual` are fused into the epilogue."""

    def forward(self, x, act=None, dropout_p=0.0, residual=None, out_dtype=None):
        return F.linear(x, self.weight, self.bias, act=act, dropout_p=dropout_p if self.training else 0.0,
                        residual=residual, out_dtype=out_dtype)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x, self.weight, self.bias, self.eps)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        import ctypes as C
        from .. import _lib as L
        rt.need_gpu(x)
        x = x.contiguous()
        seed = rt.next_seed()
        o = torch.empty_like(x)
        L.check(L.lib().hs_dropout(rt.hs_dtype(x), rt.p(x), rt.p(o), x.numel(), p, seed, rt.stream()), "hs_dropout")
        ctx.meta = (p, seed)
        return o

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        p, seed = ctx.meta
        g = g.contiguous()
        o = torch.empty_like(g)
        L.check(L.lib().hs_dropout(rt.hs_dtype(g), rt.p(g), rt.p(o), g.numel(), p, seed, rt.stream()), "hs_dropout")
        return o, None


class Dropout(nn.Dropout):
    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return DropoutFn.apply(x, float(self.p))


class MultiheadAttention(nn.MultiheadAttention):
    """torch.nn.MultiheadAttention(batch_first=True) with the same parameters / state-dict keys
    (in_proj_weight | q/k/v_proj_weight, in_proj_bias, out_proj.{weight,bias}); forward is one fused
    autograd node (hamspine.functional.MHAFn)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, batch_first=True, kdim=None, vdim=None, **kw):
        if not batch_first:
            raise ValueError("hamspine MultiheadAttention is batch_first only (as every reference call site)")
        super().__init__(embed_dim, num_heads, dropout=dropout, bias=bias, batch_first=True, kdim=kdim, vdim=vdim, **kw)

    def attend(self, query, key=None, valid_mask=None, residual=None):
        """query (B,Lq,E); key (B,Lk,kdim) or None for self-attention; valid_mask (B,Lk) with 1 = attend.
        Returns out_proj(attention) (+ residual)."""
        self_attn = key is None or key is query
        meta = {"heads": self.num_heads, "dropout": float(self.dropout) if self.training else 0.0, "self_attn": self_attn}
        if valid_mask is not None:
            valid_mask = valid_mask.contiguous()
            if valid_mask.dtype != torch.int64:
                valid_mask = valid_mask.long()
        packed = self._qkv_same_embed_dim
        return F.MHAFn.apply(
            query, None if self_attn else key, valid_mask, residual, meta,
            self.in_proj_weight if packed else None, self.in_proj_bias,
            None if packed else self.q_proj_weight, None if packed else self.k_proj_weight,
            None if packed else self.v_proj_weight, self.out_proj.weight, self.out_proj.bias)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None, **kw):
        if value is not key:
            raise NotImplementedError("hamspine MultiheadAttention needs value is key (true for every reference call)")
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is not used by the reference and is not implemented")
        valid = None
        if key_padding_mask is not None:
            valid = (~key_padding_mask.bool()).long()
        return self.attend(query, None if key is query else key, valid), None


class MLPHead(nn.Module):
    """nn.Sequential(Linear, ReLU, Dropout, Linear) of reference model.py:195-200 with the same
    state-dict keys ("0.weight", "0.bias", "3.weight", "3.bias"); runs as two fused GEMM nodes in f32."""

    def __init__(self, in_dim, hidden_dim, out_dim, dropout):
        super().__init__()
        self.add_module("0", Linear(in_dim, hidden_dim))
        self.add_module("1", nn.ReLU())
        self.add_module("2", nn.Dropout(dropout))
        self.add_module("3", Linear(hidden_dim, out_dim))

    def __getitem__(self, i):
        return getattr(self, str(i))

    def forward(self, x):
        if x.dtype != torch.float32:
            x = F.axpby(x, None, 1.0, 0.0, torch.float32)
        p = self[2].p if self.training else 0.0
        h = self[0](x, act="relu", dropout_p=p)
        return self[3](h)
