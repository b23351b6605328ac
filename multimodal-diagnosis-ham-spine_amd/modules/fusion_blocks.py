"""Image/text fusion operators on the hamspine kernels.

API, parameter names and state-dict keys follow reference modules/fusion_blocks.py; every forward is
a short chain of fused autograd nodes (LayerNorm, MultiheadAttention incl. projections, GEMM+GELU,
token pooling).  Token tensors keep the compute dtype of the towers; pooled features leave as f32.
"""
import torch
import torch.nn as nn

from hamspine import functional as F
from hamspine import small as S
from hamspine.nn import LayerNorm, Linear, MultiheadAttention

_LEVELS = ("layer2", "layer3", "layer4")


class _FeedForward(nn.Module):
    """Linear - GELU - Dropout - Linear with nn.Sequential's key layout (0.*, 3.*)."""

    def __init__(self, dim, dropout):
        super().__init__()
        self.add_module("0", Linear(dim, dim * 4))
        self.add_module("1", nn.GELU())
        self.add_module("2", nn.Dropout(dropout))
        self.add_module("3", Linear(dim * 4, dim))

    def forward(self, x, residual=None):
        up, down, drop = getattr(self, "0"), getattr(self, "3"), getattr(self, "2")
        hidden = up(x, act="gelu", dropout_p=drop.p if self.training else 0.0)
        return down(hidden, residual=residual)


def _same_dtype(t, like):
    return t if t.dtype == like.dtype else t.to(like.dtype)


class BasicTransformerBlock(nn.Module):
    """pre-LN self-attention -> pre-LN cross-attention over the text context -> pre-LN GELU FFN,
    each with a residual connection (reference modules/fusion_blocks.py:7-71)."""

    def __init__(self, dim, context_dim, num_heads, dropout=0.1):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn1 = MultiheadAttention(dim, num_heads, dropout=dropout, batch_first=True)
        self.norm2 = LayerNorm(dim)
        self.attn2 = MultiheadAttention(dim, num_heads, dropout=dropout, batch_first=True,
                                        kdim=context_dim, vdim=context_dim)
        self.norm3 = LayerNorm(dim)
        self.ff = _FeedForward(dim, dropout)

    def forward(self, x, context, context_mask=None):
        context = _same_dtype(context, x)
        x = self.attn1.attend(self.norm1(x), residual=x)
        x = self.attn2.attend(self.norm2(x), key=context, valid_mask=context_mask, residual=x)
        return self.ff(self.norm3(x), residual=x)


class FusionModule(nn.Module):
    """BasicTransformerBlock followed by mean pooling over the image tokens (fusion_blocks.py:74-100)."""

    def __init__(self, text_dim, hidden_dim, num_heads=4, dropout=0.1):
        super().__init__()
        self.transformer_block = BasicTransformerBlock(hidden_dim, text_dim, num_heads, dropout)
        self.pool = nn.AdaptiveAvgPool1d(1)

    def forward(self, img_tokens, txt_tokens, txt_mask=None):
        return F.mean_tokens(self.transformer_block(img_tokens, txt_tokens, txt_mask), out_f32=True)


class CrossAttentionBlock(nn.Module):
    """LayerNorm(img + MHA(q=img, k=v=Linear(txt))) (fusion_blocks.py:103-128)."""

    def __init__(self, text_dim, hidden_dim, num_heads=4, dropout=0.1):
        super().__init__()
        self.txt_proj = Linear(text_dim, hidden_dim)
        self.attn = MultiheadAttention(hidden_dim, num_heads, dropout=dropout, batch_first=True)
        self.norm = LayerNorm(hidden_dim)

    def forward(self, img_tokens, txt_tokens, txt_mask=None):
        keys = self.txt_proj(_same_dtype(txt_tokens, img_tokens))
        return self.norm(self.attn.attend(img_tokens, key=keys, valid_mask=txt_mask, residual=img_tokens))


class MultiScaleFusionModule(nn.Module):
    """one CrossAttentionBlock per ResNet tap, pooled and averaged (fusion_blocks.py:131-160)."""

    def __init__(self, text_dim, hidden_dim, num_heads=4, dropout=0.1):
        super().__init__()
        self.cross_l2 = CrossAttentionBlock(text_dim, hidden_dim, num_heads, dropout)
        self.cross_l3 = CrossAttentionBlock(text_dim, hidden_dim, num_heads, dropout)
        self.cross_l4 = CrossAttentionBlock(text_dim, hidden_dim, num_heads, dropout)
        self.pool = nn.AdaptiveAvgPool1d(1)

    def forward(self, img_tokens, txt_tokens, txt_mask=None):
        blocks = (self.cross_l2, self.cross_l3, self.cross_l4)
        pooled = [F.mean_tokens(blk(img_tokens[k], txt_tokens, txt_mask), out_f32=True) for blk, k in zip(blocks, _LEVELS)]
        two = F.axpby(pooled[0], pooled[1], 1.0 / 3.0, 1.0 / 3.0)
        return F.axpby(two, pooled[2], 1.0, 1.0 / 3.0)


def pool_image_tokens(image_tokens):
    """mean over tokens; for the multi-scale dict the mean of the three per-level means (f32 out)."""
    if isinstance(image_tokens, dict):
        m = [F.mean_tokens(image_tokens[k], out_f32=True) for k in _LEVELS]
        return F.axpby(F.axpby(m[0], m[1], 1.0 / 3.0, 1.0 / 3.0), m[2], 1.0, 1.0 / 3.0)
    return F.mean_tokens(image_tokens, out_f32=True)


def pool_text_tokens(text_tokens, mode):
    return F.mean_tokens(text_tokens, out_f32=True) if mode == "mean" else S.select_token(text_tokens, 0)


class _PooledFusion(nn.Module):
    def __init__(self, text_pool):
        super().__init__()
        self.text_pool = text_pool

    def _pool_text(self, text_tokens):
        return pool_text_tokens(text_tokens, self.text_pool)

    def _pool_image(self, image_tokens):
        return pool_image_tokens(image_tokens)


class ConcatFusionModule(_PooledFusion):
    """Linear([mean(img) | pool(txt)]) (fusion_blocks.py:163-187)."""

    def __init__(self, text_dim, hidden_dim, text_pool="cls"):
        super().__init__(text_pool)
        self.proj = Linear(hidden_dim + text_dim, hidden_dim)

    def forward(self, image_tokens, text_tokens, txt_mask=None):
        return self.proj(S.concat2(self._pool_image(image_tokens), self._pool_text(text_tokens)))


class WeightedConcatFusionModule(ConcatFusionModule):
    """concat with learnable sigmoid scalars per modality (fusion_blocks.py:190-202)."""

    def __init__(self, text_dim, hidden_dim, text_pool="cls"):
        super().__init__(text_dim, hidden_dim, text_pool=text_pool)
        self.w_img = nn.Parameter(torch.zeros(1))
        self.w_txt = nn.Parameter(torch.zeros(1))

    def forward(self, image_tokens, text_tokens, txt_mask=None):
        img = S.scale_by_sigmoid(self._pool_image(image_tokens), self.w_img)
        txt = S.scale_by_sigmoid(self._pool_text(text_tokens), self.w_txt)
        return self.proj(S.concat2(img, txt))


class HadamardFusionModule(_PooledFusion):
    """LayerNorm(Linear(img) * Linear(txt)) (fusion_blocks.py:205-231)."""

    def __init__(self, text_dim, hidden_dim, text_pool="cls"):
        super().__init__(text_pool)
        self.img_proj = Linear(hidden_dim, hidden_dim)
        self.txt_proj = Linear(text_dim, hidden_dim)
        self.norm = LayerNorm(hidden_dim)

    def forward(self, image_tokens, text_tokens, txt_mask=None):
        prod = S.mul(self.img_proj(self._pool_image(image_tokens)), self.txt_proj(self._pool_text(text_tokens)))
        return self.norm(prod)


class BilinearFusionModule(_PooledFusion):
    """low-rank bilinear pooling: LayerNorm(Linear_r->H(Linear_H->r(img) * Linear_T->r(txt))) (fusion_blocks.py:234-261)."""

    def __init__(self, text_dim, hidden_dim, text_pool="cls", rank=128):
        super().__init__(text_pool)
        self.img_proj = Linear(hidden_dim, rank)
        self.txt_proj = Linear(text_dim, rank)
        self.out_proj = Linear(rank, hidden_dim)
        self.norm = LayerNorm(hidden_dim)

    def forward(self, image_tokens, text_tokens, txt_mask=None):
        prod = S.mul(self.img_proj(self._pool_image(image_tokens)), self.txt_proj(self._pool_text(text_tokens)))
        return self.norm(self.out_proj(prod))


class SSMFusionModule(nn.Module):
    """The reference needs the external CUDA package `mamba_ssm` here (fusion_blocks.py:264-272); it is
    not part of the reference tree, so this variant is out of scope and fails the same way."""

    def __init__(self, text_dim, hidden_dim, text_pool="cls"):
        super().__init__()
        raise ImportError("SSM/Mamba fusion requires `mamba-ssm`, which has no MI355X build in this framework.")


class VMambaFusionModule(nn.Module):
    """Needs the external `EnergeSnake` checkout in the reference (fusion_blocks.py:295-310); out of scope."""

    def __init__(self, text_dim, hidden_dim, text_pool="cls", vmamba_dim=32):
        super().__init__()
        raise ImportError("VMamba not found. The VMAMBA2Block dependency is external to the reference tree.")
