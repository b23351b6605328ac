"""Classifier heads (reference modules/heads.py) on the hamspine f32 GEMM path."""
import torch
import torch.nn as nn

from hamspine import functional as F
from hamspine.nn import LayerNorm, Linear, MultiheadAttention
from hamspine.nn.layers import Dropout

from ConNexT.models.block.kan1 import KANLinear

_KAN_ACTS = {"gelu": nn.GELU, "silu": nn.SiLU, "swish": nn.SiLU, "relu": nn.ReLU, "identity": nn.Identity}


class GroupKANLinear(KANLinear):
    """Stand-in for `ikan.GroupKAN.GroupKANLinear`, which the reference imports from an external package whose source
    is not in the reference tree (heads.py:7-25), so its arithmetic cannot be restated or pinned.

    DEVIATION: the layer follows the reference's own in-tree KAN layer instead --
    `ConNexT/models/block/kan1.py:77-165` KANLinear (B-spline edges on a 5-interval grid, order 3, plus a base path
    `base_weight @ act(x)`) -- with the constructor surface `build_kan_head` uses (heads.py:125-139):
      * act_mode  -> the layer's `base_activation` (gelu | silu/swish | relu | identity),
      * drop      -> dropout on the layer input (train mode only),
      * num_groups-> validated (must divide in_features, as heads.py:119-122 does) and kept; the in-tree layer has one
                     spline per edge, so there is nothing for groups to share and the value does not change the result.
    Parameters / state-dict keys are KANLinear's (base_weight, spline_weight, spline_scaler, grid).  A checkpoint
    trained with the ikan layer cannot be loaded (different parameterisation)."""

    def __init__(self, in_features, out_features, act_mode="gelu", drop=0.0, num_groups=8, bias=True):
        if act_mode not in _KAN_ACTS:
            raise ValueError(f"act_mode must be one of {sorted(_KAN_ACTS)}, got {act_mode!r}")
        if in_features % num_groups != 0:
            raise ValueError(f"num_groups ({num_groups}) must divide in_features ({in_features}).")
        super().__init__(in_features, out_features, base_activation=_KAN_ACTS[act_mode])
        self.act_mode, self.num_groups = act_mode, num_groups
        self.drop = Dropout(drop)

    def forward(self, x):
        return super().forward(self.drop(_f32(x)))


def _f32(x):
    return x if x.dtype == torch.float32 else F.axpby(x, None, 1.0, 0.0, torch.float32)


class ResidualBlock(nn.Module):
    """LayerNorm(x + Linear(Dropout(ReLU(Linear(x))))) (heads.py:28-43)."""

    def __init__(self, hidden_dim, dropout=0.1):
        super().__init__()
        self.linear1 = Linear(hidden_dim, hidden_dim)
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(hidden_dim, hidden_dim)
        self.norm = LayerNorm(hidden_dim)

    def forward(self, x):
        h = self.linear1(x, act="relu", dropout_p=self.dropout.p if self.training else 0.0)
        return self.norm(self.linear2(h, residual=x))


class ResidualClassifier(nn.Module):
    def __init__(self, input_dim, hidden_dim, num_classes, dropout=0.1):
        super().__init__()
        self.project = Linear(input_dim, hidden_dim)
        self.res_block = ResidualBlock(hidden_dim, dropout)
        self.classifier = Linear(hidden_dim, num_classes)
        self.act = nn.ReLU()

    def forward(self, x):
        return self.classifier(self.res_block(self.project(_f32(x), act="relu")))


class AttentionPoolingClassifier(nn.Module):
    """The reference attends a learned query over a length-1 sequence (heads.py:61-105): the softmax is over one
    key, so the result is out_proj(v_proj(x)) whatever the query holds.  The module keeps `query` (state-dict
    key) and runs the real one-key attention node so gradients flow exactly as in the reference."""

    def __init__(self, input_dim, hidden_dim, num_classes, num_heads=4, dropout=0.1):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.query = nn.Parameter(torch.randn(1, 1, hidden_dim))
        self.attn = MultiheadAttention(hidden_dim, num_heads, dropout=dropout, batch_first=True)
        self.classifier = Linear(hidden_dim, num_classes)

    def forward(self, x):
        x = _f32(x)
        B = x.size(0)
        keys = x.unsqueeze(1)
        q = self.query.expand(B, -1, -1)
        pooled = self.attn.attend(q, key=keys)
        return self.classifier(pooled.reshape(B, self.hidden_dim))


def build_kan_head(hidden_dim, num_classes, dropout=0.1, num_groups=8, act_mode="gelu"):
    """GroupKANLinear(H, H) -> LayerNorm(H) -> GroupKANLinear(H, C), the reference's structure (heads.py:108-140)
    on the stand-in layer above (see its DEVIATION note)."""
    if hidden_dim % num_groups != 0:
        raise ValueError(f"kan_num_groups ({num_groups}) must divide hidden_dim ({hidden_dim}).")
    return nn.Sequential(
        GroupKANLinear(hidden_dim, hidden_dim, act_mode=act_mode, drop=dropout, num_groups=num_groups),
        LayerNorm(hidden_dim),
        GroupKANLinear(hidden_dim, num_classes, act_mode=act_mode, drop=0.0, num_groups=num_groups),
    )


__all__ = ["ResidualClassifier", "AttentionPoolingClassifier", "build_kan_head"]
