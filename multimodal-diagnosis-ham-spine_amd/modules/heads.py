"""Classifier heads (reference modules/heads.py) on the hamspine f32 GEMM path."""
import torch
import torch.nn as nn

from hamspine import functional as F
from hamspine.nn import LayerNorm, Linear, MultiheadAttention

GroupKANLinear = None   # the reference resolves this from the external `ikan` package (heads.py:7-25)


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
    """The reference builds this head from `ikan.GroupKAN.GroupKANLinear`, whose source is not in the reference
    tree (heads.py:7-25,108-140).  Same behaviour when the package is absent: ImportError."""
    if GroupKANLinear is None:
        raise ImportError("GroupKANLinear not found. Install the ikan package to use classifier_type='kan'.")
    if hidden_dim % num_groups != 0:
        raise ValueError(f"kan_num_groups ({num_groups}) must divide hidden_dim ({hidden_dim}).")
    raise ImportError("GroupKANLinear has no hamspine implementation")


__all__ = ["ResidualClassifier", "AttentionPoolingClassifier", "build_kan_head"]
