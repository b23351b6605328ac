"""Dual-expert gate (reference modules/gating.py:5-23): sigmoid(MLP([lesion | context | entropy]))."""
import torch.nn as nn

from hamspine import small as S
from hamspine.nn import Linear


class _GateMLP(nn.Module):
    def __init__(self, in_dim, hidden_dim):
        super().__init__()
        self.add_module("0", Linear(in_dim, hidden_dim))
        self.add_module("1", nn.ReLU())
        self.add_module("2", Linear(hidden_dim, 1))

    def forward(self, x):
        return getattr(self, "2")(getattr(self, "0")(x, act="relu"))


class DualExpertGate(nn.Module):
    def __init__(self, lesion_dim, context_dim, hidden_dim=128, use_entropy=True):
        super().__init__()
        self.use_entropy = use_entropy
        self.fc = _GateMLP(lesion_dim + context_dim + (1 if use_entropy else 0), hidden_dim)

    def forward(self, lesion_feat, context_feat, entropy=None):
        gate_in = S.concat2(lesion_feat, context_feat)
        if self.use_entropy:
            if entropy is None:
                raise ValueError("entropy is required when use_entropy=True")
            gate_in = S.concat2(gate_in, entropy)
        return S.sigmoid(self.fc(gate_in))
