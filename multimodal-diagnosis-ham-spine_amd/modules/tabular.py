"""Tabular metadata encoder (reference modules/tabular.py:4-15): Linear - ReLU - Dropout - Linear."""
import torch
import torch.nn as nn

from hamspine.nn import Linear


class _Net(nn.Module):
    def __init__(self, input_dim, hidden_dim, dropout):
        super().__init__()
        self.add_module("0", Linear(input_dim, hidden_dim))
        self.add_module("1", nn.ReLU())
        self.add_module("2", nn.Dropout(dropout))
        self.add_module("3", Linear(hidden_dim, hidden_dim))

    def forward(self, x):
        drop = getattr(self, "2")
        h = getattr(self, "0")(x, act="relu", dropout_p=drop.p if self.training else 0.0)
        return getattr(self, "3")(h)


class TabularEncoder(nn.Module):
    def __init__(self, input_dim, hidden_dim=128, dropout=0.1):
        super().__init__()
        self.net = _Net(input_dim, hidden_dim, dropout)

    def forward(self, x):
        if x.dtype != torch.float32:
            x = x.float()
        return self.net(x)
