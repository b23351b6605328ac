"""Slice-sequence encoder (reference modules/sequence_blocks.py:6-70): (B, T, D) per-slice features -> (B, hidden).
`lstm` / `gru`: last time step of a (bi)directional recurrent stack, then a projection; `transformer`: sinusoidal
positions + post-LN encoder layers (ReLU FFN) + mean over slices.  Same constructor and state-dict keys as the
reference's torch.nn.LSTM / nn.GRU / nn.TransformerEncoder members; the arithmetic runs on libhamspine_hip.so in f32."""
import math

import torch
import torch.nn as nn

from hamspine import functional as F
from hamspine import rnn as R
from hamspine import small as S
from hamspine.nn import Dropout, LayerNorm, Linear, MultiheadAttention


class _RNNParams(nn.Module):
    """parameter holder with torch.nn.LSTM / nn.GRU names (weight_ih_l0, weight_hh_l0_reverse, ...) and init"""

    def __init__(self, gates, input_dim, hidden_dim, num_layers, bidirectional):
        super().__init__()
        self.hidden_size, self.num_layers, self.bidirectional = hidden_dim, num_layers, bidirectional
        k = 1.0 / math.sqrt(hidden_dim)
        for layer in range(num_layers):
            in_dim = input_dim if layer == 0 else hidden_dim * (2 if bidirectional else 1)
            for sfx in ("", "_reverse") if bidirectional else ("",):
                for name, shape in ((f"weight_ih_l{layer}{sfx}", (gates * hidden_dim, in_dim)),
                                    (f"weight_hh_l{layer}{sfx}", (gates * hidden_dim, hidden_dim)),
                                    (f"bias_ih_l{layer}{sfx}", (gates * hidden_dim,)),
                                    (f"bias_hh_l{layer}{sfx}", (gates * hidden_dim,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def direction(self, layer, reverse):
        sfx = "_reverse" if reverse else ""
        return tuple(getattr(self, f"{n}_l{layer}{sfx}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))


class _EncoderLayer(nn.Module):
    """torch.nn.TransformerEncoderLayer(batch_first=True, norm_first=False, activation=relu) parameter layout"""

    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model, eps=1e-5)
        self.norm2 = LayerNorm(d_model, eps=1e-5)
        self.dropout1 = Dropout(dropout)
        self.dropout2 = Dropout(dropout)

    def forward(self, x):
        x = self.norm1(F.axpby(x, self.dropout1(self.self_attn.attend(x)), 1.0, 1.0))
        ff = self.linear2(self.dropout(self.linear1(x, act="relu")))
        return self.norm2(F.axpby(x, self.dropout2(ff), 1.0, 1.0))


class _Encoder(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class SequenceEncoder(nn.Module):
    def __init__(self, input_dim, hidden_dim=256, encoder_type="lstm", num_layers=1, bidirectional=True,
                 dropout=0.1, num_heads=4):
        super().__init__()
        self.encoder_type = encoder_type.lower()
        self.hidden_dim = hidden_dim
        if self.encoder_type in ("lstm", "gru"):
            self.rnn = _RNNParams(4 if self.encoder_type == "lstm" else 3, input_dim, hidden_dim, num_layers, bidirectional)
            self._between = Dropout(dropout if num_layers > 1 else 0.0)
            output_dim = hidden_dim * (2 if bidirectional else 1)
            self.proj = Linear(output_dim, hidden_dim) if output_dim != hidden_dim else nn.Identity()
        elif self.encoder_type == "transformer":
            ff = max(hidden_dim * 4, input_dim * 2)
            self.encoder = _Encoder([_EncoderLayer(input_dim, num_heads, ff, dropout) for _ in range(num_layers)])
            self.proj = Linear(input_dim, hidden_dim) if input_dim != hidden_dim else nn.Identity()
            self._pe = {}
        else:
            raise ValueError(f"Unsupported sequence encoder type: {encoder_type}")

    def _positional_encoding(self, seq_len, dim, device):
        key = (seq_len, dim, str(device))
        if key not in self._pe:        # constants: built once on the host with the reference's formula
            position = torch.arange(seq_len).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
            pe = torch.zeros(seq_len, dim, dtype=torch.float32)
            pe[:, 0::2] = torch.sin(position * div_term)
            pe[:, 1::2] = torch.cos(position * div_term)
            self._pe[key] = pe.to(device)
        return self._pe[key]

    def forward(self, x):
        if x.dtype != torch.float32:
            x = F.axpby(x, None, 1.0, 0.0, torch.float32)
        if self.encoder_type in ("lstm", "gru"):
            p = self.rnn
            for layer in range(p.num_layers):
                fwd = R.run_direction(self.encoder_type, x, *p.direction(layer, False), reverse=False)
                bwd = R.run_direction(self.encoder_type, x, *p.direction(layer, True), reverse=True) if p.bidirectional else None
                if layer + 1 < p.num_layers:
                    x = self._between(R.stack_steps(fwd, bwd))
            last = S.concat2(fwd[-1], bwd[-1]) if bwd is not None else fwd[-1]    # out[:, -1, :]
            return self.proj(last)
        b, t, d = x.shape
        pe = self._positional_encoding(t, d, x.device)
        x = F.axpby(x, pe.unsqueeze(0).expand(b, t, d), 1.0, 1.0)
        for layer in self.encoder.layers:
            x = layer(x)
        return self.proj(F.mean_tokens(x))
