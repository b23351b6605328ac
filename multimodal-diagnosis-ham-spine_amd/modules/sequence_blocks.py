"""Slice-sequence encoder (reference modules/sequence_blocks.py).  The LSTM/GRU/TransformerEncoder
variants are not on any benchmarked configuration and are outside this round's hot-path scope
("LSTM/TransformerEncoder variants OUT OF SCOPE for first pass")."""
import torch.nn as nn


class SequenceEncoder(nn.Module):
    def __init__(self, input_dim, hidden_dim=256, encoder_type="lstm", num_layers=1, bidirectional=True,
                 dropout=0.1, num_heads=4):
        super().__init__()
        raise NotImplementedError(
            "SequenceEncoder (2.5-D slice sequences) has no HIP implementation yet; "
            "set model.sequence_encoder.enabled=false")
