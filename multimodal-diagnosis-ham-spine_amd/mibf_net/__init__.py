"""MIBF-Net (reference mibf_net/): ResNet50 + BERT-CLS + IBFA bidirectional cross-attention + MP-Loss."""
