"""IBFA operators of MIBF-Net (reference mibf_net/attention.py) on the hamspine f32 kernels."""
import torch.nn as nn

from hamspine import functional as F
from hamspine.nn import Linear


class SelfAttention(nn.Module):
    """Spatial self-attention over a (B, C, H, W) map (reference attention.py:5-22).  MIBF-Net builds one
    (`I2Iattention`, model_resnet.py:21) and never calls it; the parameters exist for checkpoint fidelity."""

    def __init__(self, input_dim):
        super().__init__()
        self.query = Linear(input_dim, input_dim)
        self.key = Linear(input_dim, input_dim)
        self.value = Linear(input_dim, input_dim)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        raise NotImplementedError("SelfAttention is constructed but never evaluated by the reference model")


def compute_kl_divergence(p, q, eps=1e-8):
    """KL(p||q) on clamped probabilities (attention.py:25-28).  The training path uses the fused MP-Loss kernel
    (hamspine.small.mp_loss); this stand-alone form is kept for API parity and only accepts device tensors."""
    raise NotImplementedError("use hamspine.small.mp_loss (fused MP-Loss); the stand-alone KL is not exposed")


class MultiHeadCrossAttention_v2(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        if self.head_dim * num_heads != dim:
            raise ValueError("dim must be divisible by num_heads")
        self.toK_x = Linear(dim, dim)
        self.toQ_x = Linear(dim, dim)
        self.toV_x = Linear(dim, dim)
        self.toK_y = Linear(dim, dim)
        self.toV_y = Linear(dim, dim)
        self.to_out = Linear(dim, dim)

    def forward(self, x, y):
        mods = (self.toK_x, self.toQ_x, self.toV_x, self.toK_y, self.toV_y, self.to_out)
        params = [t for m in mods for t in (m.weight, m.bias)]
        return F.CrossAttnV2Fn.apply(x, y, self.num_heads, *params)
