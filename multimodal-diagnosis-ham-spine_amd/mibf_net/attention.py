"""IBFA operators of MIBF-Net (reference mibf_net/attention.py) on the hamspine f32 kernels."""
import torch.nn as nn

from hamspine import convnext_ops as X
from hamspine import functional as F
from hamspine import small as S
from hamspine.nn import Linear


class SelfAttention(nn.Module):
    """Spatial self-attention over a (B, C, H, W) map (reference attention.py:5-22): softmax(Q K^T / sqrt(C)) V over the
    H*W positions.  MIBF-Net builds one (`I2Iattention`, model_resnet.py:21) and never calls it."""

    def __init__(self, input_dim):
        super().__init__()
        self.query = Linear(input_dim, input_dim)
        self.key = Linear(input_dim, input_dim)
        self.value = Linear(input_dim, input_dim)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        b, c, hh, ww = x.shape
        tokens = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)        # (B, HW, C); free for channels_last memory
        if not tokens.is_contiguous():
            tokens = tokens.contiguous()
        out = X.attention_core(self.query(tokens), self.key(tokens), self.value(tokens), heads=1, scale=1.0 / (c ** 0.5))
        return out.reshape(b, hh, ww, c).permute(0, 3, 1, 2)         # (B, C, H, W), channels_last memory


def compute_kl_divergence(p, q, eps=1e-8):
    """KL(p||q) over the last dim on clamped probabilities (attention.py:25-28).  The training path uses the fused
    MP-Loss kernel (hamspine.small.mp_loss); this is the stand-alone operator with the same contract."""
    return S.kl_divergence(p, q, eps)


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
        if x.shape[1] == 1 and y.shape[1] == 1:
            # the shape MIBF-Net produces (model_resnet.py:40-56): one fused node writes the projections straight into the
            # concatenated key / value buffers
            mods = (self.toK_x, self.toQ_x, self.toV_x, self.toK_y, self.toV_y, self.to_out)
            params = [t for m in mods for t in (m.weight, m.bias)]
            return F.CrossAttnV2Fn.apply(x, y, self.num_heads, *params)
        # general token counts: keys / values of x and y concatenated along the token axis (attention.py:60-70)
        b, sx, d = x.shape
        sy = y.shape[1]

        def cat_tokens(a, c):      # (B, Sx, D) | (B, Sy, D) -> (B, Sx + Sy, D): a last-dim concat of the flattened rows
            return S.concat2(a.reshape(b, sx * d), c.reshape(b, sy * d)).reshape(b, sx + sy, d)
        kcat = cat_tokens(self.toK_x(x), self.toK_y(y))
        vcat = cat_tokens(self.toV_x(x), self.toV_y(y))
        out = X.attention_core(self.toQ_x(x), kcat, vcat, heads=self.num_heads, scale=1.0 / (self.head_dim ** 0.5))
        return self.to_out(out)
