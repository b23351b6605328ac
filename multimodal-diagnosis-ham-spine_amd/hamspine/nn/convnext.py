"""ConvNeXt image tower with the module tree and state-dict keys of transformers.ConvNextModel (and, as
`convnext_features`, of torchvision's `convnext_*().features`), executed by the HIP kernels: patchify convolutions as
space-to-depth + MFMA GEMM, depthwise 7x7 / LayerNorm / layer-scale as NHWC vector kernels, pointwise convolutions
on the MFMA GEMM with bias+GELU in the epilogue.

Follows transformers.models.convnext.modeling_convnext (ConvNextEmbeddings, ConvNextLayer, ConvNextStage,
ConvNextEncoder, ConvNextModel) as used by reference ConNexT/models/ourmodel.py:41-47,75-78, and the public
torchvision.models.convnext definition for the fallback branch ourmodel.py:49-62.  Neither library is a dependency.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

import hamspine

from .. import convnext_ops as X
from .. import functional as F
from .. import rt
from .layers import Linear

CL = torch.channels_last


class ConvNextConfig:
    def __init__(self, num_channels=3, patch_size=4, num_stages=4, hidden_sizes=None, depths=None, hidden_act="gelu",
                 initializer_range=0.02, layer_norm_eps=1e-12, layer_scale_init_value=1e-6, drop_path_rate=0.0, **unused):
        if hidden_act != "gelu":
            raise ValueError(f"hidden_act={hidden_act!r}: only erf-GELU (ConvNeXt's default) is implemented")
        self.num_channels = num_channels
        self.patch_size = patch_size
        self.hidden_sizes = list(hidden_sizes or [96, 192, 384, 768])
        self.depths = list(depths or [3, 3, 9, 3])
        self.num_stages = len(self.hidden_sizes)
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.layer_scale_init_value = layer_scale_init_value
        self.drop_path_rate = drop_path_rate

    @classmethod
    def base(cls, **kw):
        """facebook/convnext-base-224: the tower the reference loads (ourmodel.py:36)."""
        return cls(hidden_sizes=[128, 256, 512, 1024], depths=[3, 3, 27, 3], **kw)

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls(**json.load(f))


class _ConvParams(nn.Conv2d):
    """Parameter holder; patchify filters are kept channels_last so their (Cout, k*k*Cin) GEMM view is free."""

    def __init__(self, cin, cout, k, stride=1, padding=0, groups=1):
        super().__init__(cin, cout, k, stride=stride, padding=padding, groups=groups)
        if groups == 1:
            self.weight.data = self.weight.data.contiguous(memory_format=CL)

    def forward(self, x):
        raise RuntimeError("executed by the parent module")


class _RowNorm(nn.LayerNorm):
    """LayerNorm over the channel of an NHWC activation (HF ConvNextLayerNorm in either data format,
    torchvision LayerNorm2d)."""

    def forward(self, x):
        return F.layer_norm(x, self.weight, self.bias, self.eps)


def _drop_path_scale(x, p, training):
    """per-sample keep/(1-p) of stochastic depth ("row" mode), or None"""
    if not training or p <= 0.0:
        return None
    keep = 1.0 - p
    return (torch.rand(x.shape[0], device=x.device) < keep).to(torch.float32) / keep


def _hooked(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None))


def _block(x, dw, ln, pw1, pw2, gamma, rowscale):
    y = X.dwconv(x, dw.weight, dw.bias)
    y = ln(y)
    if _hooked(pw1) or _hooked(pw2) or pw1.bias is None or pw2.bias is None:
        u = pw2(pw1(y, act="gelu"))                        # the two Linear modules stay the hookable path
    else:
        u = F.mlp_gelu(y, pw1.weight, pw1.bias, pw2.weight, pw2.bias)
    return X.layer_scale_residual(u, gamma, x, rowscale)


class ConvNextEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.patch_embeddings = _ConvParams(c.num_channels, c.hidden_sizes[0], c.patch_size, stride=c.patch_size)
        self.layernorm = _RowNorm(c.hidden_sizes[0], eps=1e-6)
        self.num_channels = c.num_channels
        self._k = c.patch_size

    def forward(self, x):
        return self.layernorm(X.patch_conv(x, self.patch_embeddings.weight, self.patch_embeddings.bias, self._k))


class ConvNextLayer(nn.Module):
    def __init__(self, c, dim, drop_path=0.0):
        super().__init__()
        self.dwconv = _ConvParams(dim, dim, 7, padding=3, groups=dim)
        self.layernorm = _RowNorm(dim, eps=1e-6)
        self.pwconv1 = Linear(dim, 4 * dim)
        self.pwconv2 = Linear(4 * dim, dim)
        if c.layer_scale_init_value > 0:
            self.layer_scale_parameter = nn.Parameter(c.layer_scale_init_value * torch.ones(dim))
        else:
            self.layer_scale_parameter = None
            self.register_buffer("_unit_scale", torch.ones(dim), persistent=False)
        self.drop_path_rate = float(drop_path)

    def forward(self, x):
        gamma = self.layer_scale_parameter if self.layer_scale_parameter is not None else self._unit_scale
        return _block(x, self.dwconv, self.layernorm, self.pwconv1, self.pwconv2, gamma,
                      _drop_path_scale(x, self.drop_path_rate, self.training))


class ConvNextStage(nn.Module):
    def __init__(self, c, in_channels, out_channels, kernel_size=2, stride=2, depth=2, drop_path_rates=None):
        super().__init__()
        if in_channels != out_channels or stride > 1:
            if kernel_size != stride:
                raise NotImplementedError("ConvNextStage: only stride == kernel downsampling (the ConvNeXt design)")
            self.downsampling_layer = nn.ModuleList([_RowNorm(in_channels, eps=1e-6),
                                                     _ConvParams(in_channels, out_channels, kernel_size, stride=stride)])
        else:
            self.downsampling_layer = nn.ModuleList()
        rates = drop_path_rates or [0.0] * depth
        self.layers = nn.ModuleList([ConvNextLayer(c, out_channels, rates[j]) for j in range(depth)])
        self._k = kernel_size

    def forward(self, x):
        if len(self.downsampling_layer):
            ln, conv = self.downsampling_layer
            x = X.patch_conv(ln(x), conv.weight, conv.bias, self._k)
        for layer in self.layers:
            x = layer(x)
        return x


class ConvNextEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        total = sum(c.depths)
        rates = [c.drop_path_rate * i / max(total - 1, 1) for i in range(total)]   # linspace(0, rate, total)
        self.stages = nn.ModuleList()
        prev, at = c.hidden_sizes[0], 0
        for i, (dim, depth) in enumerate(zip(c.hidden_sizes, c.depths)):
            self.stages.append(ConvNextStage(c, prev, dim, stride=2 if i > 0 else 1, depth=depth,
                                             drop_path_rates=rates[at:at + depth]))
            prev, at = dim, at + depth


class ConvNextModel(nn.Module):
    """forward(pixel_values (N,3,H,W)) -> .last_hidden_state (N, C_last, H/32, W/32), channels_last memory, in the
    compute dtype.  `.pooler_output` (LayerNorm of the spatial mean) is produced only with pool=True: the reference
    reads last_hidden_state alone (ourmodel.py:78), so by default `layernorm.*` gets no gradient, as there."""

    def __init__(self, config=None):
        super().__init__()
        self.config = c = config or ConvNextConfig()
        self.embeddings = ConvNextEmbeddings(c)
        self.encoder = ConvNextEncoder(c)
        self.layernorm = nn.LayerNorm(c.hidden_sizes[-1], eps=c.layer_norm_eps)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.normal_(m.weight, mean=0.0, std=c.initializer_range)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, pixel_values=None, pool=False, **unused):
        if pixel_values.shape[1] != self.config.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                             "configuration.")
        rt.need_gpu(pixel_values)
        x = rt.as_cl(pixel_values, hamspine.compute_dtype()).permute(0, 2, 3, 1)   # NHWC view of channels_last memory
        x = self.embeddings(x)
        for stage in self.encoder.stages:
            x = stage(x)
        pooled = None
        if pool:
            n, h, w, ch = x.shape
            pooled = F.layer_norm(F.mean_tokens(x.reshape(n, h * w, ch)), self.layernorm.weight, self.layernorm.bias,
                                  self.layernorm.eps)
        return SimpleNamespace(last_hidden_state=x.permute(0, 3, 1, 2), pooler_output=pooled)

    @classmethod
    def from_pretrained(cls, path, **kw):
        """Load a *local* HF directory (config.json + model.safetensors | pytorch_model.bin)."""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"ConvNextModel.from_pretrained({path!r}): not a local directory; hub downloads are "
                                    "unavailable")
        model = cls(ConvNextConfig.from_json_file(os.path.join(path, "config.json")))
        st_path = os.path.join(path, "model.safetensors")
        bin_path = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            sd = load_file(st_path)
        elif os.path.exists(bin_path):
            sd = torch.load(bin_path, map_location="cpu")
        elif os.environ.get("HAMSPINE_CONVNEXT_RANDOM_INIT") == "1":
            return model
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        clean = {k[len("convnext."):] if k.startswith("convnext.") else k: v for k, v in sd.items()
                 if not k.startswith("classifier.")}
        model.load_state_dict(clean, strict=True)
        return model


# =====================================================================================================================
# torchvision layout: convnext_*().features  (reference fallback branch, ourmodel.py:49-62)
# =====================================================================================================================
class _Slot(nn.Module):
    """parameter-free placeholder keeping torchvision's Sequential indices (Permute / GELU positions)"""

    def forward(self, x):
        raise RuntimeError("placeholder")


class CNBlock(nn.Module):
    def __init__(self, dim, layer_scale, sd_prob):
        super().__init__()
        self.block = nn.Sequential(_ConvParams(dim, dim, 7, padding=3, groups=dim), _Slot(), _RowNorm(dim, eps=1e-6),
                                   Linear(dim, 4 * dim), _Slot(), Linear(4 * dim, dim), _Slot())
        self.layer_scale = nn.Parameter(torch.ones(dim, 1, 1) * layer_scale)
        self.sd_prob = float(sd_prob)

    def forward(self, x):
        b = self.block
        return _block(x, b[0], b[2], b[3], b[5], self.layer_scale, _drop_path_scale(x, self.sd_prob, self.training))


class _TvStem(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(_ConvParams(cin, cout, 4, stride=4), _RowNorm(cout, eps=1e-6))

    def forward(self, x):
        return self[1](X.patch_conv(x, self[0].weight, self[0].bias, 4))


class _TvDown(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(_RowNorm(cin, eps=1e-6), _ConvParams(cin, cout, 2, stride=2))

    def forward(self, x):
        return X.patch_conv(self[0](x), self[1].weight, self[1].bias, 2)


class _TvStage(nn.Sequential):
    def forward(self, x):
        for blk in self:
            x = blk(x)
        return x


class ConvNextFeatures(nn.Sequential):
    """`torchvision.models.convnext_*().features`: (N,3,H,W) -> (N, C_last, H/32, W/32) channels_last."""

    def __init__(self, dims, depths, stochastic_depth_prob=0.0, layer_scale=1e-6):
        mods = [_TvStem(3, dims[0])]
        total, at = sum(depths), 0
        for i, (dim, depth) in enumerate(zip(dims, depths)):
            blocks = []
            for _ in range(depth):
                blocks.append(CNBlock(dim, layer_scale, stochastic_depth_prob * at / max(total - 1.0, 1.0)))
                at += 1
            mods.append(_TvStage(*blocks))
            if i + 1 < len(dims):
                mods.append(_TvDown(dim, dims[i + 1]))
        super().__init__(*mods)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        rt.need_gpu(x)
        x = rt.as_cl(x, hamspine.compute_dtype()).permute(0, 2, 3, 1)
        for m in self:
            x = m(x)
        return x.permute(0, 3, 1, 2)


def convnext_base_features(stochastic_depth_prob=0.5):
    """torchvision.models.convnext_base(weights=None).features (stochastic depth 0.5, layer scale 1e-6)."""
    return ConvNextFeatures([128, 256, 512, 1024], [3, 3, 27, 3], stochastic_depth_prob)
