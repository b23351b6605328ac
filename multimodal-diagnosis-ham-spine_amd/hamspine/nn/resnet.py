"""ResNet-18/34/50 (v1.5: stride on the 3x3 of a Bottleneck) with torchvision's module tree and
state-dict keys, executed by the fused stem / residual-block nodes of hamspine.functional.

Follows the public torchvision.models.resnet definition used by reference encoder.py:35-42 and
mibf_net/model_resnet.py:15 (torchvision itself is not a dependency).
"""
import torch
import torch.nn as nn

import hamspine

from .. import functional as F
from .layers import Linear

CL = torch.channels_last


class ConvParams(nn.Conv2d):
    """Parameter holder for a bias-free conv; the arithmetic is fused into the parent block."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=False)
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        # KRSC in memory (NHWC filters): what the implicit-GEMM kernels read; survives .to(device)
        # and load_state_dict (copy_ keeps the destination strides)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)

    def geo(self):
        return (self.in_channels, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0])

    def forward(self, x):
        raise RuntimeError("ConvParams is executed by its parent block (fused conv+BN node)")


class BatchNormParams(nn.BatchNorm2d):
    """Parameter/buffer holder of a BatchNorm2d.  num_batches_tracked is kept as a host counter and
    flushed into the buffer when the state dict is taken (saves one tiny launch per BN per step)."""

    def __init__(self, c):
        super().__init__(c)
        self._pending = 0

    def bump(self):
        self._pending += 1

    def _flush(self):
        if self._pending and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending
        self._pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *a, **k):
        self._pending = 0
        super()._load_from_state_dict(*a, **k)

    def forward(self, x):
        raise RuntimeError("BatchNormParams is executed by its parent block (fused conv+BN node)")


def _stage(conv, bn, training):
    use_batch = training or not bn.track_running_stats
    return {"geo": conv.geo(), "rm": bn.running_mean, "rv": bn.running_var, "bn": bn, "batch": use_batch}


class _Block(nn.Module):
    def _pairs(self):
        raise NotImplementedError

    def forward(self, x):
        pairs = self._pairs()
        has_ds = self.downsample is not None
        if has_ds:
            pairs = pairs + [(self.downsample[0], self.downsample[1])]
        bn0 = pairs[0][1]
        training = self.training
        cfg = {
            "dtype": hamspine.compute_dtype(), "training": training, "eps": bn0.eps,
            "momentum": bn0.momentum if bn0.momentum is not None else 0.1,
            "has_ds": has_ds, "stages": [_stage(c, b, training) for c, b in pairs],
        }
        params = []
        for c, b in pairs:
            params += [c.weight, b.weight, b.bias]
            if training:
                b.bump()
        return F.ResBlockFn.apply(x, cfg, *params)


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, 3, stride, 1)
        self.bn1 = BatchNormParams(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = ConvParams(planes, planes, 3, 1, 1)
        self.bn2 = BatchNormParams(planes)
        self.downsample = downsample
        self.stride = stride

    def _pairs(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, 1)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, stride, 1)
        self.bn2 = BatchNormParams(planes)
        self.conv3 = ConvParams(planes, planes * 4, 1)
        self.bn3 = BatchNormParams(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def _pairs(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]


def stem_forward(conv1, bn1, x, training):
    cfg = {
        "dtype": hamspine.compute_dtype(), "training": training, "eps": bn1.eps,
        "momentum": bn1.momentum if bn1.momentum is not None else 0.1,
        "running_mean": bn1.running_mean, "running_var": bn1.running_var,
    }
    if training:
        bn1.bump()
    return F.StemFn.apply(x, conv1.weight, bn1.weight, bn1.bias, cfg)


class Stem(nn.Sequential):
    """nn.Sequential(conv1, bn1, relu, maxpool) of reference encoder.py:63-68 (keys "0.weight", "1.*"),
    run as one fused node.  Output: (N, 64, H/4, W/4), NCHW-shaped with NHWC memory."""

    def forward(self, x):
        return stem_forward(self[0], self[1], x, self.training)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = ConvParams(3, 64, 7, 2, 3)
        self.bn1 = BatchNormParams(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(ConvParams(self.inplanes, planes * block.expansion, 1, stride),
                                       BatchNormParams(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def features(self, x):
        from .. import tower
        taps = tower.resnet_taps(self, x, ("layer4",))      # whole tower in one node (no hooks registered)
        if taps is not None:
            return taps[0]
        x = stem_forward(self.conv1, self.bn1, x, self.training)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        return self.layer4(x)

    def forward(self, x):
        x = self.features(x)
        N, Cc, H, W = x.shape
        tokens = x.permute(0, 2, 3, 1).reshape(N, H * W, Cc)   # a view: the memory already is NHWC
        if isinstance(self.fc, nn.Identity):
            return F.mean_tokens(tokens, out_f32=False)        # global average pool
        # a class count that is not a multiple of 8 cannot be a bf16 leading dimension: such an fc runs in f32
        pooled = F.mean_tokens(tokens, out_f32=self.fc.out_features % 8 != 0)
        return self.fc(pooled, out_dtype=torch.float32)   # logits / embeddings leave the tower as f32


def _build(block, layers, weights=None, **kw):
    if weights is not None:
        raise RuntimeError("pretrained torchvision weights cannot be fetched here; load a local state_dict instead")
    return ResNet(block, layers, **kw)


def resnet18(weights=None, **kw):
    return _build(BasicBlock, [2, 2, 2, 2], weights, **kw)


def resnet34(weights=None, **kw):
    return _build(BasicBlock, [3, 4, 6, 3], weights, **kw)


def resnet50(weights=None, **kw):
    return _build(Bottleneck, [3, 4, 6, 3], weights, **kw)
