"""Image / text towers with the reference's constructor signatures, attribute tree and state-dict
keys (reference encoder.py:13-134), running on the hamspine HIP kernels.

Differences, all additive: `backbone="resnet50"` is accepted (BASELINE config 2 asks for it; the
reference's ImageEncoder stops at resnet34) and nothing is downloaded -- weights come from a local
file or stay randomly initialised.
"""
import os

import torch
import torch.nn as nn

from hamspine.nn import BertModel, Linear, Stem, resnet18, resnet34, resnet50

_BACKBONES = {
    "resnet18": (resnet18, {"layer2": 128, "layer3": 256, "layer4": 512}, "resnet18-f37072fd.pth"),
    "resnet34": (resnet34, {"layer2": 128, "layer3": 256, "layer4": 512}, "resnet34-b627a593.pth"),
    "resnet50": (resnet50, {"layer2": 512, "layer3": 1024, "layer4": 2048}, "resnet50-0676ba61.pth"),
}


def _hub_cache_file(name):
    root = os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch"))
    return os.path.join(root, "hub", "checkpoints", name)


class ImageEncoder(nn.Module):
    """ResNet tower returning projected patch tokens: (B, N, feature_dim), or a dict of the layer2/3/4
    token sets when `multi_scale`."""

    def __init__(self, feature_dim=512, pretrained=True, weights_path=None, backbone="resnet18", multi_scale=False):
        super().__init__()
        self.multi_scale = multi_scale
        backbone = backbone.lower()
        if backbone not in _BACKBONES:
            raise ValueError(f"Unsupported backbone: {backbone}. Use resnet18, resnet34 or resnet50.")
        build, channels, hub_name = _BACKBONES[backbone]
        self.model = build(weights=None)
        if weights_path:
            if not os.path.exists(weights_path):
                raise FileNotFoundError(f"weights file not found: {weights_path}")
            self.model.load_state_dict(torch.load(weights_path, map_location="cpu"), strict=False)
        elif pretrained:
            cached = _hub_cache_file(hub_name)
            if not os.path.exists(cached):
                raise FileNotFoundError(
                    f"pretrained=True but no local ImageNet weights ({cached}); this build never downloads. "
                    "Pass image_weights_path or pretrained_image=False.")
            self.model.load_state_dict(torch.load(cached, map_location="cpu"), strict=False)
        self.model.fc = nn.Identity()
        # aliases of the backbone's submodules: hookable stage boundaries and the second key family of the
        # reference's state dict (image_encoder.stem.0.weight == image_encoder.model.conv1.weight, ...)
        self.stem = Stem(self.model.conv1, self.model.bn1, self.model.relu, self.model.maxpool)
        self.layer1 = self.model.layer1
        self.layer2 = self.model.layer2
        self.layer3 = self.model.layer3
        self.layer4 = self.model.layer4
        if multi_scale:
            self.proj2 = Linear(channels["layer2"], feature_dim)
            self.proj3 = Linear(channels["layer3"], feature_dim)
        self.proj4 = Linear(channels["layer4"], feature_dim)

    @staticmethod
    def _flatten_and_project(feat_map, proj):
        n, c, h, w = feat_map.shape
        tokens = feat_map.permute(0, 2, 3, 1).reshape(n, h * w, c)   # view: activations are NHWC in memory
        return proj(tokens)

    def forward(self, x):
        from hamspine import tower
        taps = None
        if not tower._has_hooks(self):      # hooks on stem / layerN[-1] (Grad-CAM) need the per-block modules to run
            taps = tower.resnet_taps(self.model, x, ("layer2", "layer3", "layer4") if self.multi_scale else ("layer4",))
        if taps is not None:
            f2, f3, f4 = taps if self.multi_scale else (None, None, taps[0])
        else:
            x = self.layer2(self.layer1(self.stem(x)))
            f2 = x
            f3 = self.layer3(f2)
            f4 = self.layer4(f3)
        t4 = self._flatten_and_project(f4, self.proj4)
        if not self.multi_scale:
            return t4
        return {"layer2": self._flatten_and_project(f2, self.proj2),
                "layer3": self._flatten_and_project(f3, self.proj3), "layer4": t4}


class TextEncoder(nn.Module):
    """BERT tower returning token features (B, L, hidden) (reference encoder.py:112-134)."""

    def __init__(self, model_path="bert-base-uncased", feature_dim=768):
        super().__init__()
        self.model = BertModel.from_pretrained(model_path)

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
