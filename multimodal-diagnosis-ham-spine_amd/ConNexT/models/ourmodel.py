"""ConNeXT classifier (reference ConNexT/models/ourmodel.py:10-98): ConvNeXt-base image tower + BERT CLS text tower,
two 1x1-conv cross attentions (text->image and image->text), average pooling, linear head.  Same constructor
arguments, batch-dict forward contract and state-dict keys; the arithmetic runs on libhamspine_hip.so."""
import os

import torch
import torch.nn as nn

from hamspine import convnext_ops as X
from hamspine import functional as F
from hamspine import rt
from hamspine.nn import Linear
from hamspine.nn.convnext import ConvNextModel, convnext_base_features

from .BERT import BertEncoder


class _Conv1x1(nn.Conv2d):
    """nn.Conv2d(cin, cout, 1) parameters; applied to NHWC rows as a GEMM."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, kernel_size=1)

    def rows(self, x, out_dtype=None):
        return F.linear(x, self.weight.reshape(self.out_channels, self.in_channels), self.bias, out_dtype=out_dtype)

    def forward(self, x):
        """(B, C, H, W) -> (B, Cout, H, W), channels_last memory"""
        return self.rows(_nhwc(x)).permute(0, 3, 1, 2)


def _nhwc(x):
    """(B, C, H, W) of any layout -> contiguous (B, H, W, C) view / copy"""
    rt.need_gpu(x)
    return x.permute(0, 2, 3, 1).contiguous()


class CrossAttention(nn.Module):
    """softmax(Q(x) K(y)^T) V(y) over flattened positions, no scaling (ourmodel.py:10-31).  The projections write f32
    and the attention runs in f32: this sits on the f32 fusion/head side of the tower boundary."""

    def __init__(self, dim):
        super().__init__()
        self.query_conv = _Conv1x1(dim, dim)
        self.key_conv = _Conv1x1(dim, dim)
        self.value_conv = _Conv1x1(dim, dim)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, y):
        b, c, hx, wx = x.shape
        xr, yr = _nhwc(x), _nhwc(y)
        f32 = torch.float32
        q = self.query_conv.rows(xr, f32).reshape(b, hx * wx, c)
        k = self.key_conv.rows(yr, f32).reshape(b, -1, c)
        v = self.value_conv.rows(yr, f32).reshape(b, -1, c)
        out = X.attention_core(q, k, v, heads=1, scale=1.0)
        return out.reshape(b, hx, wx, c).permute(0, 3, 1, 2)


class _AvgPool(nn.AdaptiveAvgPool2d):
    def forward(self, x):
        b, c, h, w = x.shape
        return F.mean_tokens(_nhwc(x).reshape(b, h * w, c)).reshape(b, c, 1, 1)


class OurClassfierConvnextV2(nn.Module):
    def __init__(self, num_labels=2, pretrained=True, pretrained_path="/data/QLI/ConNexT/convnext-base-224",
                 bert_path="/data/QLI/BERT_pretain"):
        super().__init__()
        self.text_encoder = BertEncoder(bert_path)
        self._use_hf = False
        if pretrained and pretrained_path and os.path.isdir(pretrained_path):
            # the reference loads ConvNextForImageClassification and keeps its `.convnext` (ourmodel.py:41-45)
            self.image_encoder = ConvNextModel.from_pretrained(pretrained_path)
            self._use_hf = True
        else:
            # reference fallback: torchvision convnext_base().features; ImageNet weights cannot be fetched offline, so
            # this is the weights=None initialisation (ourmodel.py:49-62)
            self.image_encoder = convnext_base_features()
        self.conv = _Conv1x1(1024, 768)
        self.textbased_cross_attention = CrossAttention(dim=768)
        self.imagbased_cross_attention = CrossAttention(dim=768)
        self.avg_pool = _AvgPool((1, 1))
        self.fc = Linear(768, num_labels)

    def forward(self, batch_data):
        ids, mask, images = batch_data["input_ids"], batch_data["attention_mask"], batch_data["transformed_image"]

        def image_tower():
            return self.image_encoder(images).last_hidden_state if self._use_hf else self.image_encoder(images)
        if rt.towers_overlap_enabled() and ids.is_cuda:      # BERT on a side stream beside the ConvNeXt
            text, join = rt.run_on_tower_stream(lambda: self.text_encoder(ids, mask), ids, mask)   # (B, 768) f32
            image = image_tower()
            join(text)
        else:
            text = self.text_encoder(ids, mask)
            image = image_tower()
        image_reduced = self.conv(image)                                                        # (B, 768, h, w)
        text_expanded = text.unsqueeze(-1).unsqueeze(-1)
        b = images.shape[0]
        text_fused = self.textbased_cross_attention(image_reduced, text_expanded)
        pooled_1 = self.avg_pool(text_fused).reshape(b, 768)
        imag_fused = self.imagbased_cross_attention(text_expanded, image_reduced)
        pooled_2 = self.avg_pool(imag_fused).reshape(b, 768)
        return self.fc(F.axpby(pooled_1, pooled_2, 1.0, 1.0))
