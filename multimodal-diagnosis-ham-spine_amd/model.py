"""MultimodalBaselineModel with the reference's 29-keyword constructor, methods and state-dict keys
(reference model.py:21-345), assembled from the hamspine towers / fusion operators / heads.
"""
import torch
import torch.nn as nn

import hamspine
from encoder import ImageEncoder, TextEncoder
from hamspine import functional as F
from hamspine import rt
from hamspine import small as S
from hamspine.nn import Linear, MLPHead
from modules.fusion_blocks import (
    BilinearFusionModule,
    ConcatFusionModule,
    FusionModule,
    HadamardFusionModule,
    MultiScaleFusionModule,
    SSMFusionModule,
    VMambaFusionModule,
    WeightedConcatFusionModule,
    pool_image_tokens,
)
from modules.gating import DualExpertGate
from modules.heads import AttentionPoolingClassifier, ResidualClassifier, build_kan_head
from modules.sequence_blocks import SequenceEncoder
from modules.tabular import TabularEncoder

_POOLED_FUSIONS = {
    "hadamard": HadamardFusionModule, "bilinear": BilinearFusionModule, "mamba": SSMFusionModule,
    "vmamba": VMambaFusionModule, "weighted_concat": WeightedConcatFusionModule, "concat": ConcatFusionModule,
}


class _TabularFusion(nn.Module):
    """nn.Sequential(Linear, ReLU, Dropout) of reference model.py:163-167 (keys "0.weight", "0.bias")."""

    def __init__(self, in_dim, out_dim, dropout):
        super().__init__()
        self.add_module("0", Linear(in_dim, out_dim))
        self.add_module("1", nn.ReLU())
        self.add_module("2", nn.Dropout(dropout))

    def forward(self, x):
        return getattr(self, "0")(x, act="relu", dropout_p=getattr(self, "2").p if self.training else 0.0)


class MultimodalBaselineModel(nn.Module):
    def __init__(
        self, num_classes, image_feature_dim=512, text_feature_dim=768, hidden_dim=256, dropout=0.2,
        pretrained_image=True, image_weights_path="/home/medteam/.cache/torch/hub/checkpoints/resnet18-f37072fd.pth",
        text_model_name="bert-base-uncased", num_heads=8, image_backbone="resnet18", classifier_type="mlp",
        fusion_type="basic", text_pool="cls", kan_num_groups=8, kan_act_mode="gelu",
        tabular_enabled=False, tabular_input_dim=0, tabular_hidden_dim=128, tabular_dropout=0.1,
        gate_enabled=False, gate_hidden_dim=128, gate_use_entropy=True, gate_local_mode="image_only",
        gate_context_mode="full",
        sequence_enabled=False, sequence_type="lstm", sequence_hidden_dim=256, sequence_num_layers=1,
        sequence_bidirectional=True, sequence_dropout=0.1, sequence_num_heads=4,
        global_local_enabled=False, global_local_crop_ratio=0.6, global_local_combine="avg",
    ):
        super().__init__()
        light = min(dropout, 0.1)   # fusion / head dropout is clamped (reference model.py:62-63)
        self.fusion_type = fusion_type
        self.tabular_enabled = tabular_enabled
        self.sequence_enabled = sequence_enabled
        self.global_local_enabled = global_local_enabled
        self.global_local_crop_ratio = global_local_crop_ratio
        self.global_local_combine = global_local_combine

        self.image_encoder = ImageEncoder(feature_dim=hidden_dim, pretrained=pretrained_image,
                                          weights_path=image_weights_path, backbone=image_backbone,
                                          multi_scale=(fusion_type == "multiscale"))
        if sequence_enabled:
            self.sequence_encoder = SequenceEncoder(
                input_dim=hidden_dim, hidden_dim=sequence_hidden_dim, encoder_type=sequence_type,
                num_layers=sequence_num_layers, bidirectional=sequence_bidirectional, dropout=sequence_dropout,
                num_heads=sequence_num_heads)
            self.sequence_proj = (Linear(sequence_hidden_dim, hidden_dim) if sequence_hidden_dim != hidden_dim
                                  else nn.Identity())
        self.global_local_proj = None
        if global_local_enabled and global_local_combine == "concat":
            self.global_local_proj = Linear(hidden_dim * 2, hidden_dim)
        self.text_encoder = TextEncoder(model_path=text_model_name, feature_dim=text_feature_dim)

        if fusion_type == "multiscale":
            self.fusion = MultiScaleFusionModule(text_dim=text_feature_dim, hidden_dim=hidden_dim,
                                                 num_heads=num_heads, dropout=light)
        elif fusion_type in _POOLED_FUSIONS:
            self.fusion = _POOLED_FUSIONS[fusion_type](text_dim=text_feature_dim, hidden_dim=hidden_dim,
                                                       text_pool=text_pool)
        else:
            self.fusion = FusionModule(text_dim=text_feature_dim, hidden_dim=hidden_dim, num_heads=num_heads,
                                       dropout=light)

        if tabular_enabled:
            if tabular_input_dim <= 0:
                raise ValueError("tabular_input_dim must be > 0 when tabular is enabled.")
            self.tabular_encoder = TabularEncoder(tabular_input_dim, hidden_dim=tabular_hidden_dim,
                                                  dropout=tabular_dropout)
            self.tabular_fusion = _TabularFusion(hidden_dim + tabular_hidden_dim, hidden_dim, light)

        self.gate_enabled = gate_enabled
        self.gate_local_mode = gate_local_mode
        self.gate_context_mode = gate_context_mode
        if gate_enabled:
            self.gate = DualExpertGate(lesion_dim=hidden_dim, context_dim=hidden_dim, hidden_dim=gate_hidden_dim,
                                       use_entropy=gate_use_entropy)

        self.classifier_type = classifier_type
        if classifier_type == "kan":
            self.classifier = build_kan_head(hidden_dim=hidden_dim, num_classes=num_classes, dropout=light,
                                             num_groups=kan_num_groups, act_mode=kan_act_mode)
        elif classifier_type == "residual":
            self.classifier = ResidualClassifier(hidden_dim, hidden_dim, num_classes, light)
        elif classifier_type == "attention_pooling":
            self.classifier = AttentionPoolingClassifier(hidden_dim, hidden_dim, num_classes, num_heads, light)
        else:
            self.classifier = MLPHead(hidden_dim, hidden_dim, num_classes, light)

    # ------------------------------------------------------------------------------------------
    def forward_features(self, image_input, text_input_ids, text_attention_mask, tabular_input=None,
                         ablation_mode=None):
        if ablation_mode == "image_only":
            return self._encode_image_tokens(image_input)[1]
        if rt.towers_overlap_enabled() and image_input.is_cuda:
            # the towers are independent until the fusion operator: BERT on a side stream beside the ResNet
            text_tokens, join = rt.run_on_tower_stream(
                lambda: self.text_encoder(text_input_ids, text_attention_mask), text_input_ids, text_attention_mask)
            image_tokens, pooled_image = self._encode_image_tokens(image_input)
            join(text_tokens)
        else:
            image_tokens, pooled_image = self._encode_image_tokens(image_input)
            text_tokens = self.text_encoder(text_input_ids, text_attention_mask)
        if ablation_mode == "text_off":
            text_tokens = F.axpby(text_tokens, None, 0.0, 0.0)   # zeros_like, keeps the graph shape
        if self.sequence_enabled and isinstance(self.fusion, MultiScaleFusionModule):
            image_tokens = {k: image_tokens for k in ("layer2", "layer3", "layer4")}
        fused = self.fusion(image_tokens, text_tokens, text_attention_mask)
        if self.tabular_enabled:
            if tabular_input is None:
                raise ValueError("tabular_input is required when tabular is enabled.")
            fused = self.tabular_fusion(S.concat2(fused, self.tabular_encoder(tabular_input)))
        return fused

    def forward(self, image_input, text_input_ids, text_attention_mask, tabular_input=None, ablation_mode=None):
        if ablation_mode is not None or not self.gate_enabled:
            return self.classifier(self.forward_features(image_input, text_input_ids, text_attention_mask,
                                                         tabular_input=tabular_input, ablation_mode=ablation_mode))
        # dual-expert gating: two feature passes, entropy of the local expert, convex mix of the logits
        context_mode = None if self.gate_context_mode == "full" else self.gate_context_mode
        context_feat = self.forward_features(image_input, text_input_ids, text_attention_mask,
                                             tabular_input=tabular_input, ablation_mode=context_mode)
        local_feat = self.forward_features(image_input, text_input_ids, text_attention_mask,
                                           tabular_input=tabular_input, ablation_mode=self.gate_local_mode)
        logits_context = self.classifier(context_feat)
        logits_local = self.classifier(local_feat)
        entropy = S.softmax_entropy(logits_local) if self.gate.use_entropy else None
        alpha = self.gate(local_feat, context_feat, entropy)
        return S.gate_mix(alpha, logits_local, logits_context)

    # ------------------------------------------------------------------------------------------
    def _pool_image_tokens(self, image_tokens):
        return pool_image_tokens(image_tokens)

    def _center_crop(self, x, ratio):
        return S.center_crop_resize(x, ratio)

    def _combine_tokens(self, global_tokens, local_tokens):
        g_dict, l_dict = isinstance(global_tokens, dict), isinstance(local_tokens, dict)
        if g_dict or l_dict:
            if not (g_dict and l_dict):
                raise ValueError("global/local token types must match.")
            return {k: F.axpby(global_tokens[k], local_tokens[k], 0.5, 0.5) for k in global_tokens}
        if self.global_local_combine == "concat":
            return self.global_local_proj(S.concat_tokens(global_tokens, local_tokens))
        return F.axpby(global_tokens, local_tokens, 0.5, 0.5)

    def _encode_image_tokens(self, image_input):
        if image_input.dim() == 5:
            if not self.sequence_enabled:
                raise ValueError("Sequence input provided but sequence encoder is disabled.")
            batch_size, seq_len = image_input.size(0), image_input.size(1)
            flat = image_input.reshape(batch_size * seq_len, *image_input.shape[2:])
            tokens = self.image_encoder(flat)
            if self.global_local_enabled:
                local_tokens = self.image_encoder(self._center_crop(flat, self.global_local_crop_ratio))
                tokens = self._combine_tokens(tokens, local_tokens)
            pooled = self._pool_image_tokens(tokens)                        # (B*T, H) f32
            seq_encoded = self.sequence_proj(self.sequence_encoder(pooled.reshape(batch_size, seq_len, -1)))
            seq_tokens = seq_encoded.unsqueeze(1)                           # one image token per study, (B, 1, H)
            if seq_tokens.dtype != hamspine.compute_dtype():                # token tensors live in the compute dtype
                seq_tokens = F.axpby(seq_tokens, None, 1.0, 0.0, hamspine.compute_dtype())
            return seq_tokens, seq_encoded
        tokens = self.image_encoder(image_input)
        if self.global_local_enabled:
            local_tokens = self.image_encoder(self._center_crop(image_input, self.global_local_crop_ratio))
            tokens = self._combine_tokens(tokens, local_tokens)
        return tokens, self._pool_image_tokens(tokens)

    def freeze_encoders(self):
        for p in self.image_encoder.parameters():
            p.requires_grad = False
        for p in self.text_encoder.parameters():
            p.requires_grad = False
