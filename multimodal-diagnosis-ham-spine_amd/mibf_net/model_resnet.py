"""MIBF-Net assembly (reference mibf_net/model_resnet.py:10-94): same constructor, forward contract
(dict batch in, dict of three logit sets out), cal_loss and state-dict keys."""
import torch
import torch.nn as nn

from hamspine import functional as F
from hamspine import rt
from hamspine import small as S
from hamspine.nn import Linear, resnet50

from .attention import MultiHeadCrossAttention_v2, SelfAttention
from .bert import BertEncoder


class _MLP(nn.Module):
    """nn.Sequential(Flatten, Linear, ReLU, Linear) -> keys "1.*" and "3.*" (model_resnet.py:28-34)."""

    def __init__(self, input_dim, num_labels):
        super().__init__()
        self.add_module("0", nn.Flatten(start_dim=1))
        self.add_module("1", Linear(input_dim, 512))
        self.add_module("2", nn.ReLU())
        self.add_module("3", Linear(512, num_labels))

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)
        return getattr(self, "3")(getattr(self, "1")(x, act="relu"))


class Resnet50WithOurs(nn.Module):
    def __init__(self, num_labels=6, loss_class="KL_loss", bert_path="/data/QLI/BERT_pretain", image_weights_path=None):
        super().__init__()
        self.text_encoder = BertEncoder(model_path=bert_path)
        # the reference fetches ImageNet weights here (models.resnet50(pretrained=True)); offline we take a local
        # state-dict file when given, else the torchvision initialisation
        backbone = resnet50()
        if image_weights_path:
            backbone.load_state_dict(torch.load(image_weights_path, map_location="cpu"), strict=False)
        backbone.fc = Linear(backbone.fc.in_features, 768)
        self.image_encoder = backbone
        self.textbased_cross_attention = MultiHeadCrossAttention_v2(dim=768, num_heads=1)
        self.imagbased_cross_attention = MultiHeadCrossAttention_v2(dim=768, num_heads=1)
        self.I2Iattention = SelfAttention(input_dim=768)
        self.fc = Linear(768 * 2, num_labels)
        self.fc_image = self._build_mlp(768, num_labels)
        self.fc_text = self._build_mlp(768, num_labels)
        self.loss_class = loss_class
        self.loss = nn.CrossEntropyLoss()   # kept for attribute parity; cal_loss uses the fused kernels

    def _build_mlp(self, input_dim, num_labels):
        return _MLP(input_dim, num_labels)

    def forward(self, batch_data):
        ids, mask = batch_data["input_ids"], batch_data["attention_mask"]
        if rt.towers_overlap_enabled() and ids.is_cuda:      # BERT on a side stream beside the ResNet
            text, join = rt.run_on_tower_stream(lambda: self.text_encoder(ids, mask), ids, mask)   # (B, 768) f32
            image = self.image_encoder(batch_data["transformed_image"])                      # (B, 768) f32
            join(text)
        else:
            text = self.text_encoder(ids, mask)
            image = self.image_encoder(batch_data["transformed_image"])
        text_tok, image_tok = text.unsqueeze(1), image.unsqueeze(1)
        text_fused = self.textbased_cross_attention(image_tok, text_tok)    # query = image (reference naming quirk)
        imag_fused = self.imagbased_cross_attention(text_tok, image_tok)
        b = image.shape[0]
        both = S.concat2(text_fused.reshape(b, 768), imag_fused.reshape(b, 768))
        return {"image_text": self.fc(both), "text": self.fc_text(text_fused), "image": self.fc_image(imag_fused)}

    def cal_loss(self, output, labels):
        if self.loss_class == "textimage_loss":
            return F.cross_entropy(output["image_text"], labels)
        if self.loss_class == "text_image_textimage_loss":
            a = F.cross_entropy(output["image"], labels)
            t = F.cross_entropy(output["text"], labels)
            f = F.cross_entropy(output["image_text"], labels)
            return F.axpby(F.axpby(a, t, 1.0, 1.0), f, 1.0, 1.0)
        return self.compute_kl_loss(output, labels)

    def compute_kl_loss(self, output, labels):
        return S.mp_loss(output["image"], output["text"], output["image_text"], labels)
