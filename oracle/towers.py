"""ORACLE -- test infrastructure only (never imported by the product package).

Plain-PyTorch CPU fp32 restatement of the two towers the reference takes from third-party packages
that are absent from /root/reference:

* torchvision.models.resnet18/34/50 (pinned torchvision 0.22.1; reference call sites encoder.py:35-42,
  mibf_net/model_resnet.py:15).  torchvision is not installed here -> PARITY UNPINNED for the tower as a
  whole; it follows the published ResNet v1.5 definition and is pinned by known-answer facts (parameter
  counts 11 689 512 / 21 797 672 / 25 557 032, torchvision state-dict key names) in tests/test_oracle_cpu.py.
* transformers.BertModel (pinned 4.57.1; call sites encoder.py:125-131, mibf_net/bert.py:9-12).
  Pinned against the installed transformers (5.15.0) BertModel on seeded weights in tests/test_oracle_cpu.py.
* transformers.ConvNextModel (call site ConNexT/models/ourmodel.py:43,78).  Pinned against the installed
  transformers ConvNextModel on seeded weights in tests/test_oracle_cpu.py and, through the reference's own
  OurClassfierConvnextV2 forward/backward, by tests/golden/e2e_connext.npz.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# ResNet (torchvision layout)
# ---------------------------------------------------------------------------------------------------
def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class OBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 3, stride, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, 1, 1), nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + idn)


class OBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride, 1), nn.BatchNorm2d(planes)   # v1.5: stride here
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + idn)


class OResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                               nn.BatchNorm2d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*mods)

    def features(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):
        x = self.features(x).mean(dim=(2, 3))
        return x if isinstance(self.fc, nn.Identity) else self.fc(x)


def oresnet(name, **kw):
    cfg = {"resnet18": (OBasicBlock, [2, 2, 2, 2]), "resnet34": (OBasicBlock, [3, 4, 6, 3]),
           "resnet50": (OBottleneck, [3, 4, 6, 3])}[name]
    return OResNet(*cfg, **kw)


# ---------------------------------------------------------------------------------------------------
# BERT (transformers.BertModel layout; eager attention, erf GELU, post-LN, eps 1e-12)
# ---------------------------------------------------------------------------------------------------
class _NS(nn.Module):
    """empty namespace module so parameter paths match the HF tree"""


class OBertLayer(nn.Module):
    def __init__(self, hidden, heads, inter, eps, p_hidden, p_attn):
        super().__init__()
        self.heads = heads
        att, so = _NS(), _NS()
        att.query, att.key, att.value = nn.Linear(hidden, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, hidden)
        att.dropout = nn.Dropout(p_attn)
        so.dense, so.LayerNorm, so.dropout = nn.Linear(hidden, hidden), nn.LayerNorm(hidden, eps=eps), nn.Dropout(p_hidden)
        self.attention = _NS()
        self.attention.self, self.attention.output = att, so
        self.intermediate = _NS()
        self.intermediate.dense = nn.Linear(hidden, inter)
        self.output = _NS()
        self.output.dense, self.output.LayerNorm = nn.Linear(inter, hidden), nn.LayerNorm(hidden, eps=eps)
        self.output.dropout = nn.Dropout(p_hidden)

    def forward(self, x, add_mask):
        B, L, H = x.shape
        a = self.attention.self
        hd = H // self.heads

        def split(t):
            return t.view(B, L, self.heads, hd).transpose(1, 2)
        q, k, v = split(a.query(x)), split(a.key(x)), split(a.value(x))
        s = q @ k.transpose(-1, -2) / math.sqrt(hd)
        if add_mask is not None:
            s = s + add_mask
        p = a.dropout(torch.softmax(s, dim=-1))
        ctx = (p @ v).transpose(1, 2).reshape(B, L, H)
        so = self.attention.output
        x1 = so.LayerNorm(so.dropout(so.dense(ctx)) + x)
        g = F.gelu(self.intermediate.dense(x1))
        return self.output.LayerNorm(self.output.dropout(self.output.dense(g)) + x1)


class OBertModel(nn.Module):
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **unused):
        super().__init__()
        e = _NS()
        e.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=0)   # HF: padding_idx=pad_token_id
        e.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        e.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size)
        e.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        e.dropout = nn.Dropout(hidden_dropout_prob)
        self.embeddings = e
        self.encoder = _NS()
        self.encoder.layer = nn.ModuleList([
            OBertLayer(hidden_size, num_attention_heads, intermediate_size, layer_norm_eps, hidden_dropout_prob,
                       attention_probs_dropout_prob) for _ in range(num_hidden_layers)])
        self.pooler = _NS()
        self.pooler.dense = nn.Linear(hidden_size, hidden_size)

    def forward(self, input_ids, attention_mask=None):
        B, L = input_ids.shape
        e = self.embeddings
        pos = torch.arange(L, device=input_ids.device)
        x = e.word_embeddings(input_ids) + e.position_embeddings(pos)[None] + e.token_type_embeddings.weight[0]
        x = e.dropout(e.LayerNorm(x))
        add = None
        if attention_mask is not None:
            add = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
        for layer in self.encoder.layer:
            x = layer(x, add)
        return x


# ---------------------------------------------------------------------------------------------------
# ConvNeXt (transformers layout: embeddings / encoder.stages[i].{downsampling_layer,layers} / layernorm)
# ---------------------------------------------------------------------------------------------------
class _ChannelNorm(nn.LayerNorm):
    """LayerNorm over C of an NCHW map"""

    def forward(self, x):
        return super().forward(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class OConvNextLayer(nn.Module):
    def __init__(self, dim, scale_init=1e-6):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.layernorm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.layer_scale_parameter = nn.Parameter(scale_init * torch.ones(dim))

    def forward(self, x):
        y = self.layernorm(self.dwconv(x).permute(0, 2, 3, 1))
        y = self.layer_scale_parameter * self.pwconv2(F.gelu(self.pwconv1(y)))
        return x + y.permute(0, 3, 1, 2)


class OConvNextStage(nn.Module):
    def __init__(self, cin, cout, depth, down):
        super().__init__()
        self.downsampling_layer = nn.ModuleList([_ChannelNorm(cin, eps=1e-6), nn.Conv2d(cin, cout, 2, stride=2)] if down else [])
        self.layers = nn.ModuleList([OConvNextLayer(cout) for _ in range(depth)])

    def forward(self, x):
        for m in self.downsampling_layer:
            x = m(x)
        for m in self.layers:
            x = m(x)
        return x


class OConvNextModel(nn.Module):
    def __init__(self, hidden_sizes=(128, 256, 512, 1024), depths=(3, 3, 27, 3), num_channels=3, patch_size=4,
                 layer_norm_eps=1e-12):
        super().__init__()
        self.embeddings = _NS()
        self.embeddings.patch_embeddings = nn.Conv2d(num_channels, hidden_sizes[0], patch_size, stride=patch_size)
        self.embeddings.layernorm = _ChannelNorm(hidden_sizes[0], eps=1e-6)
        self.encoder = _NS()
        self.encoder.stages = nn.ModuleList()
        prev = hidden_sizes[0]
        for i, (dim, depth) in enumerate(zip(hidden_sizes, depths)):
            self.encoder.stages.append(OConvNextStage(prev, dim, depth, down=i > 0))
            prev = dim
        self.layernorm = nn.LayerNorm(hidden_sizes[-1], eps=layer_norm_eps)   # pooled branch, unused by the reference

    def forward(self, x):
        """-> last_hidden_state (N, C_last, H/32, W/32)"""
        x = self.embeddings.layernorm(self.embeddings.patch_embeddings(x))
        for st in self.encoder.stages:
            x = st(x)
        return x
