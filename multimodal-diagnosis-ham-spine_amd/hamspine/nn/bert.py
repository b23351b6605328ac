"""BERT encoder with the module tree and state-dict keys of transformers.BertModel, executed by the
fused embedding / BertLayer nodes of hamspine.functional.

Follows transformers.models.bert.modeling_bert (BertEmbeddings, BertLayer: post-LN, erf-GELU,
LayerNorm eps 1e-12, additive key mask) as used by reference encoder.py:125-134, mibf_net/bert.py:9-13.
transformers is not a dependency of the product path: config.json and the weight file are read here.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

import hamspine

from .. import functional as F
from .layers import Linear, LayerNorm


class BertConfig:
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, pad_token_id=0, **unused):
        if hidden_act not in ("gelu",):
            raise ValueError(f"hidden_act={hidden_act!r}: only erf-GELU (BERT's default) is implemented")
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.layer_norm_eps = layer_norm_eps
        self.pad_token_id = pad_token_id

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls(**json.load(f))


def _init_linear(m, std=0.02):
    nn.init.normal_(m.weight, mean=0.0, std=std)
    if m.bias is not None:
        nn.init.zeros_(m.bias)


class BertEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=c.pad_token_id)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)
        for e in (self.word_embeddings, self.position_embeddings, self.token_type_embeddings):
            nn.init.normal_(e.weight, mean=0.0, std=0.02)
        with torch.no_grad():
            self.word_embeddings.weight[c.pad_token_id].zero_()

    def forward(self, input_ids):
        cfg = {"dtype": hamspine.compute_dtype(), "training": self.training, "dropout": float(self.dropout.p),
               "eps": float(self.LayerNorm.eps),
               "pad_id": -1 if self.word_embeddings.padding_idx is None else int(self.word_embeddings.padding_idx)}
        return F.BertEmbedFn.apply(input_ids, cfg, self.word_embeddings.weight, self.position_embeddings.weight,
                                   self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias)


class BertSelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query = Linear(c.hidden_size, c.hidden_size)
        self.key = Linear(c.hidden_size, c.hidden_size)
        self.value = Linear(c.hidden_size, c.hidden_size)
        self.dropout = nn.Dropout(c.attention_probs_dropout_prob)


class BertSelfOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = BertSelfAttention(c)
        self.output = BertSelfOutput(c)


class BertIntermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = Linear(c.hidden_size, c.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.dropout = nn.Dropout(c.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = BertAttention(c)
        self.intermediate = BertIntermediate(c)
        self.output = BertOutput(c)
        self._heads = c.num_attention_heads
        self._inter = c.intermediate_size

    def forward(self, hidden_states, attention_mask=None):
        a, so, o = self.attention.self, self.attention.output, self.output
        cfg = {"dtype": hamspine.compute_dtype(), "training": self.training, "heads": self._heads, "inter": self._inter,
               "eps": float(so.LayerNorm.eps), "hidden_dropout": float(so.dropout.p), "attn_dropout": float(a.dropout.p)}
        return F.BertLayerFn.apply(
            hidden_states, attention_mask, cfg,
            a.query.weight, a.query.bias, a.key.weight, a.key.bias, a.value.weight, a.value.bias,
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias,
            self.intermediate.dense.weight, self.intermediate.dense.bias,
            o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias)


class BertEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(c) for _ in range(c.num_hidden_layers)])


class BertPooler(nn.Module):
    """Present for state-dict fidelity (pooler.dense.*); the reference discards pooler_output
    (encoder.py:133, mibf_net/bert.py:13), so it is never evaluated and its parameters get no grad."""

    def __init__(self, c):
        super().__init__()
        self.dense = Linear(c.hidden_size, c.hidden_size)
        self.activation = nn.Tanh()


class BertModel(nn.Module):
    def __init__(self, config=None, add_pooling_layer=True):
        super().__init__()
        self.config = config or BertConfig()
        self.embeddings = BertEmbeddings(self.config)
        self.encoder = BertEncoder(self.config)
        self.pooler = BertPooler(self.config) if add_pooling_layer else None
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _init_linear(m)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, **unused):
        if token_type_ids is not None:
            raise NotImplementedError("token_type_ids: the reference never passes them (all zeros) -- not implemented")
        if attention_mask is not None:
            attention_mask = attention_mask.contiguous()
            if attention_mask.dtype != torch.int64:
                attention_mask = attention_mask.long()
        from .. import tower
        h = tower.bert_hidden(self, input_ids, attention_mask)      # whole tower in one node (no hooks registered)
        if h is None:
            h = self.embeddings(input_ids)
            for layer in self.encoder.layer:
                h = layer(h, attention_mask)
        return SimpleNamespace(last_hidden_state=h, pooler_output=None)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, **kw):
        """Load a *local* HF-format directory (config.json + model.safetensors | pytorch_model.bin).
        Hub names cannot be resolved (no network): raise instead of guessing."""
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"BertModel.from_pretrained({path!r}): not a local directory; hub downloads are unavailable")
        config = BertConfig.from_json_file(os.path.join(path, "config.json"))
        model = cls(config)
        st_path = os.path.join(path, "model.safetensors")
        bin_path = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            sd = load_file(st_path)
        elif os.path.exists(bin_path):
            sd = torch.load(bin_path, map_location="cpu")
        elif os.environ.get("HAMSPINE_BERT_RANDOM_INIT") == "1":
            return model     # synthetic benchmarks: architecture from config.json, random weights (stated in `data`)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        model.load_hf_state_dict(sd)
        return model

    def load_hf_state_dict(self, sd):
        clean = {}
        for k, v in sd.items():
            if k.startswith("bert."):
                k = k[5:]
            if k.startswith("cls.") or k.endswith("position_ids"):
                continue
            k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
            clean[k] = v
        missing, unexpected = self.load_state_dict(clean, strict=False)
        missing = [m for m in missing if not m.startswith("pooler.")]
        if missing or unexpected:
            raise RuntimeError(f"BERT checkpoint mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
