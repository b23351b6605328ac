"""Case table shared by the CPU (oracle vs golden) and GPU (product vs golden / oracle) tests.

Each entry mirrors one case of oracle/gen_golden.py: same seed, same constructor arguments.  `oracle` builds
the CPU restatement, `product` the HIP-backed module (import deferred so the CPU suite never needs a GPU).
"""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 20260
T, H, HEADS = 96, 64, 4
TINY_BERT = dict(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0)
MIBF_BERT = dict(vocab_size=200, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=256,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0)
CONNEXT_CFG = dict(hidden_sizes=[16, 32, 64, 1024], depths=[1, 1, 2, 1])
LEVELS = ("layer2", "layer3", "layer4")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        if "::" in k:
            a, b = k.split("::", 1)
            out.setdefault(a, {})[b] = torch.from_numpy(z[k])
        else:
            out[k] = torch.from_numpy(z[k]) if z[k].dtype.kind != "U" else z[k]
    return out


def _o():
    from oracle import models as om
    return om


def _p():
    import modules.fusion_blocks as fb
    import modules.gating as gt
    import modules.heads as hd
    import modules.tabular as tb
    from mibf_net import attention as att
    return fb, gt, hd, tb, att


def _kan():
    from ConNexT.models.block import kan1
    return kan1


def call_fusion(m, inp):
    if "img" in inp:
        return m(inp["img"], inp["txt"], inp["mask"])
    return m({k: inp[k] for k in LEVELS}, inp["txt"], inp["mask"])


def call_kwargs(m, inp):
    return m(**inp)


# name -> (seed, oracle factory, product factory, caller)
MODULE_CASES = {
    "fusion_basic": (SEED + 1, lambda: _o().OFusionModule(T, H, HEADS, 0.0), lambda: _p()[0].FusionModule(T, H, HEADS, 0.0),
                     call_fusion),
    "fusion_crossblock": (SEED + 2, lambda: _o().OCrossAttentionBlock(T, H, HEADS, 0.0),
                          lambda: _p()[0].CrossAttentionBlock(T, H, HEADS, 0.0), call_fusion),
    "fusion_multiscale": (SEED + 3, lambda: _o().OMultiScaleFusionModule(T, H, HEADS, 0.0),
                          lambda: _p()[0].MultiScaleFusionModule(T, H, HEADS, 0.0), call_fusion),
    "head_residual": (SEED + 50, lambda: _o().OResidualClassifier(H, H, 7, 0.0),
                      lambda: _p()[2].ResidualClassifier(H, H, 7, 0.0), call_kwargs),
    "head_attnpool": (SEED + 51, lambda: _o().OAttentionPoolingClassifier(H, H, 7, 4, 0.0),
                      lambda: _p()[2].AttentionPoolingClassifier(H, H, 7, 4, 0.0), call_kwargs),
    "gate_entropy": (SEED + 52, lambda: _o().ODualExpertGate(H, H, 32, True), lambda: _p()[1].DualExpertGate(H, H, 32, True),
                     call_kwargs),
    "gate_plain": (SEED + 53, lambda: _o().ODualExpertGate(H, H, 32, False), lambda: _p()[1].DualExpertGate(H, H, 32, False),
                   call_kwargs),
    "tabular": (SEED + 54, lambda: _o().OTabularEncoder(11, 32, 0.0), lambda: _p()[3].TabularEncoder(11, 32, 0.0),
                call_kwargs),
    "ibfa_h1": (SEED + 60, lambda: _o().OCrossAttnV2(64, 1), lambda: _p()[4].MultiHeadCrossAttention_v2(64, 1), call_kwargs),
    "ibfa_h4": (SEED + 61, lambda: _o().OCrossAttnV2(64, 4), lambda: _p()[4].MultiHeadCrossAttention_v2(64, 4), call_kwargs),
    "ibfa_selfattention": (SEED + 63, lambda: _o().OSelfAttention(64), lambda: _p()[4].SelfAttention(64), call_kwargs),
    "ibfa_tokens": (SEED + 62, lambda: _o().OCrossAttnV2(64, 4), lambda: _p()[4].MultiHeadCrossAttention_v2(64, 4), call_kwargs),
}

MODULE_CASES["kan_linear"] = (SEED + 80, lambda: _o().OKANLinear(16, 12), lambda: _kan().KANLinear(16, 12), call_kwargs)
MODULE_CASES["kan_linear_gelu"] = (SEED + 84, lambda: _o().OKANLinear(16, 12, base_activation=torch.nn.GELU),
                                   lambda: _kan().KANLinear(16, 12, base_activation=torch.nn.GELU), call_kwargs)
# classifier_type="kan" head: reference modules/heads.py:108-140 run with its in-tree KANLinear in place of the
# absent external ikan layer (oracle/gen_golden.py: install_kan_head_standin)
MODULE_CASES["head_kan"] = (SEED + 85, lambda: _o().okan_head(H, 7, 0.0, 8, "gelu"),
                            lambda: _p()[2].build_kan_head(H, 7, dropout=0.0, num_groups=8, act_mode="gelu"),
                            lambda m, inp: m(inp["input"]))
MODULE_CASES["head_kan_silu"] = (SEED + 86, lambda: _o().okan_head(H, 7, 0.0, 4, "silu"),
                                 lambda: _p()[2].build_kan_head(H, 7, dropout=0.0, num_groups=4, act_mode="silu"),
                                 lambda m, inp: m(inp["input"]))
MODULE_CASES["kan1_stack"] = (SEED + 81, lambda: _o().OKAN1([16, 24, 8]), lambda: _kan().KAN1([16, 24, 8]), call_kwargs)

_KINDS = {"concat": ("OConcatFusion", "ConcatFusionModule"), "weighted_concat": ("OWeightedConcatFusion", "WeightedConcatFusionModule"),
          "hadamard": ("OHadamardFusion", "HadamardFusionModule"), "bilinear": ("OBilinearFusion", "BilinearFusionModule")}
_i = 0
for _kn, (_on, _pn) in _KINDS.items():
    for _pool in ("cls", "mean"):
        _i += 1
        for _var, _off in (("tensor", 10), ("dict", 30)):
            MODULE_CASES[f"fusion_{_kn}_{_pool}_{_var}"] = (
                SEED + _off + _i,
                (lambda on=_on, pool=_pool: getattr(_o(), on)(T, H, pool)),
                (lambda pn=_pn, pool=_pool: getattr(_p()[0], pn)(T, H, text_pool=pool)),
                call_fusion)

E2E_COMMON = dict(num_classes=7, text_feature_dim=64, hidden_dim=64, dropout=0.0, num_heads=4, image_backbone="resnet18")
E2E_CASES = {
    "e2e_basic_mlp": (SEED + 100, dict(fusion_type="basic", classifier_type="mlp")),
    "e2e_multiscale_residual": (SEED + 101, dict(fusion_type="multiscale", classifier_type="residual")),
    "e2e_concat_tabular": (SEED + 102, dict(fusion_type="concat", classifier_type="mlp", tabular_enabled=True,
                                            tabular_input_dim=9, tabular_hidden_dim=32, tabular_dropout=0.0)),
    "e2e_bilinear_attnpool": (SEED + 103, dict(fusion_type="bilinear", classifier_type="attention_pooling", text_pool="mean")),
    "e2e_gate_globallocal": (SEED + 104, dict(fusion_type="basic", classifier_type="mlp", gate_enabled=True,
                                              gate_hidden_dim=32, global_local_enabled=True, global_local_crop_ratio=0.6)),
    "e2e_hadamard_imageonly": (SEED + 105, dict(fusion_type="hadamard", classifier_type="mlp")),
    "e2e_globallocal_concat": (SEED + 106, dict(fusion_type="basic", classifier_type="residual", global_local_enabled=True,
                                                global_local_crop_ratio=0.5, global_local_combine="concat")),
    "e2e_sequence_lstm": (SEED + 107, dict(fusion_type="basic", classifier_type="mlp", sequence_enabled=True,
                                           sequence_type="lstm", sequence_hidden_dim=32, sequence_bidirectional=True,
                                           sequence_dropout=0.0)),
    "e2e_sequence_gru2": (SEED + 108, dict(fusion_type="concat", classifier_type="mlp", sequence_enabled=True,
                                           sequence_type="gru", sequence_hidden_dim=64, sequence_num_layers=2,
                                           sequence_bidirectional=False, sequence_dropout=0.0)),
    "e2e_sequence_transformer": (SEED + 109, dict(fusion_type="multiscale", classifier_type="mlp", sequence_enabled=True,
                                                  sequence_type="transformer", sequence_hidden_dim=32,
                                                  sequence_num_layers=2, sequence_dropout=0.0, sequence_num_heads=4)),
    # BASELINE config 5: hierarchical multiscale fusion + global-local dual stream + dual-expert gate + KAN head
    "e2e_c5_multiscale_gl_gate_kan": (SEED + 110, dict(fusion_type="multiscale", classifier_type="kan", kan_num_groups=8,
                                                       kan_act_mode="gelu", gate_enabled=True, gate_hidden_dim=32,
                                                       global_local_enabled=True, global_local_crop_ratio=0.6)),
}
LOSS_CASES = ("focal_g2", "focal_g1p5_weighted", "supcon_t007", "supcon_t05_singletons", "ce_smooth002",
              "ce_smooth002_weighted")


def e2e_inputs(kw=None):
    """(images, ids, mask, labels, tabular) of an end-to-end case; sequence cases take (2, T=3, 3, 64, 64) slices and the
    first two rows of ids / mask / labels (oracle/gen_golden.py: seq_images)"""
    from oracle.procedural import synthetic_batch
    images, ids, mask, labels = synthetic_batch(4, 64, 24, TINY_BERT["vocab_size"], 7, seed=81, min_len=3)
    tab = torch.randn((4, 9), generator=torch.Generator().manual_seed(82))
    if kw and kw.get("sequence_enabled"):
        images = torch.randn((2, 3, 3, 64, 64), generator=torch.Generator().manual_seed(83))
        ids, mask, labels, tab = ids[:2], mask[:2], labels[:2], tab[:2]
    return images, ids, mask, labels, tab


def mibf_inputs():
    from oracle.procedural import synthetic_batch
    return synthetic_batch(4, 64, 16, MIBF_BERT["vocab_size"], 6, seed=91, min_len=3)


def connext_inputs():
    from oracle.procedural import synthetic_batch
    return synthetic_batch(3, 64, 16, MIBF_BERT["vocab_size"], 5, seed=311, min_len=3)


def save_convnext_dir(cfg, path):
    """config.json + seeded-shape weights in HF layout for ConvNextModel.from_pretrained (product side)."""
    import json
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(dict(cfg, model_type="convnext", hidden_act="gelu"), f)
    from oracle.towers import OConvNextModel
    sd = {"convnext." + k: v for k, v in OConvNextModel(**cfg).state_dict().items()}
    sd["classifier.weight"] = torch.zeros(1000, cfg["hidden_sizes"][-1])     # ConvNextForImageClassification head: dropped
    sd["classifier.bias"] = torch.zeros(1000)
    torch.save(sd, os.path.join(path, "pytorch_model.bin"))
    return path


def e2e_forward(model, name, kw, images, ids, mask, tab):
    t = tab if kw.get("tabular_enabled") else None
    if name == "e2e_hadamard_imageonly":
        return model(images, ids, mask, tabular_input=t, ablation_mode="image_only")
    if kw.get("gate_enabled"):
        return model(images, ids, mask, tabular_input=t)
    if kw.get("sequence_enabled"):
        return model.classifier(model.forward_features(images, ids, mask))
    return model.classifier(model.forward_features(images, ids, mask, tabular_input=t))


def save_bert_dir(cfg, path):
    """config.json + seeded weights in HF layout for BertModel.from_pretrained (product side, no transformers)."""
    import json
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(dict(cfg, hidden_act="gelu", layer_norm_eps=1e-12, model_type="bert"), f)
    from oracle.towers import OBertModel
    torch.save(OBertModel(**cfg).state_dict(), os.path.join(path, "pytorch_model.bin"))
    return path
