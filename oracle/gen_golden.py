"""ORACLE -- test infrastructure only.  Generates tests/golden/*.npz by RUNNING THE REFERENCE.

Run here (the reference tree is not shipped to the GPU box; only the .npz vectors travel):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

What is executed from /root/reference (read-only, imported, never copied):
  modules.fusion_blocks / modules.heads / modules.gating / modules.tabular, mibf_net.attention,
  mibf_net.bert (with a locally saved tiny transformers BertModel), and -- through a stand-in
  `torchvision.models` that maps resnet18/34/50 to oracle.towers.oresnet -- the reference's own
  model.py / encoder.py / mibf_net/model_resnet.py forward, cal_loss and autograd backward.
The stand-in only replaces the third-party tower the container lacks; every line of
glue that produces the vectors is the reference's.

Every vector is a function of (case name, seed): weights come from oracle.procedural, so a fixture holds
inputs, outputs and gradients only.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import towers  # noqa: E402
from oracle.procedural import load_procedural, synthetic_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 20260
TINY_BERT = dict(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0)
MIBF_BERT = dict(vocab_size=200, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=256,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0)


def install_torchvision_standin():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    def builder(name):
        def build(weights=None, pretrained=False, **kw):   # weights / pretrained accepted and ignored: no fetch
            return towers.oresnet(name)
        return build
    for n in ("resnet18", "resnet34", "resnet50"):
        setattr(models, n, builder(n))
    for n in ("ResNet18_Weights", "ResNet34_Weights", "ResNet50_Weights"):
        setattr(models, n, types.SimpleNamespace(IMAGENET1K_V1=None, DEFAULT=None))
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}::{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    print(f"wrote {name}.npz ({sum(a.size for a in flat.values())} values)")


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def run_case(name, module, inputs, seed, train=True, out_fn=None):
    """seeded weights -> forward -> sum(out * cotangent) -> backward; stores out, input grads, param grads"""
    load_procedural(module, seed)
    module.train(train)
    leaves = {k: v.clone().requires_grad_(True) if torch.is_tensor(v) and v.is_floating_point() else v
              for k, v in inputs.items()}
    out = out_fn(module, leaves) if out_fn else module(**leaves)
    cot = rnd(tuple(out.shape), seed + 1)
    (out * cot).sum().backward()
    gin = {k: v.grad for k, v in leaves.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    gw = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
    save(name, out=out, cot=cot, inp={k: v for k, v in inputs.items() if torch.is_tensor(v)}, gin=gin, gw=gw)


def gen_fusion():
    from modules import fusion_blocks as fb
    B, N, H, T, Lk, heads = 4, 10, 64, 96, 16, 4
    lens = torch.tensor([16, 9, 5, 2])
    mask = (torch.arange(Lk)[None] < lens[:, None]).long()
    img, txt = rnd((B, N, H), 11), rnd((B, Lk, T), 12)

    def call3(m, l):
        return m(l["img"], l["txt"], l["mask"])
    run_case("fusion_basic", fb.FusionModule(T, H, heads, 0.0), dict(img=img, txt=txt, mask=mask), SEED + 1, out_fn=call3)
    run_case("fusion_crossblock", fb.CrossAttentionBlock(T, H, heads, 0.0), dict(img=img, txt=txt, mask=mask), SEED + 2,
             out_fn=call3)
    ms = {"layer2": rnd((B, 36, H), 21), "layer3": rnd((B, 16, H), 22), "layer4": rnd((B, 4, H), 23)}

    def call_ms(m, l):
        return m({k: l[k] for k in ("layer2", "layer3", "layer4")}, l["txt"], l["mask"])
    run_case("fusion_multiscale", fb.MultiScaleFusionModule(T, H, heads, 0.0), dict(txt=txt, mask=mask, **ms), SEED + 3,
             out_fn=call_ms)
    kinds = {"concat": fb.ConcatFusionModule, "weighted_concat": fb.WeightedConcatFusionModule,
             "hadamard": fb.HadamardFusionModule, "bilinear": fb.BilinearFusionModule}
    i = 0
    for kn, cls in kinds.items():
        for pool in ("cls", "mean"):
            i += 1
            run_case(f"fusion_{kn}_{pool}_tensor", cls(T, H, text_pool=pool), dict(img=img, txt=txt, mask=mask),
                     SEED + 10 + i, out_fn=call3)
            run_case(f"fusion_{kn}_{pool}_dict", cls(T, H, text_pool=pool), dict(txt=txt, mask=mask, **ms),
                     SEED + 30 + i, out_fn=call_ms)


def gen_heads():
    from modules import gating, heads, tabular
    B, H, Cn = 6, 64, 7
    x = rnd((B, H), 31)
    run_case("head_residual", heads.ResidualClassifier(H, H, Cn, 0.0), dict(x=x), SEED + 50)
    run_case("head_attnpool", heads.AttentionPoolingClassifier(H, H, Cn, 4, 0.0), dict(x=x), SEED + 51)
    ent = rnd((B, 1), 33).abs()
    run_case("gate_entropy", gating.DualExpertGate(H, H, 32, True), dict(lesion_feat=x, context_feat=rnd((B, H), 32),
                                                                        entropy=ent), SEED + 52)
    run_case("gate_plain", gating.DualExpertGate(H, H, 32, False), dict(lesion_feat=x, context_feat=rnd((B, H), 32)),
             SEED + 53)
    run_case("tabular", tabular.TabularEncoder(11, 32, 0.0), dict(x=rnd((B, 11), 34)), SEED + 54)


def gen_ibfa():
    from mibf_net import attention as att
    B, D = 5, 64
    x, y = rnd((B, 1, D), 41), rnd((B, 1, D), 42)
    run_case("ibfa_h1", att.MultiHeadCrossAttention_v2(D, 1), dict(x=x, y=y), SEED + 60)
    run_case("ibfa_h4", att.MultiHeadCrossAttention_v2(D, 4), dict(x=x, y=y), SEED + 61)
    run_case("ibfa_selfattention", att.SelfAttention(D), dict(x=rnd((3, D, 5, 4), 47)), SEED + 63)
    run_case("ibfa_tokens", att.MultiHeadCrossAttention_v2(D, 4), dict(x=rnd((B, 3, D), 45), y=rnd((B, 6, D), 46)), SEED + 62)
    p = torch.softmax(rnd((B, 6), 43), -1)
    q = torch.softmax(rnd((B, 6), 44), -1)
    q[0, 0] = 0.0   # exercises the 1e-8 clamp
    save("kl_divergence", p=p, q=q, kl=att.compute_kl_divergence(p, q))


def gen_kan_moe():
    """reference ConNexT/models/block/{kan1,moe}.py run directly (namespace-package import)"""
    from ConNexT.models.block import kan1, moe
    x = rnd((6, 16), 51, 0.7)
    run_case("kan_linear", kan1.KANLinear(16, 12), dict(x=x), SEED + 80)
    run_case("kan1_stack", kan1.KAN1([16, 24, 8]), dict(x=x), SEED + 81)
    m = moe.MoE(16, 5, num_experts=4, hidden_size=8, k=2, layers_hidden=[16, 24, 5])
    load_procedural(m, SEED + 82)
    m.eval()                                   # noise off: the only deterministic gating mode
    xi = x.clone().requires_grad_(True)
    y, aux = m(xi)
    cot = rnd(tuple(y.shape), SEED + 83)
    ((y * cot).sum() + 3.0 * aux).backward()
    save("moe_eval", out=y, aux=aux, cot=cot, inp={"x": x}, gin={"x": xi.grad},
         gw={k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    # TRAIN mode (noisy top-k gating, moe.py:231-265, and the normal-CDF load estimate, :198-229): the only random draw of the
    # forward is torch.randn_like(clean_logits) (:247).  It is recorded by drawing it first under the same seed: randn of the
    # same shape / dtype from the same CPU generator state is the same tensor.
    xt = rnd((24, 16), 53, 0.7)
    m = moe.MoE(16, 5, num_experts=4, hidden_size=8, k=2, layers_hidden=[16, 24, 5])
    load_procedural(m, SEED + 84)
    with torch.no_grad():          # procedural w_noise is small: scale it up so the noise decides some of the top-k picks
        m.w_noise.mul_(3.0)
    m.train()
    torch.manual_seed(SEED + 85)
    noise = torch.randn(24, 4)
    torch.manual_seed(SEED + 85)
    xi = xt.clone().requires_grad_(True)
    gates, load = m.noisy_top_k_gating(xi, True)
    torch.manual_seed(SEED + 85)
    y, aux = m(xi)
    cot = rnd(tuple(y.shape), SEED + 86)
    ((y * cot).sum() + 3.0 * aux).backward()
    assert m.w_noise.grad is not None and m.w_noise.grad.abs().max() > 0
    clean_top = (xt @ m.w_gate).softmax(1).topk(2, dim=1).indices.sort(1).values
    noisy_top = (gates > 0).nonzero()[:, 1].view(24, 2)
    assert (clean_top != noisy_top).any(), "the noise should change at least one top-k decision"
    save("moe_train_noisy", out=y, aux=aux, cot=cot, noise=noise, gates=gates.detach(), load=load.detach(), inp={"x": xt},
         gin={"x": xi.grad}, gw={k: p.grad for k, p in m.named_parameters() if p.grad is not None})


def gen_kan_update_grid():
    """KANLinear.update_grid / KAN1.forward(update_grid=True) of the reference (kan1.py:167-212, 275-283) run directly:
    new knot grid, re-fitted spline coefficients and the layer's output after the update"""
    from ConNexT.models.block import kan1
    x = rnd((40, 16), 52, 0.9)
    lay = kan1.KANLinear(16, 12)
    load_procedural(lay, SEED + 87)
    before = lay(x)
    lay.update_grid(x)
    save("kan_update_grid", x=x, out_before=before, grid=lay.grid, spline_weight=lay.spline_weight, out_after=lay(x))
    net = kan1.KAN1([16, 24, 8])
    load_procedural(net, SEED + 88)
    y = net(x, update_grid=True)
    save("kan1_stack_update_grid", x=x, out=y, grid0=net.layers[0].grid, grid1=net.layers[1].grid,
         spline0=net.layers[0].spline_weight, spline1=net.layers[1].spline_weight)


def install_kan_head_standin():
    """`ikan.GroupKAN.GroupKANLinear` is an external package absent from the reference tree and from this container
    (modules/heads.py:7-25 resolves it to None and build_kan_head raises).  For the classifier_type="kan" vectors the
    reference's OWN in-tree KAN layer (ConNexT/models/block/kan1.py KANLinear) is put in its place with the constructor
    surface build_kan_head uses: act_mode -> base_activation, drop -> input dropout, num_groups validated only.
    Everything else that runs (build_kan_head's Sequential, model.py's wiring, KANLinear's arithmetic) is the reference's."""
    from ConNexT.models.block import kan1
    from modules import heads
    acts = {"gelu": torch.nn.GELU, "silu": torch.nn.SiLU, "swish": torch.nn.SiLU, "relu": torch.nn.ReLU,
            "identity": torch.nn.Identity}

    class GroupKANLinear(kan1.KANLinear):
        def __init__(self, in_features, out_features, act_mode="gelu", drop=0.0, num_groups=8):
            if in_features % num_groups != 0:
                raise ValueError("num_groups must divide in_features")
            super().__init__(in_features, out_features, base_activation=acts[act_mode])
            self.drop = torch.nn.Dropout(drop)

        def forward(self, x):
            return super().forward(self.drop(x))
    heads.GroupKANLinear = GroupKANLinear
    import model as ref_model
    assert ref_model.build_kan_head is heads.build_kan_head


def gen_kan_head(tmp):
    from ConNexT.models.block import kan1
    from modules import heads
    install_kan_head_standin()
    x = rnd((6, 64), 55, 0.7)
    run_case("kan_linear_gelu", kan1.KANLinear(16, 12, base_activation=torch.nn.GELU), dict(x=rnd((6, 16), 51, 0.7)), SEED + 84)
    run_case("head_kan", heads.build_kan_head(64, 7, dropout=0.0, num_groups=8, act_mode="gelu"), dict(input=x), SEED + 85)
    run_case("head_kan_silu", heads.build_kan_head(64, 7, dropout=0.0, num_groups=4, act_mode="silu"), dict(input=x), SEED + 86)


def install_train_script_standins():
    """scripts/train.py imports torch.utils.tensorboard (absent) and data_loader (needs PIL/cv2/torchvision.transforms)
    at module level (train.py:12,21); neither is touched by SupConLoss / FocalLoss (train.py:23-61).  Empty stand-ins
    make the module importable so that the two loss classes that run are the reference's own."""
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {})
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    dl = types.ModuleType("data_loader")
    dl.create_data_loader = None
    sys.modules.setdefault("data_loader", dl)


def gen_losses(tmp):
    """G9: FocalLoss / SupConLoss of scripts/train.py:23-61 and CE(label_smoothing=0.02, class weights) of train.py:240-254"""
    install_train_script_standins()
    from scripts import train as ref_train
    B, Cn, D = 12, 7, 32
    labels = torch.tensor([0, 3, 3, 1, 6, 0, 2, 3, 5, 1, 1, 4])
    weight = 0.5 + torch.rand(Cn, generator=torch.Generator().manual_seed(91))
    for name, gamma, w in (("focal_g2", 2.0, None), ("focal_g1p5_weighted", 1.5, weight)):
        logits = rnd((B, Cn), 92, 2.0).requires_grad_(True)
        loss = ref_train.FocalLoss(gamma=gamma, weight=w)(logits, labels)
        loss.backward()
        save(name, logits=logits.detach(), labels=labels, weight=w if w is not None else torch.zeros(0), gamma=gamma,
             loss=loss.detach(), dlogits=logits.grad)
    for name, temp, lab in (("supcon_t007", 0.07, labels), ("supcon_t05_singletons", 0.5, torch.tensor([0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 3]))):
        feat = rnd((B, D), 93).requires_grad_(True)
        loss = ref_train.SupConLoss(temperature=temp)(feat, lab)
        loss.backward()
        save(name, features=feat.detach(), labels=lab, temperature=temp, loss=loss.detach(), dfeatures=feat.grad)
    for name, sm, w in (("ce_smooth002", 0.02, None), ("ce_smooth002_weighted", 0.02, weight)):
        logits = rnd((B, Cn), 94, 2.0).requires_grad_(True)
        loss = torch.nn.CrossEntropyLoss(weight=w, label_smoothing=sm)(logits, labels)
        loss.backward()
        save(name, logits=logits.detach(), labels=labels, weight=w if w is not None else torch.zeros(0), smoothing=sm,
             loss=loss.detach(), dlogits=logits.grad)


def save_tiny_bert(cfg, tmp):
    from transformers import BertConfig, BertModel
    d = os.path.join(tmp, f"bert_{cfg['hidden_size']}_{cfg['num_hidden_layers']}")
    BertModel(BertConfig(**cfg)).save_pretrained(d)
    return d


def gen_bert(tmp):
    from mibf_net.bert import BertEncoder
    d = save_tiny_bert(TINY_BERT, tmp)
    enc = BertEncoder(model_path=d)
    load_procedural(enc, SEED + 70)
    enc.train()
    _, ids, mask, _ = synthetic_batch(4, 8, 24, TINY_BERT["vocab_size"], 7, seed=71, min_len=3)
    cls = enc(ids, mask)
    hidden = enc.bert(input_ids=ids, attention_mask=mask).last_hidden_state
    cot = rnd(tuple(hidden.shape), 72)
    (hidden * cot).sum().backward()
    gw = {k: p.grad for k, p in enc.named_parameters() if p.grad is not None}
    save("bert_tiny", ids=ids, mask=mask, cls=cls, hidden=hidden, cot=cot, gw=gw)


def e2e_store(name, model, logits, loss, extra=None):
    loss.backward()
    norms = {k: p.grad.norm() for k, p in model.named_parameters() if p.grad is not None}
    small = {k: p.grad for k, p in model.named_parameters() if p.grad is not None and p.numel() <= 20000}
    none_grad = sorted(k for k, p in model.named_parameters() if p.grad is None)
    save(name, logits=logits, loss=loss, gnorm=norms, gw=small, nograd=np.array(none_grad), **(extra or {}))


def gen_e2e_baseline(tmp):
    import model as ref_model
    d = save_tiny_bert(TINY_BERT, tmp)
    images, ids, mask, labels = synthetic_batch(4, 64, 24, TINY_BERT["vocab_size"], 7, seed=81, min_len=3)
    tab = rnd((4, 9), 82)
    common = dict(num_classes=7, text_feature_dim=64, hidden_dim=64, dropout=0.0, pretrained_image=False,
                  image_weights_path=None, text_model_name=d, num_heads=4, image_backbone="resnet18")
    cases = {
        "e2e_basic_mlp": dict(fusion_type="basic", classifier_type="mlp"),
        "e2e_multiscale_residual": dict(fusion_type="multiscale", classifier_type="residual"),
        "e2e_concat_tabular": dict(fusion_type="concat", classifier_type="mlp", tabular_enabled=True, tabular_input_dim=9,
                                   tabular_hidden_dim=32, tabular_dropout=0.0),
        "e2e_bilinear_attnpool": dict(fusion_type="bilinear", classifier_type="attention_pooling", text_pool="mean"),
        "e2e_gate_globallocal": dict(fusion_type="basic", classifier_type="mlp", gate_enabled=True, gate_hidden_dim=32,
                                     global_local_enabled=True, global_local_crop_ratio=0.6),
        "e2e_hadamard_imageonly": dict(fusion_type="hadamard", classifier_type="mlp"),
        "e2e_globallocal_concat": dict(fusion_type="basic", classifier_type="residual", global_local_enabled=True,
                                       global_local_crop_ratio=0.5, global_local_combine="concat"),
        # 5-D slice sequences (B, T, 3, H, W): per-slice tower -> SequenceEncoder -> one image token
        "e2e_sequence_lstm": dict(fusion_type="basic", classifier_type="mlp", sequence_enabled=True, sequence_type="lstm",
                                  sequence_hidden_dim=32, sequence_bidirectional=True, sequence_dropout=0.0),
        "e2e_sequence_gru2": dict(fusion_type="concat", classifier_type="mlp", sequence_enabled=True, sequence_type="gru",
                                  sequence_hidden_dim=64, sequence_num_layers=2, sequence_bidirectional=False,
                                  sequence_dropout=0.0),
        "e2e_sequence_transformer": dict(fusion_type="multiscale", classifier_type="mlp", sequence_enabled=True,
                                         sequence_type="transformer", sequence_hidden_dim=32, sequence_num_layers=2,
                                         sequence_dropout=0.0, sequence_num_heads=4),
    }
    # BASELINE config 5: hierarchical multiscale fusion + global-local dual stream + dual-expert gate + KAN head
    # (reference configs: fusion_type multiscale, global_local.enabled, gate.enabled, classifier_type kan); the KAN
    # layer is the stand-in of install_kan_head_standin (external ikan package absent)
    install_kan_head_standin()
    cases["e2e_c5_multiscale_gl_gate_kan"] = dict(fusion_type="multiscale", classifier_type="kan", kan_num_groups=8,
                                                  kan_act_mode="gelu", gate_enabled=True, gate_hidden_dim=32,
                                                  global_local_enabled=True, global_local_crop_ratio=0.6)
    only = os.environ.get("GEN_E2E_ONLY")            # regenerate a single case without touching the others
    seq_images = rnd((2, 3, 3, 64, 64), 83)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.02)
    for i, (name, kw) in enumerate(cases.items()):
        if only and name != only:
            continue
        m = ref_model.MultimodalBaselineModel(**common, **kw)
        load_procedural(m, SEED + 100 + i)
        m.train()
        t = tab if kw.get("tabular_enabled") else None
        if kw.get("sequence_enabled"):
            logits = m.classifier(m.forward_features(seq_images, ids[:2], mask[:2]))
            e2e_store(name, m, logits, crit(logits, labels[:2]))
            m.eval()
            with torch.no_grad():
                ev = m(seq_images, ids[:2], mask[:2])
            np.savez_compressed(os.path.join(OUT, name + "_eval.npz"), logits=ev.numpy())
            continue
        if name == "e2e_hadamard_imageonly":
            logits = m(images, ids, mask, tabular_input=t, ablation_mode="image_only")
        elif kw.get("gate_enabled"):
            logits = m(images, ids, mask, tabular_input=t)
        else:   # the training script calls forward_features + classifier separately (scripts/train.py:373-380)
            logits = m.classifier(m.forward_features(images, ids, mask, tabular_input=t))
        e2e_store(name, m, logits, crit(logits, labels))
        # eval-mode logits (running statistics) for the inference path
        m.eval()
        with torch.no_grad():
            ev = m(images, ids, mask, tabular_input=t)
        np.savez_compressed(os.path.join(OUT, name + "_eval.npz"), logits=ev.numpy())


def gen_e2e_mibf(tmp):
    from mibf_net import model_resnet as mr
    d = save_tiny_bert(MIBF_BERT, tmp)
    images, ids, mask, labels = synthetic_batch(4, 64, 16, MIBF_BERT["vocab_size"], 6, seed=91, min_len=3)
    for j, lc in enumerate(("KL_loss", "text_image_textimage_loss")):
        m = mr.Resnet50WithOurs(num_labels=6, loss_class=lc, bert_path=d)
        load_procedural(m, SEED + 200)
        m.train()
        out = m({"input_ids": ids, "attention_mask": mask, "transformed_image": images})
        loss = m.cal_loss(out, labels)
        e2e_store(f"e2e_mibf_{lc}", m, out["image_text"], loss, extra=dict(text=out["text"], image=out["image"]))


CONNEXT_CFG = dict(hidden_sizes=[16, 32, 64, 1024], depths=[1, 1, 2, 1])


def gen_convnext(tmp):
    """HF ConvNextModel tower alone (forward + backward), then the reference's OurClassfierConvnextV2 end to end.
    The reference hard-codes its BERT directory (ConNexT/models/BERT.py:11); the redirect below only maps that
    literal path to the tiny local model, every other line that runs is the reference's."""
    from transformers import BertModel, ConvNextConfig, ConvNextForImageClassification, ConvNextModel
    cfg = ConvNextConfig(**CONNEXT_CFG)
    m = ConvNextModel(cfg)
    load_procedural(m, SEED + 300)
    m.train()
    x = rnd((3, 3, 64, 64), 301)
    xi = x.clone().requires_grad_(True)
    out = m(xi).last_hidden_state
    cot = rnd(tuple(out.shape), 302)
    (out * cot).sum().backward()
    gw = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    save("convnext_tiny", x=x, out=out, cot=cot, dx=xi.grad, gnorm={k: g.norm() for k, g in gw.items()},
         gw={k: g for k, g in gw.items() if g.numel() <= 20000})

    cdir = os.path.join(tmp, "convnext")
    ConvNextForImageClassification(cfg).save_pretrained(cdir)
    bdir = save_tiny_bert(MIBF_BERT, tmp)
    orig = BertModel.from_pretrained.__func__

    def redirected(cls, path, *a, **k):
        return orig(cls, bdir if path == "/data/QLI/BERT_pretain" else path, *a, **k)
    BertModel.from_pretrained = classmethod(redirected)
    try:
        from ConNexT.models.ourmodel import OurClassfierConvnextV2
        net = OurClassfierConvnextV2(num_labels=5, pretrained=True, pretrained_path=cdir)
    finally:
        BertModel.from_pretrained = classmethod(orig)
    assert net._use_hf, "reference fell back to torchvision: the HF directory did not load"
    load_procedural(net, SEED + 310)
    net.train()
    images, ids, mask, labels = synthetic_batch(3, 64, 16, MIBF_BERT["vocab_size"], 5, seed=311, min_len=3)
    logits = net({"input_ids": ids, "attention_mask": mask, "transformed_image": images})
    e2e_store("e2e_connext", net, logits, torch.nn.CrossEntropyLoss()(logits, labels))


def main():
    torch.set_num_threads(8)
    from transformers import BertConfig, BertModel  # noqa: F401  (import before the stand-in exists: transformers probes torchvision)
    install_torchvision_standin()
    sys.path.insert(0, REF)
    groups = {"fusion": lambda tmp: gen_fusion(), "heads": lambda tmp: gen_heads(), "ibfa": lambda tmp: gen_ibfa(),
              "kan_moe": lambda tmp: gen_kan_moe(), "kan_update_grid": lambda tmp: gen_kan_update_grid(), "kan_head": gen_kan_head, "losses": gen_losses, "bert": gen_bert,
              "e2e_baseline": gen_e2e_baseline,
              "e2e_mibf": gen_e2e_mibf, "convnext": gen_convnext}
    want = sys.argv[1:] or list(groups)          # optional: regenerate only the named groups
    with tempfile.TemporaryDirectory() as tmp:
        for g in want:
            groups[g](tmp)


if __name__ == "__main__":
    main()
