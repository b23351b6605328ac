"""CPU suite (no GPU): pins the oracle against the golden vectors that were produced by running the
reference (oracle/gen_golden.py), against the installed transformers BertModel, and against the
known-answer facts of SURVEY.md Appendix B; checks the C ABI surface and the product's module trees."""
import os

import pytest
import torch

import golden_cases as gc
from oracle import models as om
from oracle import towers
from oracle.procedural import load_procedural

TOL = dict(rtol=2e-5, atol=2e-6)


def _close(a, b, what, rtol=2e-5, atol=2e-6):
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + atol, f"{what}: max err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("name", sorted(gc.MODULE_CASES))
def test_oracle_module_matches_reference_vectors(name):
    seed, ofac, _, caller = gc.MODULE_CASES[name]
    fx = gc.load(name)
    m = load_procedural(ofac(), seed).train()
    inp = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in fx["inp"].items()}
    out = caller(m, inp)
    _close(out, fx["out"], f"{name}: out")
    (out * fx["cot"]).sum().backward()
    for k, g in fx.get("gin", {}).items():
        _close(inp[k].grad, g, f"{name}: d/d{k}", rtol=1e-4)
    params = dict(m.named_parameters())
    assert set(fx["gw"]) <= set(params), f"{name}: key mismatch {set(fx['gw']) - set(params)}"
    for k, g in fx["gw"].items():
        _close(params[k].grad, g, f"{name}: grad {k}", rtol=1e-4)


def test_oracle_moe_matches_reference_vectors():
    fx = gc.load("moe_eval")
    m = load_procedural(om.OMoE(16, 5, 4, 2, [16, 24, 5]), gc.SEED + 82).eval()
    x = fx["inp"]["x"].clone().requires_grad_(True)
    y, aux = m(x)
    _close(y, fx["out"], "moe out")
    _close(aux, fx["aux"], "moe aux loss")
    ((y * fx["cot"]).sum() + 3.0 * aux).backward()
    _close(x.grad, fx["gin"]["x"], "moe dx", rtol=1e-4)
    params = dict(m.named_parameters())
    for k, g in fx["gw"].items():
        _close(params[k].grad, g, f"moe grad {k}", rtol=1e-4)


def test_oracle_moe_train_mode_matches_reference_vectors():
    """noisy top-k gating + normal-CDF load estimate (reference moe.py:198-265) with the recorded torch.randn_like draw:
    gates, load, aux loss, outputs and every gradient incl. w_noise"""
    fx = gc.load("moe_train_noisy")
    m = load_procedural(om.OMoE(16, 5, 4, 2, [16, 24, 5]), gc.SEED + 84)
    with torch.no_grad():
        m.w_noise.mul_(3.0)
    m.train()
    x = fx["inp"]["x"].clone().requires_grad_(True)
    gates, load = m.gating(x, fx["noise"])
    _close(gates, fx["gates"], "moe train gates")
    _close(load, fx["load"], "moe train load")
    assert torch.equal(gates > 0, fx["gates"] > 0), "top-k decisions"
    y, aux = m(x, noise=fx["noise"])
    _close(y, fx["out"], "moe train out")
    _close(aux, fx["aux"], "moe train aux loss")
    ((y * fx["cot"]).sum() + 3.0 * aux).backward()
    _close(x.grad, fx["gin"]["x"], "moe train dx", rtol=1e-4)
    params = dict(m.named_parameters())
    assert "w_noise" in fx["gw"]
    for k, g in fx["gw"].items():
        _close(params[k].grad, g, f"moe train grad {k}", rtol=1e-4)


def test_kl_divergence_vector():
    fx = gc.load("kl_divergence")
    _close(om.okl(fx["p"], fx["q"]), fx["kl"], "kl")


def test_oracle_bert_matches_reference_vectors():
    fx = gc.load("bert_tiny")
    m = towers.OBertModel(**gc.TINY_BERT)
    # the reference wraps the HF model as `.bert`; procedural names carry that prefix
    holder = torch.nn.Module()
    holder.bert = m
    load_procedural(holder, gc.SEED + 70).train()
    hidden = m(fx["ids"], fx["mask"])
    _close(hidden, fx["hidden"], "bert hidden")
    _close(hidden[:, 0], fx["cls"], "bert cls")
    (hidden * fx["cot"]).sum().backward()
    grads = {"bert." + k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    for k, g in fx["gw"].items():
        # key.bias gradients are analytically zero (softmax is shift invariant): both sides hold f32 noise
        _close(grads[k], g, f"bert grad {k}", rtol=2e-4, atol=2e-5)
    assert all(k.startswith("bert.pooler") for k in set(grads) ^ set(fx["gw"]))


def test_oracle_bert_matches_installed_transformers():
    tr = pytest.importorskip("transformers")
    cfg = dict(gc.TINY_BERT)
    hf = tr.BertModel(tr.BertConfig(**cfg)).eval()
    m = towers.OBertModel(**cfg).eval()
    load_procedural(m, 7)
    missing, unexpected = hf.load_state_dict(m.state_dict(), strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing)
    ids = torch.randint(0, cfg["vocab_size"], (3, 20), generator=torch.Generator().manual_seed(1))
    mask = torch.ones(3, 20, dtype=torch.long)
    mask[1, 11:] = 0
    mask[2, 3:] = 0
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        got = m(ids, mask)
    _close(got, ref, "oracle vs transformers")


@pytest.mark.parametrize("name", sorted(gc.E2E_CASES))
def test_oracle_e2e_matches_reference_vectors(name):
    seed, kw = gc.E2E_CASES[name]
    fx = gc.load(name)
    images, ids, mask, labels, tab = gc.e2e_inputs(kw)
    m = om.OMultimodalBaselineModel(bert_cfg=gc.TINY_BERT, **gc.E2E_COMMON, **kw)
    load_procedural(m, seed).train()
    logits = gc.e2e_forward(m, name, kw, images, ids, mask, tab)
    _close(logits, fx["logits"], f"{name}: logits", rtol=1e-4)
    assert torch.equal(logits.argmax(1), fx["logits"].argmax(1))
    loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=0.02)
    _close(loss, fx["loss"], f"{name}: loss", rtol=1e-4)
    loss.backward()
    params = dict(m.named_parameters())
    for k, n in fx["gnorm"].items():
        _close(params[k].grad.norm(), n, f"{name}: |grad {k}|", rtol=5e-3, atol=1e-7)
    nograd = sorted(k for k, p in params.items() if p.grad is None)
    assert nograd == sorted(str(s) for s in fx["nograd"])
    m.eval()
    with torch.no_grad():
        ev = m(images, ids, mask, tabular_input=tab if kw.get("tabular_enabled") else None)
    _close(ev, gc.load(name + "_eval")["logits"], f"{name}: eval logits", rtol=1e-4)


@pytest.mark.parametrize("loss_class", ["KL_loss", "text_image_textimage_loss"])
def test_oracle_mibf_matches_reference_vectors(loss_class):
    fx = gc.load(f"e2e_mibf_{loss_class}")
    images, ids, mask, labels = gc.mibf_inputs()
    m = om.OResnet50WithOurs(6, gc.MIBF_BERT, loss_class)
    load_procedural(m, gc.SEED + 200).train()
    out = m({"input_ids": ids, "attention_mask": mask, "transformed_image": images})
    for k, fk in (("image_text", "logits"), ("text", "text"), ("image", "image")):
        _close(out[k], fx[fk], f"mibf {k}", rtol=1e-4)
    loss = m.cal_loss(out, labels)
    _close(loss, fx["loss"], "mibf loss", rtol=1e-4)
    loss.backward()
    params = dict(m.named_parameters())
    for k, n in fx["gnorm"].items():
        _close(params[k].grad.norm(), n, f"mibf |grad {k}|", rtol=5e-3, atol=1e-7)
    # unused parameters keep grad None (DDP find_unused_parameters semantics, train_resnet.py:134)
    nograd = sorted(k for k, p in params.items() if p.grad is None)
    assert nograd == sorted(str(s) for s in fx["nograd"])
    assert any(k.startswith("I2Iattention") for k in nograd) and any("pooler" in k for k in nograd)


def test_oracle_convnext_matches_reference_vectors():
    fx = gc.load("convnext_tiny")
    m = load_procedural(towers.OConvNextModel(**gc.CONNEXT_CFG), gc.SEED + 300).train()
    x = fx["x"].clone().requires_grad_(True)
    out = m(x)
    _close(out, fx["out"], "convnext last_hidden_state", rtol=1e-4)
    (out * fx["cot"]).sum().backward()
    _close(x.grad, fx["dx"], "convnext dx", rtol=1e-3)
    params = dict(m.named_parameters())
    for k, n in fx["gnorm"].items():
        _close(params[k].grad.norm(), n, f"convnext |grad {k}|", rtol=1e-3, atol=1e-7)
    for k, g in fx["gw"].items():
        _close(params[k].grad, g, f"convnext grad {k}", rtol=1e-3)
    assert sorted(k for k, p in params.items() if p.grad is None) == ["layernorm.bias", "layernorm.weight"]


def test_oracle_convnext_matches_installed_transformers():
    tr = pytest.importorskip("transformers")
    hf = tr.ConvNextModel(tr.ConvNextConfig(**gc.CONNEXT_CFG)).eval()
    m = load_procedural(towers.OConvNextModel(**gc.CONNEXT_CFG), 11).eval()
    hf.load_state_dict(m.state_dict(), strict=True)
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        _close(m(x), hf(x).last_hidden_state, "oracle convnext vs transformers", rtol=1e-4)
    # convnext-base-224 geometry: parameter count of the published model (87 566 464 without the classifier head)
    base = towers.OConvNextModel()
    assert sum(p.numel() for p in base.parameters()) == 87566464


def test_oracle_connext_matches_reference_vectors():
    fx = gc.load("e2e_connext")
    images, ids, mask, labels = gc.connext_inputs()
    m = load_procedural(om.OConNeXT(5, gc.MIBF_BERT, gc.CONNEXT_CFG), gc.SEED + 310).train()
    logits = m({"input_ids": ids, "attention_mask": mask, "transformed_image": images})
    _close(logits, fx["logits"], "connext logits", rtol=1e-4)
    assert torch.equal(logits.argmax(1), fx["logits"].argmax(1))
    loss = torch.nn.functional.cross_entropy(logits, labels)
    _close(loss, fx["loss"], "connext loss", rtol=1e-4)
    loss.backward()
    params = dict(m.named_parameters())
    for k, n in fx["gnorm"].items():
        _close(params[k].grad.norm(), n, f"connext |grad {k}|", rtol=5e-3, atol=1e-7)
    nograd = sorted(k for k, p in params.items() if p.grad is None)
    assert nograd == sorted(str(s) for s in fx["nograd"])
    # softmax over the single text key is identically 1: the text-side query/key convolutions get exactly zero gradient
    for k in ("textbased_cross_attention.query_conv.weight", "textbased_cross_attention.key_conv.bias"):
        assert float(fx["gnorm"][k]) == 0.0 and float(params[k].grad.abs().max()) == 0.0


def test_resnet_known_answers():
    counts = {"resnet18": 11689512, "resnet34": 21797672, "resnet50": 25557032}
    for name, n in counts.items():
        m = towers.oresnet(name)
        assert sum(p.numel() for p in m.parameters()) == n
    keys = set(towers.oresnet("resnet50").state_dict())
    for k in ("conv1.weight", "bn1.running_var", "layer1.0.downsample.0.weight", "layer1.0.downsample.1.num_batches_tracked",
              "layer4.2.conv3.weight", "layer3.5.bn2.bias", "fc.weight"):
        assert k in keys
    r50 = towers.oresnet("resnet50")
    r50.fc = torch.nn.Linear(2048, 768)
    assert sum(p.numel() for p in r50.parameters()) == 25081664


@pytest.mark.parametrize("name", gc.LOSS_CASES)
def test_oracle_losses_match_reference_vectors(name):
    """G9: vectors made by the reference's own FocalLoss / SupConLoss (scripts/train.py:23-61) and CE"""
    fx = gc.load(name)
    if name.startswith("supcon"):
        f = fx["features"].clone().requires_grad_(True)
        loss = om.osupcon(f, fx["labels"], float(fx["temperature"]))
        loss.backward()
        _close(loss, fx["loss"], f"{name}: loss")
        _close(f.grad, fx["dfeatures"], f"{name}: dfeatures", rtol=1e-4)
        return
    z = fx["logits"].clone().requires_grad_(True)
    w = fx["weight"] if fx["weight"].numel() else None
    if name.startswith("focal"):
        loss = om.ofocal(z, fx["labels"], gamma=float(fx["gamma"]), weight=w)
    else:
        loss = torch.nn.functional.cross_entropy(z, fx["labels"], weight=w, label_smoothing=float(fx["smoothing"]))
    loss.backward()
    _close(loss, fx["loss"], f"{name}: loss")
    _close(z.grad, fx["dlogits"], f"{name}: dlogits", rtol=1e-4)


def test_focal_and_supcon_basics():
    z = torch.randn(8, 7, generator=torch.Generator().manual_seed(3))
    y = torch.randint(0, 7, (8,), generator=torch.Generator().manual_seed(4))
    assert torch.allclose(om.ofocal(z, y, gamma=0.0), torch.nn.functional.cross_entropy(z, y))
    f = torch.randn(8, 16, generator=torch.Generator().manual_seed(5))
    assert om.osupcon(f, y).isfinite()


# ------------------------------------------------------------------------------ product, host side only
def test_c_abi_exports_every_declared_symbol():
    import hamspine._lib as L
    lib = L.lib()
    syms = L.exported_symbols()
    assert len(syms) >= 60
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.hs_version() >= 100


def test_ctypes_mirrors_match_the_compiled_abi_structs():
    import ctypes
    import hamspine._lib as L
    lib = L.lib()
    for which, mirror in L.abi_structs():
        assert lib.hs_abi_sizeof(which) == ctypes.sizeof(mirror), (which, mirror.__name__, lib.hs_abi_sizeof(which))
    assert lib.hs_abi_sizeof(99) == -1


def test_batchnorm_workspace_holds_the_in_launch_statistics_rows():
    """Host logic only (no launch): for every 1x1-convolution shape of ResNet50 at batch 32 -- and at batch 1024 -- the
    colstats buffer a BatchNorm-finishing GEMM needs (hs_gemm_bn_finish_rows: tile rows + one merge row per 32) fits the
    workspace hs_batchnorm_ws_bytes sizes, and it is the tile-row count plus ceil(rows / 32) (0 merge rows up to 32 tiles)."""
    import ctypes as C
    import hamspine._lib as L
    lib = L.lib()
    for batch in (32, 1024):
        for hw, cout, cin in ((56 * 56, 64, 64), (56 * 56, 256, 64), (28 * 28, 512, 128), (14 * 14, 1024, 256), (7 * 7, 2048, 512),
                              (7 * 7, 512, 2048)):
            M = batch * hw
            p = L.GemmParams()
            p.dtype = p.out_dtype = L.HS_BF16
            p.a_kind, p.b_kind = L.A_KC, L.B_KC
            p.M, p.N, p.K = M, cout, cin
            p.A = p.B = p.D = p.colstats = 4096               # only the configuration is looked at
            p.a_elems, p.b_elems = M * cin, cout * cin
            p.lda = p.ldb = cin
            p.ldd = cout
            p.alpha = 1.0
            rows, frows = lib.hs_gemm_stat_rows(C.byref(p)), lib.hs_gemm_bn_finish_rows(C.byref(p))
            want = rows + (0 if rows <= 32 else -(-rows // 32))
            # 0 = this launch's tile has no finishing variant (the 128x128 tiles of wide layers at large batch): BatchNorm
            # then finishes its statistics in its own launch, as before
            assert rows > 0 and frows in ((want,) if batch == 32 else (want, 0)), (M, cout, cin, rows, frows)
            assert lib.hs_batchnorm_ws_bytes(M, cout, L.HS_BF16) >= max(frows, rows) * cout * 12, (M, cout, cin)
    p.colstats = None
    assert lib.hs_gemm_bn_finish_rows(C.byref(p)) == 0        # no statistics requested: nothing to finish


def test_product_refuses_cpu_tensors():
    import hamspine
    from hamspine import functional as F
    with pytest.raises(hamspine.HamspineError):
        F.linear(torch.zeros(2, 8), torch.zeros(4, 8), None)


def test_product_state_dict_keys_match_oracle(tmp_path):
    import model as product_model
    from mibf_net.model_resnet import Resnet50WithOurs
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / "bert"))
    for name, (seed, kw) in gc.E2E_CASES.items():
        pm = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                                   **gc.E2E_COMMON, **kw)
        o = om.OMultimodalBaselineModel(bert_cfg=gc.TINY_BERT, **gc.E2E_COMMON, **kw)
        pk = {k: tuple(v.shape) for k, v in pm.state_dict().items()}
        ok = {k: tuple(v.shape) for k, v in o.state_dict().items()}
        assert pk == ok, f"{name}: {set(pk) ^ set(ok)}"
        assert "image_encoder.stem.0.weight" in pk and "image_encoder.model.conv1.weight" in pk
    d2 = gc.save_bert_dir(gc.MIBF_BERT, str(tmp_path / "bert768"))
    pm = Resnet50WithOurs(num_labels=6, bert_path=d2)
    o = om.OResnet50WithOurs(6, gc.MIBF_BERT)
    assert {k: tuple(v.shape) for k, v in pm.state_dict().items()} == {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert sum(p.numel() for p in pm.image_encoder.parameters()) == 25081664


def test_product_convnext_keys_match_oracle(tmp_path):
    from hamspine.nn.convnext import ConvNextConfig, ConvNextModel, convnext_base_features
    pm = ConvNextModel(ConvNextConfig(**gc.CONNEXT_CFG))
    o = towers.OConvNextModel(**gc.CONNEXT_CFG)
    assert {k: tuple(v.shape) for k, v in pm.state_dict().items()} == {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert sum(p.numel() for p in ConvNextModel(ConvNextConfig.base()).parameters()) == 87566464
    # torchvision convnext_base().features: same parameters minus the pooled-branch LayerNorm (2 x 1024)
    tv = convnext_base_features()
    assert sum(p.numel() for p in tv.parameters()) == 87566464 - 2048
    assert abs(tv[7][2].sd_prob - 0.5) < 1e-12 and tv[1][0].sd_prob == 0.0
    from ConNexT.models.ourmodel import OurClassfierConvnextV2
    bdir = gc.save_bert_dir(gc.MIBF_BERT, str(tmp_path / "b"))
    cdir = gc.save_convnext_dir(gc.CONNEXT_CFG, str(tmp_path / "c"))
    m = OurClassfierConvnextV2(num_labels=5, pretrained_path=cdir, bert_path=bdir)
    oo = om.OConNeXT(5, gc.MIBF_BERT, gc.CONNEXT_CFG)
    assert m._use_hf
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in oo.state_dict().items()}


def test_product_param_counts_headline():
    from hamspine.nn import BertConfig, BertModel, resnet18, resnet34, resnet50
    assert sum(p.numel() for p in resnet18().parameters()) == 11689512
    assert sum(p.numel() for p in resnet34().parameters()) == 21797672
    assert sum(p.numel() for p in resnet50().parameters()) == 25557032
    assert sum(p.numel() for p in BertModel(BertConfig()).parameters()) == 109482240
    import modules.fusion_blocks as fb
    assert sum(p.numel() for p in fb.FusionModule(768, 256, 8).parameters()) == 1315584
    assert sum(p.numel() for p in fb.CrossAttentionBlock(768, 256, 8).parameters()) == 460544
    assert sum(p.numel() for p in fb.BilinearFusionModule(768, 256).parameters()) == 164864
    from mibf_net.attention import MultiHeadCrossAttention_v2
    assert sum(p.numel() for p in MultiHeadCrossAttention_v2(768, 1).parameters()) == 3543552


def test_oracle_kan_update_grid_matches_reference_vectors():
    """KANLinear.update_grid / KAN1.forward(update_grid=True) (reference kan1.py:167-212, 275-283): new knots, re-fitted
    coefficients and the layer output after the update, against vectors made by running the reference"""
    import numpy as np
    g = np.load(os.path.join(gc.GOLDEN, "kan_update_grid.npz"))
    x = torch.from_numpy(g["x"])
    lay = load_procedural(om.OKANLinear(16, 12), gc.SEED + 87)
    assert torch.allclose(lay(x), torch.from_numpy(g["out_before"]), atol=1e-5)
    lay.update_grid(x)
    assert torch.allclose(lay.grid, torch.from_numpy(g["grid"]), atol=1e-6)
    assert torch.allclose(lay.spline_weight, torch.from_numpy(g["spline_weight"]), atol=2e-4, rtol=1e-3)
    assert torch.allclose(lay(x), torch.from_numpy(g["out_after"]), atol=1e-4)


def test_pretrained_weight_files_load_into_the_towers(tmp_path, monkeypatch):
    """The reference's weight-file branches (encoder.py:44-52 `image_weights_path` / torchvision hub cache;
    BertModel.from_pretrained on a local HF directory holding model.safetensors or pytorch_model.bin, encoder.py:123,
    mibf_net/bert.py:9): the product towers end up with exactly the tensors in the files (no kernels run: CPU test)."""
    import encoder as product_encoder
    from transformers import BertConfig
    from transformers import BertModel as HFBert
    from hamspine.nn.bert import BertModel
    # --- HF directory written by transformers itself: model.safetensors, then pytorch_model.bin ------------------------
    hf = HFBert(BertConfig(**gc.TINY_BERT)).eval()
    for safe in (True, False):
        d = str(tmp_path / f"hf_{int(safe)}")
        if safe:
            hf.save_pretrained(d)                                          # this transformers writes model.safetensors
        else:                                                              # the older layout: config.json + pytorch_model.bin
            hf.config.save_pretrained(d)
            torch.save(hf.state_dict(), os.path.join(d, "pytorch_model.bin"))
        assert os.path.exists(os.path.join(d, "model.safetensors" if safe else "pytorch_model.bin"))
        assert not os.path.exists(os.path.join(d, "pytorch_model.bin" if safe else "model.safetensors"))
        pm = BertModel.from_pretrained(d)
        want = {k: v for k, v in hf.state_dict().items() if "position_ids" not in k}
        got = pm.state_dict()
        assert set(want) <= set(got)
        for k, v in want.items():
            assert torch.equal(got[k], v), k
    with pytest.raises(FileNotFoundError):
        BertModel.from_pretrained(str(tmp_path / "nowhere"))
    # --- torchvision-format ResNet .pth through image_weights_path, and through the hub cache (pretrained=True) ----------
    o = load_procedural(towers.oresnet("resnet18", num_classes=1000), 77)
    sd = o.state_dict()                                    # torchvision key layout (conv1, bn1, layer1.0.conv1, ..., fc)
    pth = str(tmp_path / "resnet18.pth")
    torch.save(sd, pth)
    enc = product_encoder.ImageEncoder(feature_dim=64, pretrained=False, weights_path=pth, backbone="resnet18")
    got = enc.state_dict()
    for k, v in sd.items():
        if k.startswith("fc."):
            continue                                       # the tower drops the classifier (encoder.py:54)
        assert torch.equal(got["model." + k], v), k
    assert torch.equal(got["stem.0.weight"], sd["conv1.weight"])         # alias key family of the reference
    with pytest.raises(FileNotFoundError):
        product_encoder.ImageEncoder(pretrained=False, weights_path=str(tmp_path / "missing.pth"), backbone="resnet18")
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "torch_home"))
    with pytest.raises(FileNotFoundError):                 # no cached ImageNet file and no network: refuse, never download
        product_encoder.ImageEncoder(pretrained=True, backbone="resnet18")
    cache = os.path.dirname(product_encoder._hub_cache_file("x"))
    os.makedirs(cache)
    name = product_encoder._BACKBONES["resnet18"][2]
    torch.save(sd, os.path.join(cache, name))
    enc2 = product_encoder.ImageEncoder(pretrained=True, backbone="resnet18")
    assert torch.equal(enc2.state_dict()["model.layer4.1.conv2.weight"], sd["layer4.1.conv2.weight"])
