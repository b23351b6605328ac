"""GPU parity tests of the ConNeXT path: ConvNeXt operators against plain torch f32
references of the same ops, the ConvNeXt tower and the full OurClassfierConvnextV2 against the golden vectors made
by running the reference (tests/golden/convnext_tiny.npz, e2e_connext.npz).

Tolerances: f32 mode 1e-4 relative to max|ref| on outputs/logits (argmax bit-exact), 1e-3 on gradients;
bf16 mode 3e-2 on outputs (6e-2 behind the unscaled attention), 8e-2 on gradient norms."""
import pytest
import torch
import torch.nn.functional as TF

import golden_cases as gc

pytestmark = pytest.mark.gpu

import hamspine  # noqa: E402
from hamspine import convnext_ops as X  # noqa: E402
from oracle import towers  # noqa: E402
from oracle.procedural import load_procedural  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _f32_mode():
    hamspine.set_compute_dtype("f32")
    yield
    hamspine.set_compute_dtype("bf16")


def _close(a, b, what, rtol, atol=2e-6):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + atol, f"{what}: max err {err:.3e} (scale {scale:.3e}, rtol {rtol})"


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("n,h,w,c,k", [(2, 14, 14, 64, 7), (3, 7, 5, 132, 7), (1, 9, 10, 8, 3), (2, 6, 6, 16, 5)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_dwconv_fwd_bwd(n, h, w, c, k, dt):
    x = torch.randn(n, c, h, w, generator=_g(1))
    wt = torch.randn(c, 1, k, k, generator=_g(2)) / k
    b = torch.randn(c, generator=_g(3))
    cot = torch.randn(n, c, h, w, generator=_g(4))
    if dt == torch.bfloat16:   # reference on the bf16-rounded inputs, so only accumulation / output rounding differ
        x, cot = x.bfloat16().float(), cot.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = TF.conv2d(xr, wr, br, padding=k // 2, groups=c)
    (yr * cot).sum().backward()
    xp = x.permute(0, 2, 3, 1).contiguous().to(DEV, dt).requires_grad_(True)
    wp, bp = wt.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yp = X.dwconv(xp, wp, bp)
    (yp.float() * cot.permute(0, 2, 3, 1).to(DEV)).sum().backward()
    tol = 2e-5 if dt == torch.float32 else 1e-2
    _close(yp.permute(0, 3, 1, 2), yr, "dwconv y", tol)
    _close(xp.grad.permute(0, 3, 1, 2), xr.grad, "dwconv dx", tol)
    _close(wp.grad, wr.grad, "dwconv dw", 1e-4 if dt == torch.float32 else 1e-4)
    _close(bp.grad, br.grad, "dwconv db", 1e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layerscale_and_patchify(dt):
    n, h, w, c = 3, 8, 6, 24
    u = torch.randn(n, h, w, c, generator=_g(1))
    res = torch.randn(n, h, w, c, generator=_g(2))
    gamma = torch.randn(c, generator=_g(3))
    rs = torch.tensor([0.0, 2.0, 2.0])
    cot = torch.randn(n, h, w, c, generator=_g(4))
    if dt == torch.bfloat16:
        u, res, cot = u.bfloat16().float(), res.bfloat16().float(), cot.bfloat16().float()
    ur, gr, rr = u.clone().requires_grad_(True), gamma.clone().requires_grad_(True), res.clone().requires_grad_(True)
    outr = rr + gr * rs[:, None, None, None] * ur
    (outr * cot).sum().backward()
    up, gp, rp = (t.to(DEV) for t in (u.to(dt), gamma, res.to(dt)))
    up.requires_grad_(True), gp.requires_grad_(True), rp.requires_grad_(True)
    outp = X.layer_scale_residual(up, gp, rp, rs.to(DEV))
    (outp.float() * cot.to(DEV)).sum().backward()
    tol = 1e-5 if dt == torch.float32 else 1e-2
    _close(outp, outr, "layerscale out", tol)
    _close(up.grad, ur.grad, "layerscale du", tol)
    _close(gp.grad, gr.grad, "layerscale dgamma", 1e-4)
    _close(rp.grad, rr.grad, "layerscale dres", tol)
    # patchify conv (stride == kernel), including the 3-channel image case and a ragged border (H % k != 0)
    for cin, cout, k, hh, ww in ((3, 16, 4, 18, 16), (24, 40, 2, 9, 8)):
        x = torch.randn(2, cin, hh, ww, generator=_g(5))
        wt = torch.randn(cout, cin, k, k, generator=_g(6)) / (k * cin ** 0.5)
        b = torch.randn(cout, generator=_g(7))
        if dt == torch.bfloat16:
            x, wt = x.bfloat16().float(), wt.bfloat16().float()
        xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = TF.conv2d(xr, wr, br, stride=k)
        c2 = torch.randn(yr.shape, generator=_g(8))
        if dt == torch.bfloat16:
            c2 = c2.bfloat16().float()
        (yr * c2).sum().backward()
        xp = x.permute(0, 2, 3, 1).contiguous().to(DEV, dt).requires_grad_(True)
        wp = wt.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bp = b.to(DEV).requires_grad_(True)
        yp = X.patch_conv(xp, wp, bp, k)
        (yp.float() * c2.permute(0, 2, 3, 1).to(DEV)).sum().backward()
        tol = 1e-4 if dt == torch.float32 else 2e-2
        _close(yp.permute(0, 3, 1, 2), yr, f"patch_conv{k} y", tol)
        _close(xp.grad.permute(0, 3, 1, 2), xr.grad, f"patch_conv{k} dx", tol)
        _close(wp.grad, wr.grad, f"patch_conv{k} dw", tol)
        _close(bp.grad, br.grad, f"patch_conv{k} db", tol)


def _product_tower(seed):
    from hamspine.nn.convnext import ConvNextConfig, ConvNextModel
    m = ConvNextModel(ConvNextConfig(**gc.CONNEXT_CFG))
    load_procedural(m, seed)
    return m.to(DEV)


def test_convnext_tower_matches_reference_vectors():
    fx = gc.load("convnext_tiny")
    m = _product_tower(gc.SEED + 300).train()
    x = fx["x"].to(DEV).requires_grad_(True)
    out = m(x).last_hidden_state
    assert out.shape == fx["out"].shape and out.is_contiguous(memory_format=torch.channels_last)
    _close(out, fx["out"], "convnext last_hidden_state", 1e-4)
    (out.float() * fx["cot"].to(DEV)).sum().backward()
    _close(x.grad, fx["dx"], "convnext dx", 1e-3)
    params = dict(m.named_parameters())
    for k, n in fx["gnorm"].items():
        _close(params[k].grad.norm(), n, f"convnext |grad {k}|", 1e-3, 1e-7)
    for k, g in fx["gw"].items():
        _close(params[k].grad, g, f"convnext grad {k}", 1e-3)
    assert sorted(k for k, p in params.items() if p.grad is None) == ["layernorm.bias", "layernorm.weight"]
    # pooled branch (ConvNextModel.pooler_output) against the oracle's final LayerNorm of the spatial mean
    o = load_procedural(towers.OConvNextModel(**gc.CONNEXT_CFG), gc.SEED + 300).eval()
    m.eval()
    with torch.no_grad():
        ref = o.layernorm(o(fx["x"]).mean(dim=(2, 3)))
        _close(m(fx["x"].to(DEV), pool=True).pooler_output, ref, "pooler_output", 1e-4)


def test_convnext_torchvision_layout_equals_hf_layout():
    """The fallback branch (torchvision `features`, reference ourmodel.py:49-62) runs the same arithmetic under
    torchvision's key names: map a seeded HF-layout state dict across and compare, then check stochastic depth."""
    from hamspine.nn.convnext import ConvNextFeatures
    hf = _product_tower(3).eval()
    dims, depths = gc.CONNEXT_CFG["hidden_sizes"], gc.CONNEXT_CFG["depths"]
    tv = ConvNextFeatures(dims, depths, stochastic_depth_prob=0.0)
    keys = set(tv.state_dict())
    for k in ("0.0.weight", "0.1.bias", "1.0.layer_scale", "1.0.block.0.weight", "1.0.block.2.weight", "1.0.block.3.bias",
              "1.0.block.5.weight", "2.0.weight", "2.1.weight", "5.1.block.5.bias", "6.1.bias", "7.0.layer_scale"):
        assert k in keys, k
    sd = hf.state_dict()
    mapped = {"0.0.weight": sd["embeddings.patch_embeddings.weight"], "0.0.bias": sd["embeddings.patch_embeddings.bias"],
              "0.1.weight": sd["embeddings.layernorm.weight"], "0.1.bias": sd["embeddings.layernorm.bias"]}
    names = {"dwconv": "block.0", "layernorm": "block.2", "pwconv1": "block.3", "pwconv2": "block.5"}
    for i, depth in enumerate(depths):
        if i > 0:
            for j in (0, 1):
                for leaf in ("weight", "bias"):
                    mapped[f"{2 * i}.{j}.{leaf}"] = sd[f"encoder.stages.{i}.downsampling_layer.{j}.{leaf}"]
        for d in range(depth):
            src = f"encoder.stages.{i}.layers.{d}."
            dst = f"{2 * i + 1}.{d}."
            mapped[dst + "layer_scale"] = sd[src + "layer_scale_parameter"].reshape(-1, 1, 1)
            for a, bname in names.items():
                for leaf in ("weight", "bias"):
                    mapped[dst + f"{bname}.{leaf}"] = sd[src + f"{a}.{leaf}"]
    tv.load_state_dict(mapped, strict=True)
    tv = tv.to(DEV).eval()
    x = torch.randn(2, 3, 64, 96, generator=_g(9)).to(DEV)
    with torch.no_grad():
        a, b = hf(x).last_hidden_state, tv(x)
    assert torch.equal(a, b)
    # stochastic depth ("row" mode): in train mode every sample's branch is either dropped or scaled by 1/(1-p)
    from hamspine.nn.convnext import CNBlock
    blk = CNBlock(16, 1.0, 0.5).to(DEV)
    xin = torch.randn(64, 5, 5, 16, generator=_g(10)).to(DEV)
    torch.manual_seed(0)
    blk.train()
    y_tr = blk(xin)
    blk.eval()
    y_ev = blk(xin)
    branch_tr, branch_ev = (y_tr - xin).flatten(1), (y_ev - xin).flatten(1)
    dropped = branch_tr.abs().amax(1) == 0
    assert 8 <= int(dropped.sum()) <= 56
    _close(branch_tr[~dropped], 2.0 * branch_ev[~dropped], "kept rows scaled by 1/(1-p)", 1e-5)


def _build_connext(tmp_path):
    from ConNexT.models.ourmodel import OurClassfierConvnextV2
    bdir = gc.save_bert_dir(gc.MIBF_BERT, str(tmp_path / "bert768"))
    cdir = gc.save_convnext_dir(gc.CONNEXT_CFG, str(tmp_path / "convnext"))
    m = OurClassfierConvnextV2(num_labels=5, pretrained=True, pretrained_path=cdir, bert_path=bdir)
    assert m._use_hf
    load_procedural(m, gc.SEED + 310)
    return m.to(DEV)


def test_connext_e2e_matches_reference_vectors(tmp_path):
    from hamspine import functional as F
    fx = gc.load("e2e_connext")
    images, ids, mask, labels = [t.to(DEV) for t in gc.connext_inputs()]
    m = _build_connext(tmp_path).train()
    logits = m({"input_ids": ids, "attention_mask": mask, "transformed_image": images})
    _close(logits, fx["logits"], "connext logits", 1e-4)
    assert torch.equal(logits.argmax(1).cpu(), fx["logits"].argmax(1))
    loss = F.cross_entropy(logits, labels)
    _close(loss, fx["loss"], "connext loss", 1e-4)
    loss.backward()
    params = dict(m.named_parameters())
    for k, n in fx["gnorm"].items():
        _close(params[k].grad.norm(), n, f"connext |grad {k}|", 5e-3, 1e-7)
    for k, g in fx["gw"].items():
        _close(params[k].grad, g, f"connext grad {k}", 2e-3, 1e-7)
    nograd = sorted(k for k, p in params.items() if p.grad is None)
    assert nograd == sorted(str(s) for s in fx["nograd"])
    # state-dict keys are the reference's (HF branch): image_encoder.* are transformers ConvNextModel names
    from oracle import models as om
    o = om.OConNeXT(5, gc.MIBF_BERT, gc.CONNEXT_CFG)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in o.state_dict().items()}


def test_connext_bf16_mode_close_to_reference(tmp_path):
    from hamspine import functional as F
    fx = gc.load("e2e_connext")
    images, ids, mask, labels = [t.to(DEV) for t in gc.connext_inputs()]
    hamspine.set_compute_dtype("bf16")
    m = _build_connext(tmp_path).train()
    logits = m({"input_ids": ids, "attention_mask": mask, "transformed_image": images})
    # the image->text attention is an UNSCALED softmax over 768-dim dot products (ourmodel.py:21-26): bf16 rounding of
    # the tower output moves its logits by O(1), so the bound is wider than for the scaled attentions (6e-2 + 3e-2)
    _close(logits, fx["logits"], "bf16 connext logits", 6e-2, 3e-2)
    loss = F.cross_entropy(logits, labels)
    _close(loss, fx["loss"], "bf16 connext loss", 6e-2)
    loss.backward()
    params = dict(m.named_parameters())
    # same sensitivity on the way back: the softmax weights of one sample sit near a switching point, which rescales
    # every upstream gradient together.  Check direction (cosine >= 0.95 on the stored gradients) and magnitude (40 %).
    floor = 1e-4 * max(float(n) for n in fx["gnorm"].values())   # analytic zeros (key.bias, single-key softmax) hold noise
    for k, n in fx["gnorm"].items():
        g = params[k].grad.norm().item()
        assert abs(g - n.item()) <= 0.4 * n.item() + floor, (k, g, n.item())
    for k, g in fx["gw"].items():
        if float(g.norm()) <= floor:
            continue
        cos = torch.nn.functional.cosine_similarity(params[k].grad.flatten().cpu().double(), g.flatten().double(), dim=0)
        assert cos >= 0.95, (k, float(cos))


def test_convnext_tower_bf16_close_to_reference():
    """tower alone in throughput mode (no unscaled attention behind it): 3e-2 on the output, 8e-2 on gradient norms"""
    fx = gc.load("convnext_tiny")
    hamspine.set_compute_dtype("bf16")
    m = _product_tower(gc.SEED + 300).train()
    out = m(fx["x"].to(DEV)).last_hidden_state
    assert out.dtype == torch.bfloat16
    _close(out, fx["out"], "bf16 convnext last_hidden_state", 3e-2)
    (out.float() * fx["cot"].to(DEV)).sum().backward()
    params = dict(m.named_parameters())
    bad = []
    for k, n in fx["gnorm"].items():
        g = params[k].grad.norm().item()
        if abs(g - n.item()) > 8e-2 * n.item() + 1e-6:
            bad.append((k, g, n.item()))
    assert len(bad) <= len(fx["gnorm"]) // 20, bad[:10]


def test_convnext_base_geometry_bf16_runs():
    """convnext-base-224 at the reference's input size, one training step in throughput mode (finite outputs / grads)."""
    from hamspine.nn.convnext import ConvNextConfig, ConvNextModel
    hamspine.set_compute_dtype("bf16")
    m = ConvNextModel(ConvNextConfig.base()).to(DEV).train()
    assert sum(p.numel() for p in m.parameters()) == 87566464
    x = torch.randn(4, 3, 224, 224, generator=_g(1)).to(DEV)
    out = m(x).last_hidden_state
    assert out.shape == (4, 1024, 7, 7) and out.dtype == torch.bfloat16
    out.float().square().mean().backward()
    for k, p in m.named_parameters():
        if k.startswith("layernorm."):
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
