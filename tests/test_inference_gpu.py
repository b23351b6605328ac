"""GPU parity tests of the callers either side of the path: fused test-time augmentation,
Grad-CAM at the stage boundaries, host->device input staging.  Checker: oracle/inference.py (+ torch on the CPU)."""
import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu

import hamspine  # noqa: E402
from hamspine import inference as inf  # noqa: E402
from hamspine import staging  # noqa: E402
from oracle import inference as oinf  # noqa: E402
from oracle import models as om  # noqa: E402
from oracle.procedural import load_procedural  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _f32_mode():
    hamspine.set_compute_dtype("f32")
    yield
    hamspine.set_compute_dtype("bf16")


def _pair(name, tmp_path):
    """(product on the GPU, oracle on the CPU) with identical procedural weights, both in eval mode."""
    import model as product_model
    seed, kw = gc.E2E_CASES[name]
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / "bert"))
    p = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                              **gc.E2E_COMMON, **kw)
    load_procedural(p, seed)
    o = load_procedural(om.OMultimodalBaselineModel(bert_cfg=gc.TINY_BERT, **gc.E2E_COMMON, **kw), seed)
    return p.to(DEV).eval(), o.eval(), kw


# ------------------------------------------------------------------------------------------------------
# test-time augmentation
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 3, 32, 32), (2, 3, 24, 40), (2, 3, 3, 16, 16)])
def test_tta_variants_bit_exact(shape):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(4))
    names = ["hflip", "vflip", "rot90", "unknown"] if shape[-1] == shape[-2] else ["vflip", "hflip"]
    got, V = inf.apply_tta(x.to(DEV), names)
    ref = torch.cat([v.contiguous() for v in oinf.apply_tta(x, names)], 0)
    assert V == len(oinf.apply_tta(x, names))
    assert torch.equal(got.cpu(), ref)
    if shape[-1] != shape[-2]:
        with pytest.raises(hamspine.HamspineError, match="square"):
            inf.apply_tta(x.to(DEV), ["rot90"])


@pytest.mark.parametrize("name,transforms", [
    ("e2e_basic_mlp", ["hflip"]),                              # the reference default (scripts/predict.py:64)
    ("e2e_concat_tabular", ["hflip", "vflip", "rot90"]),       # tabular rows are tiled with the variants
    ("e2e_gate_globallocal", ["vflip", "rot90"]),              # two feature passes per call + centre crop
])
def test_fused_tta_equals_per_variant_loop(name, transforms, tmp_path):
    p, o, kw = _pair(name, tmp_path)
    images, ids, mask, labels, tab = gc.e2e_inputs(kw)
    tab_c = tab if kw.get("tabular_enabled") else None
    ref = oinf.predict_tta(o, images, ids, mask, tab_c, transforms)
    with torch.no_grad():
        got = inf.predict_tta(p, images.to(DEV), ids.to(DEV), mask.to(DEV),
                              tabular_input=tab_c.to(DEV) if tab_c is not None else None, transforms=transforms)
        loop = [p(v.contiguous().to(DEV), ids.to(DEV), mask.to(DEV),
                  tabular_input=tab_c.to(DEV) if tab_c is not None else None) for v in oinf.apply_tta(images, transforms)]
    assert p.text_encoder.__class__.__name__ == "TextEncoder"          # the tower is back in place
    scale = ref.abs().max().item()
    assert (got.cpu() - ref).abs().max().item() <= 1e-4 * scale + 2e-6
    assert torch.equal(got.argmax(1).cpu(), ref.argmax(1))
    # against the product's own per-variant loop the fused batch differs only by f32 summation order of the mean
    assert (got - torch.stack(loop, 0).mean(0)).abs().max().item() <= 2e-6 * max(scale, 1.0)
    with pytest.raises(RuntimeError, match="no_grad"):
        inf.predict_tta(p, images.to(DEV), ids.to(DEV), mask.to(DEV))


# ------------------------------------------------------------------------------------------------------
# Grad-CAM
# ------------------------------------------------------------------------------------------------------
def _layers(m):
    enc = m.image_encoder       # the five boundaries of reference scripts/run_analysis.py:126-132
    return {"stem": enc.stem, "layer1": enc.layer1[-1], "layer2": enc.layer2[-1], "layer3": enc.layer3[-1],
            "layer4": enc.layer4[-1]}


def test_gradcam_maps_match_oracle(tmp_path):
    p, o, kw = _pair("e2e_basic_mlp", tmp_path)
    images, ids, mask, labels, tab = gc.e2e_inputs(kw)
    ho, hp = oinf.GradCamHooks(_layers(o)), oinf.GradCamHooks(_layers(p))
    lo, to = ho.run(o, images, ids, mask)
    lp, tp = hp.run(p, images.to(DEV), ids.to(DEV), mask.to(DEV))
    assert torch.equal(tp.cpu(), to)
    for name in _layers(o):
        act_o, grad_o = ho.activations[name].detach().numpy(), ho.gradients[name].numpy()
        assert tuple(hp.activations[name].shape) == act_o.shape and tuple(hp.gradients[name].shape) == grad_o.shape
        cams = inf.grad_cam(hp.activations[name], hp.gradients[name]).cpu().numpy()
        for i in range(images.shape[0]):
            ref = oinf.cam_map(act_o[i], grad_o[i])
            # maps are max-normalised to [0, 1]; the gradients reach the stem through 16 eval-mode conv+BN layers
            assert np.abs(cams[i] - ref).max() <= 2e-3, f"{name}[{i}]: {np.abs(cams[i] - ref).max():.3e}"
    ho.remove()
    hp.remove()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gradcam_kernel_against_numpy(dtype):
    g = torch.Generator().manual_seed(8)
    act = torch.randn(3, 96, 7, 5, generator=g).to(dtype)
    grad = torch.randn(3, 96, 7, 5, generator=g).to(dtype)
    grad[2] = -grad[2].abs() * (act[2] > 0)       # an all-negative map: stays zero, no division
    act[2] = act[2].abs()
    for layout in ("nhwc", "nchw"):
        a, gr = act.to(DEV), grad.to(DEV)
        if layout == "nhwc":
            a = a.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            gr = gr.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        cams = inf.grad_cam(a, gr).cpu().numpy()
        for i in range(3):
            ref = oinf.cam_map(act[i].float().numpy(), grad[i].float().numpy())
            assert np.abs(cams[i] - ref).max() <= 1e-5, layout
    assert cams[2].max() == 0.0


# ------------------------------------------------------------------------------------------------------
# input staging
# ------------------------------------------------------------------------------------------------------
def test_u8_normalise_bit_exact():
    u8 = torch.randint(0, 256, (5, 37, 29, 3), generator=torch.Generator().manual_seed(2), dtype=torch.uint8)
    mean = torch.tensor(staging.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(staging.IMAGENET_STD).view(1, 3, 1, 1)
    ref = (u8.permute(0, 3, 1, 2).float().div(255) - mean) / std      # ToTensor, then Normalize (sub_ / div_)
    got = staging.normalize_u8(u8.to(DEV))
    assert got.shape == ref.shape and got.is_contiguous()
    assert torch.equal(got.cpu(), ref)


def test_batch_stager_and_loss_meter():
    g = torch.Generator().manual_seed(3)
    batches = []
    for i in range(5):
        b = 4 if i < 4 else 2                                   # ragged last batch
        batches.append((torch.randn(b, 3, 16, 16, generator=g), torch.randint(0, 50, (b, 12), generator=g),
                        torch.ones(b, 12, dtype=torch.int64), None, torch.randint(0, 7, (b,), generator=g),
                        [f"img{i}_{j}" for j in range(b)]))
    for prefetch in (1, 3):
        seen = list(staging.BatchStager(batches, DEV, prefetch=prefetch))
        assert len(seen) == 5
        for ref, got in zip(batches, seen):
            assert isinstance(got, tuple) and len(got) == 6 and got[3] is None and got[5] == ref[5]
            for k in (0, 1, 2, 4):
                assert got[k].is_cuda and got[k].dtype == ref[k].dtype and torch.equal(got[k].cpu(), ref[k])
    # dict batches (the MIBF-Net loader's format) with decoded u8 images normalised on the device
    u8 = torch.randint(0, 256, (3, 8, 8, 3), generator=g, dtype=torch.uint8)
    out = list(staging.BatchStager([{"transformed_image": u8, "input_ids": torch.arange(6).view(3, 2), "image_id": ["a"]}],
                                   DEV, u8_images="transformed_image"))[0]
    assert out["transformed_image"].shape == (3, 3, 8, 8) and out["image_id"] == ["a"]
    assert torch.equal(out["transformed_image"].cpu(), staging.normalize_u8(u8.to(DEV)).cpu())
    with pytest.raises(hamspine.HamspineError):
        staging.BatchStager(batches, "cpu")
    meter = staging.LossMeter(DEV)
    vals = [0.5, 1.25, 2.0]
    for v in vals:
        meter.add(torch.tensor(v, device=DEV))
    assert meter.count == 3 and abs(meter.total() - sum(vals)) < 1e-6 and abs(meter.mean() - sum(vals) / 3) < 1e-6
    meter.reset()
    assert meter.total() == 0.0
