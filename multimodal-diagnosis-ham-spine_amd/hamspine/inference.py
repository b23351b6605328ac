"""Inference-side callers of the path.

* `apply_tta` / `predict_tta`: the reference evaluates each test-time-augmentation variant with its own model call
  (scripts/predict.py:33-42,63-70).  Here the variants are assembled into ONE batch by a gather kernel, the text tower
  runs once on the B original rows (its inputs do not change between variants) and its tokens are tiled, and the V
  logit sets are averaged by one kernel.  In eval mode every layer is row-independent (BatchNorm uses running
  statistics, folded into the convolutions), so the result equals the per-variant loop.
* `grad_cam`: the per-layer map of reference analysis_tools.py:78-93 (channel-mean of the gradient as weights,
  weighted activation sum, ReLU, max-normalise) as one kernel on the NHWC stage outputs the hooks capture.
"""
import ctypes as C

import torch

from . import _lib as L
from . import rt

_TTA_OPS = {"identity": 0, "hflip": 1, "vflip": 2, "rot90": 3}


def _ops_of(transforms):
    # unknown names are skipped, exactly like the reference's if/elif chain (scripts/predict.py:35-41)
    return [0] + [_TTA_OPS[t] for t in transforms if t in ("hflip", "vflip", "rot90")]


def apply_tta(images, transforms):
    """(B, ..., H, W) f32 -> (V*B, ..., H, W): variant-major stack of [images] + one entry per known transform."""
    rt.need_gpu(images)
    if images.dtype != torch.float32:
        raise TypeError("apply_tta: images must be float32 (the stem's input type)")
    ops = _ops_of(transforms)
    x = images.contiguous()
    H, W = x.shape[-2:]
    out = torch.empty((len(ops) * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    arr = (C.c_int32 * len(ops))(*ops)
    L.check(L.lib().hs_tta_expand(rt.p(x), rt.p(out), x.numel() // (H * W), H, W, arr, len(ops), rt.stream()),
            "hs_tta_expand")
    return out, len(ops)


def repeat_rows(t, V):
    """tile a contiguous tensor V times along dim 0 (text tokens, ids, masks, tabular rows)."""
    rt.need_gpu(t)
    t = t.contiguous()
    if V == 1:
        return t
    out = torch.empty((V * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    L.check(L.lib().hs_repeat(t.element_size(), rt.p(t), rt.p(out), t.numel(), V, rt.stream()), "hs_repeat")
    return out


def mean_over_variants(logits, V):
    """(V*B, K) f32 -> (B, K): torch.stack(logits_list, 0).mean(0) of the reference loop."""
    rt.need_gpu(logits)
    logits = logits.contiguous()
    if logits.dtype != torch.float32:
        raise TypeError("mean_over_variants: logits are f32 on this path")
    B = logits.shape[0] // V
    out = torch.empty((B,) + tuple(logits.shape[1:]), dtype=torch.float32, device=logits.device)
    L.check(L.lib().hs_group_mean(rt.p(logits), rt.p(out), V, out.numel(), rt.stream()), "hs_group_mean")
    return out


class _Replay(torch.nn.Module):
    """stands in for the text tower during the fused TTA pass: returns the tiled tokens computed once."""

    def __init__(self, tokens):
        super().__init__()
        self._tokens = tokens

    def forward(self, *a, **k):
        return self._tokens


def predict_tta(model, images, input_ids, attention_mask, tabular_input=None, transforms=("hflip",), **model_kwargs):
    """Averaged logits over [identity] + transforms with one pass of each tower (MultimodalBaselineModel signature).

    rot90 of non-square images changes the image shape, so such a variant cannot share the batch: it falls back
    to the reference's per-variant loop for that call."""
    if torch.is_grad_enabled():
        raise RuntimeError("predict_tta is an inference helper: call it under torch.no_grad() (reference predict.py:49)")
    H, W = images.shape[-2:]
    if "rot90" in transforms and H != W:
        call = lambda im: model(im, input_ids, attention_mask, tabular_input=tabular_input, **model_kwargs)  # noqa: E731
        outs = [call(images)]
        for t in transforms:
            if t in ("hflip", "vflip"):
                outs.append(call(apply_tta(images, [t])[0][images.shape[0]:]))
            elif t == "rot90":
                outs.append(call(torch.rot90(images, 1, (-2, -1)).contiguous()))
        return mean_over_variants(torch.cat(outs, 0), len(outs))
    batch, V = apply_tta(images, transforms)
    if V == 1:
        return model(images, input_ids, attention_mask, tabular_input=tabular_input, **model_kwargs)
    tokens = repeat_rows(model.text_encoder(input_ids, attention_mask), V)
    ids_v, mask_v = repeat_rows(input_ids, V), repeat_rows(attention_mask, V)
    tab_v = repeat_rows(tabular_input, V) if tabular_input is not None else None
    text_tower = model._modules["text_encoder"]
    model._modules["text_encoder"] = _Replay(tokens)
    try:
        logits = model(batch, ids_v, mask_v, tabular_input=tab_v, **model_kwargs)
    finally:
        model._modules["text_encoder"] = text_tower
    return mean_over_variants(logits, V)


def grad_cam(activation, gradient):
    """(B, C, H, W) stage output and its gradient (as the reference's hooks capture them) -> (B, H, W) f32 maps in
    [0, 1].  Both tensors are NHWC in memory on this path; other layouts are brought there first."""
    rt.need_gpu(activation, gradient)
    if activation.shape != gradient.shape or activation.dim() != 4:
        raise ValueError("grad_cam: activation and gradient must both be (B, C, H, W)")
    B, Cc, H, W = activation.shape
    dt = activation.dtype
    a = activation.detach().permute(0, 2, 3, 1).contiguous()       # no copy when the memory already is NHWC
    g = gradient.detach().to(dt).permute(0, 2, 3, 1).contiguous()
    cam = torch.empty((B, H, W), dtype=torch.float32, device=a.device)
    L.check(L.lib().hs_gradcam(rt.hs_dtype(dt), rt.p(a), rt.p(g), rt.p(cam), B, Cc, H * W, rt.stream()), "hs_gradcam")
    return cam
