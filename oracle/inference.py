"""ORACLE -- test infrastructure only (never imported by the product package).

CPU restatement of the reference's inference-side callers of the path:
  * test-time augmentation: scripts/predict.py:33-42 (variant list) and :63-70 (per-variant model calls, stacked mean);
  * Grad-CAM: analysis_tools.py:29-42 (forward / full-backward hooks at stage boundaries), :55-68 (one-hot backward of
    the arg-max class) and :78-93 (per-layer map, before the cv2 resize to image size).

Pinning: both are a few lines of torch / numpy whose semantics (flip, rot90, mean, hooks) are torch's own; the
reference files cannot be imported here (scripts/predict.py pulls in the data pipeline, analysis_tools.py needs cv2,
neither available), so these two restatements are checked by reading only: PARITY UNPINNED for this file.
"""
import numpy as np
import torch


def apply_tta(images, tta_transforms):
    variants = [images]
    for name in tta_transforms:
        if name == "hflip":
            variants.append(images.flip(-1))
        elif name == "vflip":
            variants.append(images.flip(-2))
        elif name == "rot90":
            variants.append(torch.rot90(images, k=1, dims=(-2, -1)))
    return variants


def predict_tta(model, images, input_ids, attention_mask, tabular=None, transforms=("hflip",)):
    with torch.no_grad():
        logits = [model(v, input_ids, attention_mask, tabular_input=tabular) for v in apply_tta(images, transforms)]
        return torch.stack(logits, dim=0).mean(dim=0)


class GradCamHooks:
    """activations / output-gradients of `layers` ({name: module}) for the arg-max (or given) class."""

    def __init__(self, layers):
        self.layers = dict(layers)
        self.activations, self.gradients = {}, {}
        self._handles = []
        for name, layer in self.layers.items():
            self._handles.append(layer.register_forward_hook(
                lambda m, i, o, name=name: self.activations.__setitem__(name, o)))
            self._handles.append(layer.register_full_backward_hook(
                lambda m, gi, go, name=name: self.gradients.__setitem__(name, go[0])))

    def run(self, model, images, input_ids, attention_mask, target=None, **kw):
        model.zero_grad()
        logits = model(images, input_ids, attention_mask, **kw)
        if target is None:
            target = logits.argmax(dim=1)
        one_hot = torch.zeros_like(logits)
        for i in range(logits.size(0)):
            one_hot[i][target[i]] = 1
        logits.backward(gradient=one_hot, retain_graph=True)
        return logits, target

    def remove(self):
        for h in self._handles:
            h.remove()


def cam_map(act, grad):
    """(C, H, W) numpy activation and gradient of one image -> (H, W) map in [0, 1] (analysis_tools.py:78-93)."""
    weights = np.mean(grad, axis=(1, 2))
    cam = np.zeros(act.shape[1:], dtype=np.float32)
    for j, w in enumerate(weights):
        cam += w * act[j]
    cam = np.maximum(cam, 0)
    if np.max(cam) > 0:
        cam = cam / np.max(cam)
    return cam
