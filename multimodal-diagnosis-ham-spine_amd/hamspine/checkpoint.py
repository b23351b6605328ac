"""Checkpoint formats either side of the path.

The product modules keep the reference's attribute tree, so `state_dict()` already has the reference's keys
(including the `image_encoder.stem.* == image_encoder.model.*` alias family) and plain `torch.save` / `torch.load`
files move between the two code bases in both directions.  This module restates the two policies the reference's
drivers wrap around that:

* `load_checkpoint`  - tolerant loader of reference mibf_net/predict_resnet.py:13-23 (`{"state_dict": ...}` wrapper,
  `module.` prefixes left by DataParallel / DDP, non-strict with a report of what did not match);
* `TopKCheckpoints`  - the "keep the 3 best epochs" policy of reference scripts/train.py:411-428 (same file names,
  strict `>` against the current worst, the evicted file is deleted, DataParallel / DDP wrappers are unwrapped).

Pretrained tower imports live with the towers: torchvision ResNet `.pth` files load through
`ImageEncoder(weights_path=...)` (encoder.py) and HF BERT directories through `BertModel.from_pretrained` (nn/bert.py).
"""
import os

import torch


def strip_module_prefix(state):
    # the reference removes every occurrence of "module." (str.replace), not just a leading one
    return {k.replace("module.", ""): v for k, v in state.items()}


def load_checkpoint(model, path, map_location="cpu", strict=False, verbose=True):
    """Load `path` into `model`; returns (missing_keys, unexpected_keys) like `load_state_dict`."""
    state = torch.load(path, map_location=map_location)
    if isinstance(state, dict) and "state_dict" in state:
        state = state["state_dict"]
    if isinstance(state, dict):
        state = strip_module_prefix(state)
    result = model.load_state_dict(state, strict=strict)
    missing, unexpected = list(result.missing_keys), list(result.unexpected_keys)
    if verbose and missing:
        print(f"Warning: missing keys: {missing[:10]}{'...' if len(missing) > 10 else ''}")
    if verbose and unexpected:
        print(f"Warning: unexpected keys: {unexpected[:10]}{'...' if len(unexpected) > 10 else ''}")
    return missing, unexpected


def unwrapped_state_dict(model):
    """state dict without wrapper prefixes (DataParallel / DDP / hamspine.ddp.DataParallel expose `.module`)."""
    return model.module.state_dict() if hasattr(model, "module") else model.state_dict()


class TopKCheckpoints:
    """Keep the `k` checkpoints with the highest validation accuracy in `output_dir`."""

    def __init__(self, output_dir, k=3):
        self.output_dir = output_dir
        self.k = k
        self.entries = []          # [(val_acc, path)], best first

    def update(self, model, epoch, val_acc):
        """`epoch` is 0-based as in the reference loop.  Returns the path written, or None when not in the top k."""
        if len(self.entries) >= self.k and not val_acc > min(self.entries, key=lambda e: e[0])[0]:
            return None
        os.makedirs(self.output_dir, exist_ok=True)
        path = os.path.join(self.output_dir, f"epoch_{epoch + 1}_val_acc_{val_acc:.2f}.pth")
        torch.save(unwrapped_state_dict(model), path)
        if len(self.entries) == self.k:
            worst = min(self.entries, key=lambda e: e[0])
            if os.path.exists(worst[1]):
                os.remove(worst[1])
            self.entries.remove(worst)
        self.entries.append((val_acc, path))
        self.entries.sort(key=lambda e: e[0], reverse=True)
        return path

    def accuracies(self):
        return [acc for acc, _ in self.entries]
