"""Seeded procedural weights and synthetic batches (inputs only -- no arithmetic of the path lives here).

Weights are a pure function of (parameter name, shape, seed): fixtures never store the multi-hundred-MB towers, and the
same function feeds the product (bench.py, tests) and -- re-exported by oracle/procedural.py -- the CPU oracle, so both
sides of a parity check start from identical values.
Synthetic batches follow SURVEY.md section 8(d) (reference data_loader.py:294-300,358: N(0,1) pixels, ids in
[1000, vocab) with CLS=101 first and SEP=102 last valid, right-padded attention masks).
"""
import hashlib
import math

import torch


def _gen(name, seed):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))


def procedural_tensor(name, like, seed):
    g = _gen(name, seed)
    shape = tuple(like.shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("grid", "mean", "std"):          # KAN knot grid / MoE Normal(mean, std) buffers keep their values
        return like.clone()
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=like.dtype)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if not like.dtype.is_floating_point:
        return like.clone()
    if like.dim() <= 1:
        is_scale = leaf in ("weight", "gamma") and ("norm" in name.lower() or "bn" in name.lower() or ".1." in name or
                                                     "downsample.1" in name or "downsampling_layer.0" in name)
        if is_scale:
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    if "embeddings" in name:
        return 0.05 * torch.randn(shape, generator=g)
    return torch.randn(shape, generator=g) * math.sqrt(1.0 / max(fan_in, 1))


def procedural_state_dict(module, seed):
    """state dict with seeded values for every key of `module` (aliased keys get identical tensors because
    aliases share storage: values are generated per storage, named by the first key that reaches it)."""
    out, seen = {}, {}
    for k, v in module.state_dict().items():
        ptr = (v.data_ptr(), tuple(v.shape)) if v.numel() > 0 else (id(v), ())
        if ptr in seen:
            out[k] = out[seen[ptr]]
            continue
        seen[ptr] = k
        out[k] = procedural_tensor(k, v, seed).to(v.dtype)
    return out


def load_procedural(module, seed):
    module.load_state_dict(procedural_state_dict(module, seed), strict=True)
    return module


def synthetic_batch(batch, image_hw=224, seq_len=128, vocab=30522, num_classes=7, seed=1234, min_len=16):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, 3, image_hw, image_hw, generator=g)
    low = min(1000, vocab // 2)
    ids = torch.randint(low, vocab, (batch, seq_len), generator=g)
    lens = torch.randint(min(min_len, seq_len), seq_len + 1, (batch,), generator=g)
    lens[0] = seq_len
    mask = (torch.arange(seq_len)[None, :] < lens[:, None]).long()
    ids[:, 0] = min(101, vocab - 1)
    ids[torch.arange(batch), lens - 1] = min(102, vocab - 1)
    ids = ids * mask
    labels = torch.randint(0, num_classes, (batch,), generator=g)
    return images, ids, mask, labels
