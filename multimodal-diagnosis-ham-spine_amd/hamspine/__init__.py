"""hamspine -- MI355X-native (gfx950) implementation of the per-batch multimodal forward/backward
hot path of IamJerryXu/Multimodal-Diagnosis-HAM-Spine.

Python host on PyTorch-ROCm (tensors, streams, autograd graph, torch.distributed) calling the
hand-written HIP kernels of libhamspine_hip.so through the C ABI in include/hamspine.h.

Numeric modes (set_compute_dtype / env HAMSPINE_DTYPE):
  "bf16" (default)  bf16 activations + bf16-in MFMA with f32 accumulation (throughput mode)
  "f32"             f32 activations + exact f32-in MFMA (parity mode, 1e-4 vs the CPU reference)
Parameters, statistics, heads, losses and gradients of parameters are f32 in both modes.
"""
import os

import torch

from . import _lib
from ._lib import HamspineError  # noqa: F401

_COMPUTE = {"bf16": torch.bfloat16, "f32": torch.float32}
_compute_dtype = _COMPUTE[os.environ.get("HAMSPINE_DTYPE", "bf16")]


def set_compute_dtype(name):
    """name: 'bf16' | 'f32' | torch dtype."""
    global _compute_dtype
    if isinstance(name, str):
        _compute_dtype = _COMPUTE[name]
    elif name in (torch.bfloat16, torch.float32):
        _compute_dtype = name
    else:
        raise ValueError(f"unsupported compute dtype {name}")


def compute_dtype():
    return _compute_dtype


def library_path():
    return _lib.LIB_PATH


def require_device():
    """Raise unless the HIP library is built and an MI355X is visible (no silent fallbacks)."""
    l = _lib.lib()
    if not torch.cuda.is_available() or not l.hs_device_ok():
        raise HamspineError("hamspine needs a gfx950 (MI355X) device; none is visible")
