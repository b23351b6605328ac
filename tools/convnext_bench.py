"""Timing of one ConvNeXt-base training step (forward + backward) on synthetic 224px images (measurement helper)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import hamspine  # noqa: E402
from hamspine.nn.convnext import ConvNextConfig, ConvNextModel  # noqa: E402


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    hamspine.set_compute_dtype(os.environ.get("HAMSPINE_DTYPE", "bf16"))
    m = ConvNextModel(ConvNextConfig.base()).cuda().train()
    x = torch.randn(bs, 3, 224, 224, device="cuda")
    for i in range(2 + steps):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = m(x).last_hidden_state
        out.float().mean().backward()
        for p in m.parameters():
            p.grad = None
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"convnext-base bs{bs} fwd+bwd: {dt * 1e3:.2f} ms/step  ({bs / dt:.0f} img/s)")


if __name__ == "__main__":
    main()
