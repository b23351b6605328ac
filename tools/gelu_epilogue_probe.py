"""What the GELU epilogue costs inside the FFN up-projection GEMM (4096 x 3072 x 768, bf16, phase-pipelined body): the same launch
with bias only / bias + GELU + pre-activation copy / bias + pre-activation copy, cold operands.  python tools/gelu_epilogue_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import torch  # noqa: E402

from hamspine import _lib as L  # noqa: E402
from hamspine import raw  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def main():
    M, N, K = 4096, 3072, 768
    A = torch.randn(M, K, device=DEV).to(BF)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(BF)
    b = torch.randn(N, device=DEV)
    D = torch.empty(M, N, dtype=BF, device=DEV)
    P = torch.empty(M, N, dtype=BF, device=DEV)
    dy = torch.randn(M, N, device=DEV).to(BF)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    cases = {
        "bias": lambda: raw.gemm(A, W, D, M, N, K, lda=K, ldb=K, bias=b),
        "bias + GELU + preact copy (FFN1 forward)": lambda: raw.gemm(A, W, D, M, N, K, lda=K, ldb=K, bias=b, act=L.ACT_GELU, preact=P),
        "x GELU'(preact) (FFN2 data gradient)": lambda: raw.gemm(A, W, D, M, N, K, lda=K, ldb=K, mul_mode=L.MUL_GELU_GRAD, mul_src=P),
        "x (src > 0) (same load, no transcendental)": lambda: raw.gemm(A, W, D, M, N, K, lda=K, ldb=K, mul_mode=L.MUL_RELU_MASK, mul_src=P),
        "bias + preact copy": lambda: raw.gemm(A, W, D, M, N, K, lda=K, ldb=K, bias=b, preact=P),
        "plain": lambda: raw.gemm(A, W, D, M, N, K, lda=K, ldb=K),
    }
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tot, n = 0.0, 20
        for _ in range(n):
            flush.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print(f"{name:44s} {tot / n * 1e3:6.1f} us (cold, incl. ~3 us of event bracket)")


if __name__ == "__main__":
    main()
