"""3x3 stride-1 convolutions of ResNet50 at batch 32 (forward, with the BatchNorm statistics finished in the launch):
microseconds per launch, cold operands (a 512 MB fill between launches).  Run once as is and once with HAMSPINE_CONV3=0
(generic implicit GEMM) for the A/B.   python tools/conv3_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import torch  # noqa: E402

from hamspine import _lib as L  # noqa: E402
from hamspine import raw  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SHAPES = [(32, 64, 56, 56, 64), (32, 128, 28, 28, 128), (32, 256, 14, 14, 256), (32, 512, 7, 7, 512)]


def main():
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    print(f"HAMSPINE_CONV3={os.environ.get('HAMSPINE_CONV3', '1')}")
    for Nb, C, H, W, K in SHAPES:
        x = (torch.randn(Nb, C, H, W, device=DEV) * 0.5).to(BF).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=DEV) * 0.1).to(BF).contiguous(memory_format=torch.channels_last)
        M, KK = Nb * H * W, 9 * C
        geom = raw.conv_geom(Nb, H, W, C, K, 3, 3, 1, 1)
        D = torch.empty((M, K), dtype=BF, device=DEV)
        gamma, beta = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        bnf = dict(gamma=gamma, beta=beta, running_mean=rm, running_var=rv, eps=1e-5, momentum=0.1)

        def run():
            raw.gemm(x, w, D, M, K, KK, a_kind=L.A_CONV, b_kind=L.B_KC, ldb=KK, ldd=K, geom=geom, bn_finish=bnf)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for cold in (True, False):
            acc = 0.0
            n = 20
            for _ in range(n):
                if cold:
                    flush.add_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                acc += e0.elapsed_time(e1)
            ts.append(acc / n * 1e3)
        fl = 2.0 * M * K * KK
        print(f"N={Nb} C={C:4d} {H:3d}x{W:<3d} K={K:4d}: cold {ts[0]:6.1f} us ({fl / ts[0] / 1e6:6.0f} TFLOP/s)   warm {ts[1]:6.1f} us ({fl / ts[1] / 1e6:6.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
