"""Steady-state rate and ramp of the GEMM bodies: a 4096 x 4096 output (256 tiles of 256 x 256: one per CU) at growing K gives
microseconds per 64-deep K tile (slope) and the per-launch ramp (intercept); the vendor
library (torch.mm -> hipBLASLt) on the same operands as a mark.  Warm operands, random data, interleaved rounds.
   python tools/p8_slope.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import torch  # noqa: E402

from hamspine import _lib as L  # noqa: E402
from hamspine import raw  # noqa: E402

lib = L.lib()
lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
DEV, BF = "cuda", torch.bfloat16


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3       # us per launch, back to back (launch gaps included on every arm)


def main():
    arms = [("p8-256", 7, 0), ("p8-128", 9, 0), ("128x64", 1, 0), ("256x128", 4, 0), ("hipBLASLt", None, 0)]
    for (M, N) in ((4096, 4096), (4096, 3072), (4096, 2304), (8192, 8192)):
        print(f"--- output {M} x {N}")
        print(f"{'K':>6s} " + " ".join(f"{n:>16s}" for n, _, _ in arms) + "    (us / TFLOP/s)")
        rows = {}
        for K in (256, 512, 768, 1536, 3072, 6144):
            A, B = torch.randn(M, K, device=DEV).to(BF), torch.randn(N, K, device=DEV).to(BF)
            D = torch.empty(M, N, device=DEV, dtype=BF)
            Bt = B.t()
            cells = []
            for name, cfg, abl in arms:
                if cfg is None:
                    fn = lambda: torch.mm(A, Bt, out=D)
                else:
                    def fn(cfg=cfg, abl=abl):
                        raw.gemm(A, B, D, M, N, K, lda=K, ldb=K)
                    lib.hs_gemm_debug(cfg, abl)
                best = min(timed(fn) for _ in range(3))
                lib.hs_gemm_debug(-1, 0)
                rows.setdefault(name, []).append((K, best))
                cells.append(f"{best:8.1f}/{2.0 * M * N * K / best / 1e6:6.0f}")
            print(f"{K:6d} " + " ".join(f"{c:>16s}" for c in cells), flush=True)
        for name, pts in rows.items():
            (k0, t0), (k1, t1) = pts[2], pts[-1]
            slope = (t1 - t0) / ((k1 - k0) / 64)
            print(f"   {name:14s}: {slope:6.3f} us per K tile of 64, intercept {t0 - slope * k0 / 64:6.2f} us")


if __name__ == "__main__":
    main()
