"""Where one K tile of the pipelined GEMM body spends its cycles: shader-clock stamps around every segment of the middle K
tile, waves 0 (early group) and 4 (late group) of every workgroup (diagnostic kernel build, hs_gemm_debug_stamps).
   python tools/p8_stamps.py M N K [ablate]"""
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
lib.hs_gemm_debug_stamps.argtypes = [C.c_void_p]
DEV, BF = "cuda", torch.bfloat16
NAMES = ["p0 reads issued+returned", "p0 dma+vmcnt+barrier", "p0 16 MFMA issued", "p0 barrier",
         "p1 reads", "p1 dma+vmcnt+barrier", "p1 MFMA", "p1 barrier",
         "p2 reads", "p2 dma+barrier", "p2 MFMA", "p2 barrier",
         "p3 dma+vmcnt+barrier", "p3 MFMA"]


def main():
    M, N, K = (int(x) for x in sys.argv[1:4])
    abl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    A, B = torch.randn(M, K, device=DEV).to(BF), torch.randn(N, K, device=DEV).to(BF)
    D = torch.empty(M, N, device=DEV, dtype=BF)
    lib.hs_gemm_debug(7, abl)
    for _ in range(3):
        raw.gemm(A, B, D, M, N, K, lda=K, ldb=K)
    torch.cuda.synchronize()
    nwg = (M // 256) * (N // 256)
    buf = torch.zeros(nwg * 2 * 20, dtype=torch.int64, device=DEV)
    lib.hs_gemm_debug_stamps(buf.data_ptr())
    raw.gemm(A, B, D, M, N, K, lda=K, ldb=K)
    torch.cuda.synchronize()
    lib.hs_gemm_debug_stamps(None)
    lib.hs_gemm_debug(-1, 0)
    st = buf.view(nwg, 2, 20).cpu().double()
    print(f"p8-256 {M}x{N}x{K} ablate {abl}: {nwg} workgroups, stamps per wave: {int(st[0, 0, 17])}; cycles (median over workgroups)")
    for grp in (0, 1):
        d = st[:, grp, 1:15] - st[:, grp, 0:14]
        tot = st[:, grp, 14] - st[:, grp, 0]
        print(f"  group {grp} (wave {grp * 4}): one K tile = {tot.median():.0f} cycles (p10 {tot.quantile(0.1):.0f}, p90 {tot.quantile(0.9):.0f})")
        for i, nm in enumerate(NAMES):
            print(f"     {nm:28s} {d[:, i].median():7.0f}   (p90 {d[:, i].quantile(0.9):7.0f})")


if __name__ == "__main__":
    main()
