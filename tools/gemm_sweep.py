"""Tile-config x split-K sweep of the bf16 GEMM core on the BERT-base shapes of the C2 step, cold caches (a 768 MB fill
between launches, like the neighbours of a GEMM inside a step).  Prints microseconds per launch and the best choice per
shape: the input of gemm.hip's tile / split heuristics.   python tools/gemm_sweep.py [--warm]"""
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
lib.hs_prof_enable.argtypes = [C.c_int32]
lib.hs_prof_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
DEV, BF = "cuda", torch.bfloat16
CFGS = {0: "128x128", 1: "128x64", 2: "64x64", 4: "256x128", 5: "128x128k32", 6: "256x128k32"}
FLUSH = None


def timeit(fn, iters=12):
    global FLUSH
    cold = "--warm" not in sys.argv
    if cold and FLUSH is None:
        FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    fl, ms, cnt = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.hs_prof_enable(1)
    for _ in range(iters):
        if cold:
            lib.hs_prof_enable(0)
            FLUSH.add_(1)
            lib.hs_prof_enable(1)
        fn()
    lib.hs_prof_collect(fl, ms, cnt)
    lib.hs_prof_enable(0)
    return sum(ms) / max(sum(cnt), 1) * 1e3     # us


def case(kind, M, N, K):
    if kind == "nt":
        A, B = torch.randn(M, K, device=DEV).to(BF), torch.randn(N, K, device=DEV).to(BF)
        kw = dict(a_kind=L.A_KC, b_kind=L.B_KC, lda=K, ldb=K)
    elif kind == "nn":
        A, B = torch.randn(M, K, device=DEV).to(BF), torch.randn(K, N, device=DEV).to(BF)
        kw = dict(a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N)
    else:
        A, B = torch.randn(K, M, device=DEV).to(BF), torch.randn(K, N, device=DEV).to(BF)
        kw = dict(a_kind=L.A_RC, b_kind=L.B_RC, lda=M, ldb=N)
    D = torch.empty(M, N, device=DEV, dtype=torch.float32 if kind == "tn" else BF)
    return lambda split: raw.gemm(A, B, D, M, N, K, split_k=split, **kw)


def main():
    shapes = [("nt", 4096, 2304, 768), ("nt", 4096, 768, 768), ("nt", 4096, 3072, 768), ("nt", 4096, 768, 3072),
              ("nn", 4096, 3072, 768), ("nn", 4096, 768, 3072), ("nn", 4096, 768, 2304), ("nn", 4096, 768, 768),
              ("tn", 3072, 768, 4096), ("tn", 768, 3072, 4096), ("tn", 2304, 768, 4096), ("tn", 768, 768, 4096)]
    splits = (1, 2, 4, 8)
    if "--resnet-wgrad" in sys.argv:      # 1x1 weight gradients of ResNet50 at batch 32: (Cout, Cin, N*H*W)
        shapes = [("tn", 256, 64, 100352), ("tn", 64, 256, 100352), ("tn", 128, 256, 100352), ("tn", 512, 128, 25088),
                  ("tn", 128, 512, 25088), ("tn", 1024, 256, 6272), ("tn", 256, 1024, 6272), ("tn", 2048, 512, 1568)]
        splits = (1, 4, 8, 16, 32, 64, 128, 256)
    if "--fill" in sys.argv:              # does a K = 4096 weight-gradient GEMM get faster per FLOP when the grid fills the chip?
        shapes = [("nt", 3072, 768, 4096), ("nt", 6912, 768, 4096), ("nt", 12288, 768, 4096), ("nt", 768, 3072, 4096),
                  ("nt", 2304, 768, 4096), ("nt", 768, 768, 4096)]
        splits = (1, 4)
    only = [a for a in sys.argv[1:] if a in ("nt", "nn", "tn")]
    for kind, M, N, K in shapes:
        if only and kind not in only:
            continue
        fn = case(kind, M, N, K)
        fl = 2.0 * M * N * K
        best = None
        cells = []
        for cfg, name in CFGS.items():
            for split in splits:
                if K // split < 384 or (split > 8 and (M // 64) * (N // 64) * split > 4096):
                    continue
                lib.hs_gemm_debug(cfg, 0)
                try:
                    us = timeit(lambda: fn(split))
                except Exception as e:  # noqa: BLE001
                    cells.append(f"{name}/s{split}: ERR")
                    continue
                cells.append(f"{name}/s{split}: {us:6.1f}")
                if best is None or us < best[0]:
                    best = (us, name, split)
        lib.hs_gemm_debug(-1, 0)
        auto = timeit(lambda: fn(1))
        print(f"{kind} M={M} N={N} K={K}: auto {auto:.1f} us ({fl / auto / 1e6:.0f} TF) | best {best[1]}/s{best[2]} {best[0]:.1f} us "
              f"({fl / best[0] / 1e6:.0f} TF)", flush=True)
        print("    " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
