"""A/B of the GEMM bodies on the BERT-base shapes of the C2 step (4096 tokens), interleaved rounds in ONE process (guide rule
24), random data (rule 25), warm and cold caches.  Prints microseconds per launch (HIP events inside hs_gemm, the bracket's
own ~3 us included on every arm) and TFLOP/s.   python tools/p8_bench.py [--cold] [--shapes fwd|all]"""
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
CFGS = {-1: "auto", 1: "128x64", 2: "64x64", 5: "128x128k32", 4: "256x128", 7: "p8-256", 8: "p8-256x128", 9: "p8-128"}
COLD = "--cold" in sys.argv
FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device=DEV) if COLD else None


def one(fn, cfg, iters):
    lib.hs_gemm_debug(cfg, 0)
    fl, ms, cnt = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.hs_prof_enable(1)
    for _ in range(iters):
        if COLD:
            lib.hs_prof_enable(0)
            FLUSH.add_(1)
            lib.hs_prof_enable(1)
        fn()
    lib.hs_prof_collect(fl, ms, cnt)
    lib.hs_prof_enable(0)
    lib.hs_gemm_debug(-1, 0)
    return sum(ms) / max(sum(cnt), 1) * 1e3


def main():
    shapes = [("QKV fwd", 4096, 2304, 768, {}), ("FFN1 fwd", 4096, 3072, 768, {"gelu": 1}), ("FFN2 fwd", 4096, 768, 3072, {}),
              ("attn-out", 4096, 768, 768, {}), ("dQKV", 4096, 768, 2304, {}),
              ("wgrad FFN", 3072, 768, 4096, {"f32": 1}), ("wgrad QKV", 2304, 768, 4096, {"f32": 1}), ("wgrad AO", 768, 768, 4096, {"f32": 1}),
              ("wgrad FFN2", 768, 3072, 4096, {"f32": 1})]
    print(f"{'shape':12s} {'M':>5s} {'N':>5s} {'K':>5s} | " + " ".join(f"{n:>12s}" for n in CFGS.values()) + "   (us median of 5 rounds x 10; " + ("cold" if COLD else "warm") + ")")
    for name, M, N, K, opt in shapes:
        A, B = torch.randn(M, K, device=DEV).to(BF), torch.randn(N, K, device=DEV).to(BF)
        D = torch.empty(M, N, device=DEV, dtype=torch.float32 if opt.get("f32") else BF)
        kw = {}
        if opt.get("gelu"):
            kw = dict(bias=torch.randn(N, device=DEV), act=L.ACT_GELU, preact=torch.empty_like(D))
        fn = lambda: raw.gemm(A, B, D, M, N, K, lda=K, ldb=K, **kw)
        res = {c: [] for c in CFGS}
        ok = {}
        for c in CFGS:
            bm, bn = {7: (256, 256), 8: (256, 128), 9: (128, 128)}.get(c, (1, 1))
            ok[c] = M % bm == 0 and N % bn == 0
        for c in CFGS:
            if ok[c]:
                one(fn, c, 3)
        for r in range(5):
            for c in CFGS:
                if ok[c]:
                    res[c].append(one(fn, c, 10))
        cells = []
        for c in CFGS:
            if not ok[c]:
                cells.append(f"{'-':>12s}")
            else:
                us = sorted(res[c])[2]
                cells.append(f"{us:6.1f}/{2.0 * M * N * K / us / 1e6:5.0f}")
        print(f"{name:12s} {M:5d} {N:5d} {K:5d} | " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
