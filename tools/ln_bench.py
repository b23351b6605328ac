"""LayerNorm backward (csrc/norm.hip) on the BERT shape (4096 x 768, bf16): microseconds per call of the variants the step
uses, back to back on one stream (HIP events around 20 calls).  python tools/ln_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import torch  # noqa: E402

from hamspine import _lib as L  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def main():
    lib = L.lib()
    M, H = 4096, 768
    dy = torch.randn(M, H, device=DEV).to(BF)
    x = torch.randn(M, H, device=DEV).to(BF)
    gamma = torch.ones(H, device=DEV)
    mean = torch.zeros(M, device=DEV)
    rstd = torch.ones(M, device=DEV)
    dx = torch.empty_like(dy)
    dxd = torch.empty_like(dy)
    dg, db, dbias = (torch.empty(H, device=DEV) for _ in range(3))
    wsb = int(lib.hs_layernorm_bwd_ws_bytes(M, H))
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)

    def plain():
        L.check(lib.hs_layernorm_bwd(L.HS_BF16, P(dy), P(x), P(gamma), P(mean), P(rstd), P(dx), P(dg), P(db), P(ws), wsb, M, H, s), "ln")

    def pre(p, with_dxd=True, with_dx=True):
        def f():
            L.check(lib.hs_layernorm_bwd_pre(L.HS_BF16, P(dy), P(x), P(gamma), P(mean), P(rstd), P(dx) if with_dx else None, P(dg), P(db),
                                             P(dxd) if with_dxd else None, P(dbias), p, 1234, P(ws), wsb, M, H, s), "ln_pre")
        return f
    for name, fn in (("plain (dx, dgamma, dbeta)", plain), ("pre p=0.1 (+ dropped copy, bias gradient)", pre(0.1)), ("pre p=0", pre(0.0)),
                     ("pre p=0.1, no dx", pre(0.1, True, False))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        for cold in (False, True):
            tot = 0.0
            n = 20
            for _ in range(n):
                if cold:
                    flush.add_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                fn()
                fn()
                fn()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            print(f"{name:48s} {'cold first' if cold else 'warm'}: {tot / n / 4 * 1e3:6.1f} us per call (kernel + final)")


if __name__ == "__main__":
    main()
