"""Per-workgroup timeline of the 3x3 halo-patch kernel (csrc/conv3.hip) from in-kernel clock stamps, ResNet50 shapes at batch 32.
python tools/conv3_stamps.py [plain]"""
import ctypes as C
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
    plain = len(sys.argv) > 1 and sys.argv[1] == "plain"
    lib = L.lib()
    lib.hs_gemm_debug_stamps.argtypes = [C.c_void_p]
    for Nb, Cc, H, W, K in SHAPES:
        x = (torch.randn(Nb, Cc, H, W, device=DEV) * 0.5).to(BF).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, Cc, 3, 3, device=DEV) * 0.1).to(BF).contiguous(memory_format=torch.channels_last)
        M, KK = Nb * H * W, 9 * Cc
        geom = raw.conv_geom(Nb, H, W, Cc, K, 3, 3, 1, 1)
        D = torch.empty((M, K), dtype=BF, device=DEV)
        bnf = None if plain else dict(gamma=torch.ones(K, device=DEV), beta=torch.zeros(K, device=DEV), running_mean=torch.zeros(K, device=DEV),
                                      running_var=torch.ones(K, device=DEV), eps=1e-5, momentum=0.1)

        def run():
            raw.gemm(x, w, D, M, K, KK, a_kind=L.A_CONV, b_kind=L.B_KC, ldb=KK, ldd=K, geom=geom, bn_finish=bnf)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        buf = torch.zeros(6 * 65536, dtype=torch.int64, device=DEV)
        lib.hs_gemm_debug_stamps(buf.data_ptr())
        run()
        torch.cuda.synchronize()
        lib.hs_gemm_debug_stamps(None)
        st = buf.view(-1, 6).cpu()
        st = st[st[:, 0] != 0].double()
        n = st.shape[0]
        us = lambda v: v / 2400.0     # the stamps tick with the shader clock (~2.4 GHz), as in tools/gemm_stamps.py
        print(f"C={Cc} {H}x{W} K={K} ({'plain' if plain else 'bn_finish'}): {n} workgroups")
        for nm, i, j in (("entry -> prologue issued", 0, 1), ("prologue issued -> patch + tap 0 landed", 1, 2), ("K walk", 2, 3),
                         ("statistics + ticket", 3, 5), ("staging + stores issued", 5, 4), ("lifetime (without the hand-off)", 0, 4)):
            d = st[:, j] - st[:, i]
            print(f"  {nm:42s} mean {us(d.mean()):7.2f} us   p10 {us(d.quantile(0.1)):7.2f}   p90 {us(d.quantile(0.9)):7.2f}")
        t0 = st[:, 0].min()
        starts = (st[:, 0] - t0).sort().values
        ends = (st[:, 4] - t0).sort().values
        print("  start times (us) at 0/25/50/75/100 %: " + " ".join(f"{us(starts[int(q * (n - 1))]):.1f}" for q in (0, .25, .5, .75, 1)) +
              "   ends: " + " ".join(f"{us(ends[int(q * (n - 1))]):.1f}" for q in (0, .5, 1)))


if __name__ == "__main__":
    main()
