"""Per-workgroup timeline of one bf16 GEMM launch from in-kernel shader-clock stamps (hs_gemm_debug_stamps):
where a tile's time goes -- set-up, first DMA latency, K loop, epilogue -- and how the workgroups' starts spread.
usage: gemm_stamps.py kind M N K [cfg]     (run on the GPU box)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import gemm_bench as gb  # noqa: E402


def main():
    kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    cfg = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    bits = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    lib = gb.lib
    lib.hs_gemm_debug_stamps.argtypes = [C.c_void_p]
    lib.hs_gemm_debug(cfg, bits)
    fn = gb.gemm_case(kind, M, N, K)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = torch.zeros(6 * 65536, dtype=torch.int64, device="cuda")
    lib.hs_gemm_debug_stamps(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.hs_gemm_debug_stamps(None)
    st = buf.view(-1, 6).cpu()
    st = st[st[:, 0] != 0].double()
    n = st.shape[0]
    t0 = st[:, 0].min()
    us = lambda x: x / 2400.0     # the stamps tick with the shader clock (~2.4 GHz): lifetimes add up to the launch time
    names = ["set-up -> first DMA issued", "first DMA issued -> first tile landed", "K loop", "epilogue"]
    print(f"{kind} {M}x{N}x{K} cfg {cfg} bits {bits}: {n} workgroups (clocks differ between XCDs: only differences inside a workgroup are meaningful)")
    for i, nm in enumerate(names):
        d = st[:, i + 1] - st[:, i]
        print(f"  {nm:40s} mean {us(d.mean()):7.2f} us   p10 {us(d.quantile(0.1)):7.2f}   p90 {us(d.quantile(0.9)):7.2f}")
    d = st[:, 5] - st[:, 3]
    print(f"  {'  of which: up to the first fragment stored':40s} mean {us(d.mean()):7.2f} us   p10 {us(d.quantile(0.1)):7.2f}   p90 {us(d.quantile(0.9)):7.2f}")
    life = st[:, 4] - st[:, 0]
    print(f"  workgroup lifetime                       mean {us(life.mean()):7.2f} us")
    starts = (st[:, 0] - t0).sort().values
    print("  start times (us) of workgroups at 0/25/50/75/100 %: " + " ".join(f"{us(starts[int(q * (n - 1))]):.1f}" for q in (0, .25, .5, .75, 1)))


if __name__ == "__main__":
    main()
