"""Measurement aid: what the vendor library (torch.mm -> hipBLASLt / rocBLAS) does on the BERT-base GEMM shapes of the C2 step,
beside the same shapes through hs_gemm (tools/gemm_sweep.py, profiles/*launch_table*).  Not part of the product path.
  python tools/blas_reference.py            (on the GPU box)"""
import torch

SHAPES = [(4096, 3072, 768), (4096, 768, 3072), (4096, 2304, 768), (4096, 768, 2304), (4096, 768, 768),
          (768, 3072, 4096), (3072, 768, 4096), (2304, 768, 4096)]


FLUSH = None


def bench_cold(M, N, K, nt, reps=12):
    """one event pair per launch, a 768 MB fill between launches (the setting of tools/gemm_sweep.py)"""
    global FLUSH
    dev = "cuda"
    if FLUSH is None:
        FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    B = torch.randn((N, K) if nt else (K, N), device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = (lambda: torch.mm(A, B.t(), out=out)) if nt else (lambda: torch.mm(A, B, out=out))
    for _ in range(3):
        f()
    tot = 0.0
    for _ in range(reps):
        FLUSH.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    us = tot / reps
    return us, 2.0 * M * N * K / us / 1e6


def bench(M, N, K, nt, reps=40, rot=8):
    dev = "cuda"
    A = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(rot)]
    B = [torch.randn((N, K) if nt else (K, N), device=dev, dtype=torch.bfloat16) for _ in range(rot)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = (lambda i: torch.mm(A[i % rot], B[i % rot].t(), out=out)) if nt else (lambda i: torch.mm(A[i % rot], B[i % rot], out=out))
    for i in range(5):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    print("torch", torch.__version__, torch.cuda.get_device_name(0))
    print(f"{'M':>6} {'N':>6} {'K':>6}  layout      us   TFLOP/s   (back-to-back launches, operands rotated over 8 buffers)")
    import sys
    cold = "--cold" in sys.argv
    if cold:
        print("(cold: one event pair per launch -- includes the pair's own ~3-10 us -- and a 768 MB fill between launches)")
    for M, N, K in SHAPES:
        for nt in (True, False):
            us, tf = (bench_cold if cold else bench)(M, N, K, nt)
            print(f"{M:6d} {N:6d} {K:6d}  {'nt' if nt else 'nn'}     {us:7.1f}   {tf:7.0f}")
