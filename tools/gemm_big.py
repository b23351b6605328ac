"""Calibration of the GEMM core against the guide's 4096^3 ladder: TFLOP/s per tile config on large square shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_bench as gb
for kind, M, N, K in (("nt", 4096, 4096, 4096), ("nt", 8192, 8192, 8192), ("nt", 4096, 4096, 768), ("nt", 16384, 3072, 768)):
    fn = gb.gemm_case(kind, M, N, K)
    row = []
    for cfg in (0, 1, 2, 4, 5, 6):
        gb.lib.hs_gemm_debug(cfg, 0)
        row.append(2.0 * M * N * K / gb.timeit(fn, 10) / 1e12)
    gb.lib.hs_gemm_debug(-1, 0)
    print(f"{kind} {M}x{N}x{K}: " + "  ".join(f"{n}={v:6.1f}" for n, v in zip(("128x128", "128x64", "64x64", "256x128", "128x128k32", "256x128k32"), row)))
