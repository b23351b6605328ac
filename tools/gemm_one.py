"""Run one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py kind M N K [cfg]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import gemm_bench as gb  # noqa: E402

kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = int(sys.argv[5]) if len(sys.argv) > 5 else -1
gb.lib.hs_gemm_debug(cfg, int(sys.argv[6]) if len(sys.argv) > 6 else 0)
fn = gb.gemm_case(kind, M, N, K)
for _ in range(5):
    fn()
torch.cuda.synchronize()
