#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8" 2>&1 | tail -5
echo "== stamps QKV rot / norot"
timeout -k 10 120 python tools/p8_stamps.py 4096 2304 768 0 > gpurun_out/r3_p8_stamps.txt 2>&1
timeout -k 10 120 python tools/p8_stamps.py 4096 2304 768 256 >> gpurun_out/r3_p8_stamps.txt 2>&1
timeout -k 10 120 python tools/p8_stamps.py 4096 4096 4096 0 >> gpurun_out/r3_p8_stamps.txt 2>&1
timeout -k 10 120 python tools/p8_stamps.py 4096 4096 4096 256 >> gpurun_out/r3_p8_stamps.txt 2>&1
cat gpurun_out/r3_p8_stamps.txt | grep -v amdgpu.ids
echo "== slope"
timeout -k 10 600 python tools/p8_slope.py > gpurun_out/r3_p8_slope.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3_p8_slope.txt
