#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "batchnorm or conv3" > gpurun_out/t16a.txt 2>&1; tail -8 gpurun_out/t16a.txt
timeout -k 10 900 python -m pytest tests/test_tower_gpu.py tests/test_product_gpu.py tests/test_fullsize_gpu.py tests/test_decisions_gpu.py -x -q > gpurun_out/t16b.txt 2>&1; tail -8 gpurun_out/t16b.txt
bash tools/iso_ab.sh base:HAMSPINE_BNB_FINISH=0 fin: 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run16.txt
