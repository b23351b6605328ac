#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_tower_gpu.py tests/test_product_gpu.py -x -q > gpurun_out/c3_test.txt 2>&1; tail -5 gpurun_out/c3_test.txt
bash tools/iso_ab.sh base:HAMSPINE_CONV3=0 c3:HAMSPINE_CONV3=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run12.txt
