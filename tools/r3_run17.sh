#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_product_gpu.py tests/test_tower_gpu.py -x -q -k "attention or attn or bert or Bert" > gpurun_out/t17.txt 2>&1; tail -5 gpurun_out/t17.txt
bash tools/iso_ab.sh r16: r8:HAMSPINE_LN_BWD_ROWS=8 r4:HAMSPINE_LN_BWD_ROWS=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run17.txt
