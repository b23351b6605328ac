#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/ln_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ln_bench.txt
HAMSPINE_LN_WAVES=4 timeout -k 10 200 python tools/ln_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ln_bench.txt
timeout -k 10 600 python -m pytest tests/test_product_gpu.py tests/test_tower_gpu.py -x -q -k "layernorm or layer_norm or bert or Bert or ln_" > gpurun_out/t26.txt 2>&1; tail -3 gpurun_out/t26.txt
