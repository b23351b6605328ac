#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/tower_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tower_timeline.txt
timeout -k 10 300 python tools/torch_ops_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/torch_ops.txt
timeout -k 10 900 python -m pytest tests/test_tower_gpu.py tests/test_product_gpu.py tests/test_train_loop_gpu.py -x -q > gpurun_out/t14.txt 2>&1; tail -5 gpurun_out/t14.txt
