#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_product_gpu.py tests/test_train_loop_gpu.py tests/test_tower_gpu.py -x -q -k "adam or Adam or optim or loop or shadow or sgd" > gpurun_out/t21.txt 2>&1; tail -4 gpurun_out/t21.txt
bash tools/iso_ab.sh b256:HAMSPINE_ADAM_BLOCKS=256 b1024: b2048:HAMSPINE_ADAM_BLOCKS=2048 b512:HAMSPINE_ADAM_BLOCKS=512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run21.txt
