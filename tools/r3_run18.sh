#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_tower_gpu.py tests/test_fullsize_gpu.py tests/test_train_loop_gpu.py tests/test_ddp_gpu.py -x -q > gpurun_out/t18.txt 2>&1; tail -6 gpurun_out/t18.txt
bash tools/iso_ab.sh old:HAMSPINE_WGRAD_LAYERS=1,HAMSPINE_P8_WGRAD=0 pairgen:HAMSPINE_P8_WGRAD=0 new: three:HAMSPINE_WGRAD_LAYERS=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run18.txt
