#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
HAMSPINE_XBLOCK_BN=1 timeout -k 10 600 python -m pytest tests/test_tower_gpu.py tests/test_product_gpu.py tests/test_fullsize_gpu.py -x -q -k "(resnet or tower or e2e or full) and not executor_equals" 2>&1 | tail -4 &&
bash tools/iso_ab.sh base: xb:HAMSPINE_XBLOCK_BN=1 2>&1 | tee gpurun_out/r3_run11.txt
