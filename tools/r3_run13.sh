#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/xb_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xb_probe.txt
bash tools/iso_ab.sh base:HAMSPINE_XBLOCK_BN=0,HAMSPINE_CONV3=0 new: shadow:HAMSPINE_WEIGHT_SHADOWS=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run13.txt
