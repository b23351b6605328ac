#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/tower_timeline.py c2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ddp_timeline.txt
timeout -k 10 300 python tools/tower_timeline.py c2 --ddp 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ddp_timeline.txt
timeout -k 10 300 python tools/tower_timeline.py c3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ddp_timeline.txt
timeout -k 10 300 python tools/tower_timeline.py c3 --ddp 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ddp_timeline.txt
