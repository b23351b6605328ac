#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
F="amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|ProcessGroupNCCL"
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py -x -q > gpurun_out/t27.txt 2>&1; tail -3 gpurun_out/t27.txt
timeout -k 10 300 python tools/tower_timeline.py c2 --ddp 2>&1 | grep -v "$F" | tee gpurun_out/ddp_timeline5.txt
timeout -k 10 300 python tools/tower_timeline.py c3 --ddp 2>&1 | grep -v "$F" | tee -a gpurun_out/ddp_timeline5.txt
timeout -k 10 600 python bench.py --no-f32 --no-cpu-baseline 2>gpurun_out/b27.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:(v['ms_per_step'],v['bucket_exchanges_in_backward'],v['bucket_exchanges_in_finish']) for k,v in d['ddp_config'].items() if isinstance(v,dict)})"
