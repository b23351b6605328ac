#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
F="amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|ProcessGroupNCCL"
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py tests/test_tower_gpu.py -x -q > gpurun_out/t24.txt 2>&1; tail -4 gpurun_out/t24.txt
timeout -k 10 300 python tools/tower_timeline.py c2 --ddp 2>&1 | grep -v "$F" | tee gpurun_out/ddp_timeline3.txt
HAMSPINE_STAGED_WGRAD=0 timeout -k 10 300 python tools/tower_timeline.py c2 --ddp 2>&1 | grep -v "$F" | tee -a gpurun_out/ddp_timeline3.txt
timeout -k 10 300 python tools/tower_timeline.py c3 --ddp 2>&1 | grep -v "$F" | tee -a gpurun_out/ddp_timeline3.txt
timeout -k 10 600 python bench.py --no-f32 --no-cpu-baseline 2>gpurun_out/b24.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:(v['ms_per_step'],v['bucket_exchanges_in_backward'],v['bucket_exchanges_in_finish'],v['bucket_mb']) for k,v in d['ddp_config'].items() if isinstance(v,dict)})"
