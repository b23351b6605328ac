#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
F="amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|ProcessGroupNCCL"
for pre in "" "--pre"; do
for lo in 1 0; do
echo "pre='$pre' learn_order=$lo" | tee -a gpurun_out/ddp_timeline10.txt
HAMSPINE_DDP_LEARN_ORDER=$lo HS_TL_STEPS=20 timeout -k 10 300 python tools/tower_timeline.py c2 --ddp $pre 2>&1 | grep -v "$F" | grep "wall clock\|hs_.*bwd" | tee -a gpurun_out/ddp_timeline10.txt
done; done
for i in 1 2; do timeout -k 10 300 python bench.py --no-f32 --no-cpu-baseline 2>gpurun_out/b31.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['ddp_config'].items() if isinstance(v,dict)})"; done
