#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
for rep in 1 2 3; do
  for cfg in "base:" "hi:HAMSPINE_TOWER_PRIORITY=-1"; do
    tag=${cfg%%:*}; v=${cfg#*:}
    ( for kv in ${v//,/ }; do export $kv; done
      echo "$tag $(timeout -k 10 300 python3 $R/tools/step_time.py --steps 30 --warmup 8 2>/dev/null)" ) | tee -a gpurun_out/prio.txt || exit 1
  done
done
timeout -k 10 300 python tools/tower_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/prio.txt
HAMSPINE_TOWER_PRIORITY=-1 timeout -k 10 300 python tools/tower_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/prio.txt
