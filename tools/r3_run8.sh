#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "streaming" 2>&1 | tail -15
echo "== pw bench"
timeout -k 10 300 python tools/pw_bench.py > gpurun_out/r3_pw_bench.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3_pw_bench.txt
echo "== model tests"
timeout -k 10 900 python -m pytest tests/test_product_gpu.py tests/test_tower_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5
echo "== bench"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_pw.json 2> gpurun_out/r3_bench_pw.err; grep "steps in" gpurun_out/r3_bench_pw.err
HAMSPINE_PW_STREAM=0 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_nopw.json 2> gpurun_out/r3_bench_nopw.err; grep "steps in" gpurun_out/r3_bench_nopw.err
