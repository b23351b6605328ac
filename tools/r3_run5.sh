#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r3_gpu_tests5.log
timeout -k 10 600 python bench.py --workload c5 --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_c5c.json 2> gpurun_out/r3_bench_c5c.err; grep "steps in" gpurun_out/r3_bench_c5c.err
