#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_product_gpu.py tests/test_train_loop_gpu.py -x -q -k "moe or train_loop or attention" 2>&1 | tail -15
echo "== full suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r3_gpu_tests4.log
echo "== bench c5"
timeout -k 10 600 python bench.py --workload c5 --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_c5b.json 2> gpurun_out/r3_bench_c5b.err; tail -2 gpurun_out/r3_bench_c5b.err
echo "== bench c2"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_c2b.json 2> gpurun_out/r3_bench_c2b.err; tail -2 gpurun_out/r3_bench_c2b.err
HAMSPINE_P8=0 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_c2b_nop8.json 2> gpurun_out/r3_bench_c2b_nop8.err; tail -2 gpurun_out/r3_bench_c2b_nop8.err
