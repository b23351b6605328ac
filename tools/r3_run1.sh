#!/bin/bash
# round-3 GPU call 1: new GEMM body parity + A/B, full GPU suite, bench with the ddp-config leg
set -o pipefail
mkdir -p gpurun_out
echo "== p8 tests" 
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8" 2>&1 | tail -15 | tee gpurun_out/r3_p8_tests.log
echo "== p8 bench warm"
timeout -k 10 300 python tools/p8_bench.py > gpurun_out/r3_p8_bench_warm.txt 2>&1; tail -12 gpurun_out/r3_p8_bench_warm.txt
echo "== p8 bench cold"
timeout -k 10 300 python tools/p8_bench.py --cold > gpurun_out/r3_p8_bench_cold.txt 2>&1; tail -12 gpurun_out/r3_p8_bench_cold.txt
echo "== full gpu suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r3_gpu_tests1.log
echo "== bench"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; tail -5 gpurun_out/r3_bench1.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3_bench1.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "final_loss")}, d["roofline"]["frac"], d.get("ddp_config"))
except Exception as e:
    print("bench parse failed", e)
PY
