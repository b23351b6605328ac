#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_product_gpu.py tests/test_train_loop_gpu.py tests/test_gemm_gpu.py -x -q -k "moe or train_loop or p8 or adamw" 2>&1 | tail -15
echo "== bench c5"
timeout -k 10 600 python bench.py --workload c5 --steps 20 --warmup 5 --no-ddp-config > gpurun_out/r3_bench_c5.json 2> gpurun_out/r3_bench_c5.err; tail -4 gpurun_out/r3_bench_c5.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_c5.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "final_loss")}, d["roofline"]["frac"], d.get("f32_mode", {}).get("ms_per_step"), d.get("cpu_baseline"))
PY
