#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "batchnorm" 2>&1 | tail -6
timeout -k 10 900 python -m pytest tests/test_product_gpu.py tests/test_tower_gpu.py tests/test_fullsize_gpu.py tests/test_edge_cases_gpu.py tests/test_decisions_gpu.py -x -q 2>&1 | tail -6
echo "== bench"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-ddp-config --no-f32 --no-cpu-baseline > gpurun_out/r3_bench_xs.json 2> gpurun_out/r3_bench_xs.err; grep "steps in" gpurun_out/r3_bench_xs.err
echo "== iso trace"
R=$(pwd); OUT=$R/gpurun_out/iso_xs; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT -o t -- python3 $R/tools/step_time.py --steps 10 --warmup 3 > $OUT/log.txt 2>&1
cd $R
python3 - <<'PY'
import sqlite3, collections, glob, os
db=glob.glob("gpurun_out/iso_xs/**/*_results.db", recursive=True)[0]
rows=sqlite3.connect(db).execute("select name, end - start from kernels").fetchall()
by=collections.defaultdict(lambda:[0,0])
for n,d in rows: by[n][0]+=1; by[n][1]+=d
steps=13
print("total ms/step", sum(v[1] for v in by.values())/steps/1e6)
for n,(c,t) in sorted(by.items(), key=lambda kv:-kv[1][1])[:24]:
    print(f"{t/steps/1e6:7.3f} {c/steps:6.1f} {t/c/1e3:7.1f}us  {n[:100]}")
PY
