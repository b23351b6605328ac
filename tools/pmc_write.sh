#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of one GEMM shape: pmc_write.sh kind M N K cfg bits
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pw_$c
  timeout -k 10 120 rocprofv3 --pmc $c -d /tmp/pw_$c --output-format csv -- python3 $R/tools/gemm_one.py "$@" > /tmp/pw_$c.log 2>&1
  python3 - $c <<'PY'
import csv, glob, sys
v=[float(r["Counter_Value"]) for f in glob.glob(f"/tmp/pw_{sys.argv[1]}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm_bf16_kernel" in r["Kernel_Name"]]
print(sys.argv[1], "n", len(v), "mean MB", sum(v)/len(v)*1024/1e6 if v else None)
PY
done
