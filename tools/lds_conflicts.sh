#!/bin/bash
# LDS bank conflicts per kernel over a few C2 steps: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (its own pass),
# summed per kernel name.  usage: bash tools/lds_conflicts.sh <outdir under the repo>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/ldsc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]]
        if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT":
            a[0] += float(r["Counter_Value"]); a[2] += 1
        elif r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
            a[1] += float(r["Counter_Value"])
print(f"{'conflict':>12} {'active':>12} {'frac':>6} {'launches':>8}  kernel")
for k, (c, a, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    if a > 0:
        print(f"{c:12.3g} {a:12.3g} {c / a:6.2f} {n:8d}  {k[:110]}")
PY
