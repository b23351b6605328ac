#!/bin/bash
# A/B of two configurations by rocprofv3 kernel traces of un-overlapped steps: per-kernel ms/step for the kernels that differ
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/isoab
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "pw:1" "nopw:0"; do
  tag=${cfg%%:*}; v=${cfg#*:}
  HAMSPINE_PW_STREAM=$v HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/$tag -o t -- python3 $R/tools/step_time.py --steps 10 --warmup 3 > $OUT/$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import sqlite3, collections, os, glob
R=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
res={}
for tag in ("pw","nopw"):
    db=glob.glob(f"{R}/gpurun_out/isoab/{tag}/**/*_results.db", recursive=True)[0]
    rows=sqlite3.connect(db).execute("select name, end - start from kernels").fetchall()
    by=collections.defaultdict(lambda:[0,0])
    for n,d in rows:
        by[n][0]+=1; by[n][1]+=d
    res[tag]=by
steps=13
names=sorted(set(res["pw"])|set(res["nopw"]), key=lambda n:-(res["pw"].get(n,[0,0])[1]+res["nopw"].get(n,[0,0])[1]))
tot={t:sum(v[1] for v in res[t].values())/steps/1e6 for t in res}
print("total kernel ms/step:", tot)
print(f"{'kernel':90s} {'pw calls':>8s} {'pw ms':>7s} {'nopw calls':>10s} {'nopw ms':>8s}")
for n in names[:45]:
    a=res["pw"].get(n,[0,0]); b=res["nopw"].get(n,[0,0])
    if abs(a[1]-b[1])/steps/1e6 < 0.01: continue
    print(f"{n[:90]:90s} {a[0]/steps:8.1f} {a[1]/steps/1e6:7.3f} {b[0]/steps:10.1f} {b[1]/steps/1e6:8.3f}")
PY
