#!/bin/bash
# A/B of configurations (environment switches) on the GPU box: un-profiled step time (3 runs each, interleaved), then
# rocprofv3 kernel traces of un-overlapped steps with per-kernel ms/step for the kernels that differ from the first config.
#   bash tools/iso_ab.sh base: xb:HAMSPINE_XBLOCK_BN=1 ...        (tag:VAR=value[,VAR=value])
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/isoab
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for cfg in "$@"; do
    tag=${cfg%%:*}; v=${cfg#*:}
    ( for kv in ${v//,/ }; do export $kv; done
      echo "$tag $(timeout -k 10 300 python3 $R/tools/step_time.py --steps 30 --warmup 8)" ) >> $OUT/step_ms.txt || exit 1
  done
done
cat $OUT/step_ms.txt
for cfg in "$@"; do
  tag=${cfg%%:*}; v=${cfg#*:}
  ( for kv in ${v//,/ }; do export $kv; done
    export HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0
    timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/$tag -o t -- python3 $R/tools/step_time.py --steps 10 --warmup 3 > $OUT/$tag.log 2>&1 ) || exit 1
done
cd $R
python3 - "$@" <<'PY'
import sqlite3, collections, os, glob, sys
R=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
tags=[c.split(":")[0] for c in sys.argv[1:]]
res={}
for tag in tags:
    db=glob.glob(f"{R}/gpurun_out/isoab/{tag}/**/*_results.db", recursive=True)[0]
    rows=sqlite3.connect(db).execute("select name, end - start from kernels").fetchall()
    by=collections.defaultdict(lambda:[0,0])
    for n,d in rows:
        by[n][0]+=1; by[n][1]+=d
    res[tag]=by
steps=13
print("total kernel ms/step:", {t:round(sum(v[1] for v in res[t].values())/steps/1e6,3) for t in tags})
names=sorted(set().union(*[set(res[t]) for t in tags]), key=lambda n:-max(res[t].get(n,[0,0])[1] for t in tags))
print(f"{'kernel':84s} " + " ".join(f"{t+' n':>8s} {t+' ms':>8s}" for t in tags))
for n in names[:60]:
    b=res[tags[0]].get(n,[0,0])
    if all(abs(res[t].get(n,[0,0])[1]-b[1])/steps/1e6 < 0.008 for t in tags[1:]): continue
    print(f"{n[:84]:84s} " + " ".join(f"{res[t].get(n,[0,0])[0]/steps:8.1f} {res[t].get(n,[0,0])[1]/steps/1e6:8.3f}" for t in tags))
PY
