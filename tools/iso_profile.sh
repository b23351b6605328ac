#!/bin/bash
# quick look: kernel trace of a few un-overlapped steps -> gpurun_out/iso_quick_summary.md   (bash tools/iso_profile.sh [workload])
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-c2}
OUT=$R/gpurun_out/isoq
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/iso -o iso -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-f32 > $OUT/iso.log 2>&1
cd $R
python3 tools/profile_summary.py $OUT/iso/iso_results.db 12 gpurun_out/iso_quick_$W "quick isolated trace ($W)" > /dev/null
head -70 gpurun_out/iso_quick_${W}_summary.md
