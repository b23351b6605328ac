R=$(pwd); OUT=$R/gpurun_out/tl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/ovl -o ovl -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 > $OUT/ovl.log 2>&1
cd $R
grep "steps in" $OUT/ovl.log | cut -c1-200
python3 tools/timeline.py $OUT/ovl/ovl_results.db 5 9 > gpurun_out/timeline.txt 2>&1
