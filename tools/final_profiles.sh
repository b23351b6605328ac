#!/bin/bash
# Regenerates the committed round profiles on the GPU box (run from the repo root):
#   bash tools/final_profiles.sh roundN
# kernel traces of bench.py with and without stream overlap -> profiles/<round>_final_{isolated,overlapped}_*;
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate; towers un-overlapped like bench.py's roofline leg, so that the counted
# dispatches come in the order of the launch log that names their classes) -> profiles/<round>_pmc_traffic.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round3}
OUT=$R/gpurun_out/final
PROF=$OUT/profiles        # gpurun merges gpurun_out/ back; copy $PROF/* into profiles/ afterwards
mkdir -p $OUT $PROF
cd /tmp && export TMPDIR=/tmp
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/iso -o iso -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-ddp-config > $OUT/iso.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/ovl -o ovl -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --no-ddp-config > $OUT/ovl.log 2>&1
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-ddp-config --gemm-log $OUT/launches_pmc.csv > $OUT/fetch.log 2>&1
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-ddp-config > $OUT/write.log 2>&1
cd $R
python3 tools/profile_summary.py $OUT/iso/iso_results.db 0 $PROF/${TAG}_final_isolated \
  "rocprofv3 --kernel-trace of \`HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32\` (${TAG}, final)" \
  "MI355X, workload C2 (ResNet50 + BERT-base L=128 + FusionModule + MLP head, B=32, bf16). Both stream overlaps off: every kernel runs alone, so durations are per-kernel costs (the configuration bench.py's roofline leg uses). Steps in the trace: 3 warm-up + 10 timed + 3 single-step host probes + 3 of the roofline leg. bench.py's live figure for the same family (HIP events around each launch) reads ~3 us more per launch: the event bracket's own dispatch latency, which bench.py measures around an empty kernel and reports as event_bracket_us. The BERT attention core runs in attn_fwd_fused_kernel / attn_bwd_fused_kernel, so its FLOP are not in the GEMM family's count (strided data gradients are counted with the filter taps they execute)." > /dev/null
python3 tools/profile_summary.py $OUT/ovl/ovl_results.db 0 $PROF/${TAG}_final_overlapped \
  "rocprofv3 --kernel-trace of \`python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32\` (${TAG}, final, default configuration)" \
  "Default configuration: text tower on its own stream beside the image tower (the weight-gradient side stream is off by default). Kernels of the two towers share the GPU, so individual durations are inflated and their sum exceeds the wall clock. Steps in the trace: 16 overlapped + 3 un-overlapped steps of the roofline leg; under the tracer the two towers' kernels hardly overlap (tools/timeline.py on this trace: two streams busy together for < 1 ms of a step) and the run is host-bound, so the overlap the un-profiled run gets (12.2 vs 14.6 ms per step with HAMSPINE_TOWER_OVERLAP=0) is not visible here." > /dev/null
python3 tools/pmc_traffic.py $OUT/fetch/fetch_results.db $OUT/write/write_results.db $PROF/${TAG}_pmc_traffic.json $OUT/launches_pmc.csv > /dev/null
cp $PROF/${TAG}_pmc_traffic.json profiles/      # bench.py reads roofline.traffic from it
head -8 $PROF/${TAG}_final_isolated_summary.md
cat $PROF/${TAG}_pmc_traffic.json | head -12
python3 tools/timeline.py $OUT/ovl/ovl_results.db 5 9 > $PROF/${TAG}_timeline.txt 2>&1
for W in c2 c3 c4 c5; do
  python3 bench.py --workload $W --gemm-log $OUT/launches_$W.csv > $PROF/${TAG}_bench_$W.json 2> $OUT/bench_$W.err
  grep "steps in" $OUT/bench_$W.err
  python3 tools/launch_table.py $OUT/launches_$W.csv > $PROF/${TAG}_launch_table_$W.txt
done
HAMSPINE_TOWER_OVERLAP=0 python3 bench.py --no-f32 --no-cpu-baseline --no-ddp-config > $OUT/bench_serial.json 2> $OUT/bench_serial.err
grep "steps in" $OUT/bench_serial.err
