#!/bin/bash
# Regenerates the committed round profiles on the GPU box (run from the repo root):
#   bash tools/final_profiles.sh roundN
# kernel traces of bench.py with and without stream overlap -> profiles/<round>_final_{isolated,overlapped}_*;
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate) -> profiles/<round>_pmc_traffic.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round1}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/iso -o iso -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/iso.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/ovl -o ovl -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/ovl.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1
cd $R
python3 tools/profile_summary.py $OUT/iso/iso_results.db 16 profiles/${TAG}_final_isolated \
  "rocprofv3 --kernel-trace of \`HAMSPINE_OVERLAP=0 HAMSPINE_TOWER_OVERLAP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline\` (${TAG}, final)" \
  "MI355X, workload C2 (ResNet50 + BERT-base L=128 + FusionModule + MLP head, B=32, bf16). Both stream overlaps off: every kernel runs alone, so durations are per-kernel costs (the configuration bench.py's roofline leg uses). 16 steps in the trace (3 warm-up + 10 timed + 3 of the roofline leg). bench.py's live figure for the same family (HIP events around each launch) reads ~6 us more per launch: the event bracket's own dispatch latency, which bench.py measures around an empty kernel and reports as event_bracket_us. The BERT attention core runs in attn_fwd_fused_kernel / attn_bwd_fused_kernel, so its FLOP are not in the GEMM family's count (strided data gradients are counted with the filter taps they execute)." > /dev/null
python3 tools/profile_summary.py $OUT/ovl/ovl_results.db 16 profiles/${TAG}_final_overlapped \
  "rocprofv3 --kernel-trace of \`python bench.py --steps 10 --warmup 3 --no-cpu-baseline\` (${TAG}, final, default configuration)" \
  "Default configuration: text tower on its own stream beside the image tower (the weight-gradient side stream is off by default). Kernels of the two towers share the GPU, so individual durations are inflated and their sum exceeds the wall clock. 16 steps in the trace (13 overlapped + 3 un-overlapped steps of the roofline leg)." > /dev/null
python3 tools/pmc_traffic.py $OUT/fetch/fetch_results.db $OUT/write/write_results.db profiles/${TAG}_pmc_traffic.json > /dev/null
head -8 profiles/${TAG}_final_isolated_summary.md
cat profiles/${TAG}_pmc_traffic.json | head -12
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
cat $OUT/bench.json
