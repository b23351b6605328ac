#!/bin/bash
# Marginal cost of each class of launches in the OVERLAPPED C2 step: the step time with the class left out (HAMSPINE_KNOCKOUT,
# csrc/blocks.hip knock(); results are garbage in those runs) against the full step.   bash tools/knockout.sh > table.txt
# The switch is compiled in only with -DHS_MEASURE, and bench.py refuses such a build / the variable, so this script rebuilds
# the library that way, times the step with tools/step_time.py (prints ms/step only, never a bench line) and restores the
# release build at the end.
CSRC=multimodal-diagnosis-ham-spine_amd/hamspine/csrc
touch $CSRC/blocks.hip && make -C $CSRC -j8 EXTRA=-DHS_MEASURE >/dev/null || exit 1
trap 'touch $CSRC/blocks.hip && make -C $CSRC -j8 >/dev/null' EXIT
run() { HAMSPINE_KNOCKOUT=$1 python3 tools/step_time.py --steps 20 --warmup 5; }
base=$(run 0)
echo "full step: $base ms"
for kv in "1:ResNet weight gradients" "2:BERT weight-gradient GEMMs" "4:transposes" "6:BERT weight gradients + transposes" "8:BatchNorm forward" "16:BatchNorm backward" "24:BatchNorm forward + backward" "32:optimizer" "64:weight casts" "128:LayerNorm backward" "256:attention cores" "512:ResNet data-gradient convolutions" "1024:BERT data-gradient GEMMs"; do
  k=${kv%%:*}; name=${kv#*:}
  t=$(run $k)
  echo "without $name (mask $k): $t ms  (marginal $(python3 -c "print(round($base-$t,2))") ms)"
done
