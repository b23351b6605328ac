#!/bin/bash
# Marginal cost of each class of launches in the OVERLAPPED C2 step: the step time with the class left out (HAMSPINE_KNOCKOUT,
# csrc/blocks.hip knock(); results are garbage in those runs) against the full step.   bash tools/knockout.sh > table.txt
run() { HAMSPINE_KNOCKOUT=$1 python3 bench.py --no-f32 --no-cpu-baseline --steps 20 --warmup 5 2>&1 >/dev/null | grep "steps in" | sed -E 's/.*-> ([0-9.]+) ms\/step.*/\1/'; }
base=$(run 0)
echo "full step: $base ms"
for kv in "1:ResNet weight gradients" "2:BERT weight-gradient GEMMs" "4:transposes" "6:BERT weight gradients + transposes" "8:BatchNorm forward" "16:BatchNorm backward" "24:BatchNorm forward + backward" "32:optimizer" "64:weight casts" "128:LayerNorm backward" "256:attention cores" "512:ResNet data-gradient convolutions" "1024:BERT data-gradient GEMMs"; do
  k=${kv%%:*}; name=${kv#*:}
  t=$(run $k)
  echo "without $name (mask $k): $t ms  (marginal $(python3 -c "print(round($base-$t,2))") ms)"
done
