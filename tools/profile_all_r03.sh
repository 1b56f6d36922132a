#!/bin/bash
# Runs HERE (not on the GPU box): one gpurun call per workload, each running tools/profile.sh for it; the summaries
# come back under gpurun_out/r03/.  usage: tools/profile_all_r03.sh [workload ...]
set -u
declare -A ARGS=(
  [detect416]="--workload detect416"
  [detect608]="--workload detect608"
  [detect_v2]="--workload detect_v2"
  [stress]="--workload stress --streams 1 --in-flight 1"
  [nms_seam]="--workload nms_seam --streams 1"
  [loss416]="--workload loss416"
  [loss416_prepared]="--workload loss416 --loss-prepared"
  [train416]="--workload train416"
)
for name in "${@:-detect416 detect608 detect_v2 stress nms_seam loss416 loss416_prepared train416}"; do
  for n in $name; do
    echo "=== $n"
    /usr/local/graft/bin/gpurun --timeout 900 -- "tools/profile.sh r03 $n ${ARGS[$n]}" 2>&1 | tail -25
  done
done
