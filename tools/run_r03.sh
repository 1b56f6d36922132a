#!/bin/bash
# Runs on the GPU box: bench lines into gpurun_out/r03/ (steps chained: stops at the first failure).
#   tools/run_r03.sh <name>:<bench.py arguments> ...
set -e
mkdir -p gpurun_out/r03
for spec in "$@"; do
    name=${spec%%:*}; args=${spec#*:}
    timeout -k 10 400 python bench.py --no-cpu-baseline $args > gpurun_out/r03/bench_$name.json 2> gpurun_out/r03/bench_$name.err
    python tools/show_bench.py gpurun_out/r03/bench_$name.json
done
