#!/bin/bash
# Runs on the GPU box: only the kernel-trace pass of tools/profile.sh.   tools/trace_only.sh <tag> <name> [bench.py arguments...]
set -e
TAG=$1; NAME=$2; shift 2
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME}_trace -- python3 $R/bench.py "$@" --steps 100 --warmup 10 --regions 2 --no-cpu-baseline --no-sweep > $OUT/${NAME}_bench_under_trace.json 2> $OUT/${NAME}_trace.err
python3 $R/tools/kstats.py "$OUT/${NAME}_trace/*/*kernel_stats.csv" > $OUT/${NAME}_kernel_stats.txt
python3 $R/tools/koverlap.py $(ls $OUT/${NAME}_trace/*/*kernel_trace.csv | head -1) >> $OUT/${NAME}_kernel_stats.txt
rm -rf $OUT/${NAME}_trace
cat $OUT/${NAME}_kernel_stats.txt
