#!/bin/bash
# Runs on the GPU box: a window of the kernel trace of one bench line (tools/kwindow.py) into gpurun_out/r03/<name>_window.txt
set -e
NAME=$1; shift
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT/kw_$NAME
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/kw_$NAME -o $NAME -- python3 bench.py --no-cpu-baseline --no-sweep "$@" > $OUT/kw_$NAME.out 2> $OUT/kw_$NAME.err
F=$(find $OUT/kw_$NAME -name "*kernel_trace.csv" | head -1)
python3 tools/kwindow.py "$F" 0.6 3600 > $OUT/${NAME}_window.txt
rm -rf $OUT/kw_$NAME
