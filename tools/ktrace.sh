#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of one bench line into gpurun_out/r03/<name>_stats.txt
#   tools/ktrace.sh <name> <bench.py arguments>
set -e
NAME=$1; shift
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT/kt_$NAME
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$NAME -o $NAME -- python3 bench.py --no-cpu-baseline --no-sweep "$@" > $OUT/kt_$NAME.out 2> $OUT/kt_$NAME.err
F=$(find $OUT/kt_$NAME -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY' | tee $OUT/${NAME}_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print('%-90s calls %7s avg %10.1f ns  total %6.2f%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
rm -rf $OUT/kt_$NAME
