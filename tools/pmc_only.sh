#!/bin/bash
# Runs on the GPU box: only the two HBM-traffic PMC passes of tools/profile.sh (separate runs), with a
# short time limit each.   tools/pmc_only.sh <tag> <name> [bench.py arguments...]
set -e
TAG=$1; NAME=$2; shift 2
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
( while sleep 30; do date >> $OUT/heartbeat.log; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
timeout -s KILL ${PMC_LIMIT:-120} rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $OUT/${NAME}_pmc_fetch -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep > $OUT/${NAME}_pmc_fetch.json 2> $OUT/${NAME}_pmc_fetch.err
timeout -s KILL ${PMC_LIMIT:-120} rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum --output-format csv -d $OUT/${NAME}_pmc_write -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep > $OUT/${NAME}_pmc_write.json 2> $OUT/${NAME}_pmc_write.err
python3 $R/tools/pmc_summary.py "$OUT/${NAME}_pmc_fetch/*/*counter_collection.csv" "$OUT/${NAME}_pmc_write/*/*counter_collection.csv" > $OUT/${NAME}_pmc.txt
rm -rf $OUT/${NAME}_pmc_fetch $OUT/${NAME}_pmc_write
cat $OUT/${NAME}_pmc.txt
