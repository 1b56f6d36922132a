#!/bin/bash
# Runs on the GPU box: HBM-traffic counters of the rows kernel at one point of the survival sweep.
#   tools/pmc_sweep.sh <tag> <name> <dense_min> <mu> [hot_mean]
# One rocprofv3 --pmc run (FETCH_SIZE TCC_MISS_sum) of tools/time_sweep.py; per-kernel means in gpurun_out/<tag>/<name>_pmc.txt
set -e
TAG=$1; NAME=$2; DM=$3; MU=$4; HOT=${5:--6.0}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $OUT/${NAME}_pmc_fetch -- python3 $R/tools/time_sweep.py detect416 $DM $HOT $MU > $OUT/${NAME}_pmc_fetch.out 2> $OUT/${NAME}_pmc_fetch.err
python3 $R/tools/pmc_summary.py "$OUT/${NAME}_pmc_fetch/*/*counter_collection.csv" > $OUT/${NAME}_pmc.txt
rm -rf $OUT/${NAME}_pmc_fetch
cat $OUT/${NAME}_pmc.txt
