#!/bin/bash
# Runs on the GPU box: two SQ counter passes (eight counters each, the block's limit) over a workload
# of bench.py; per-kernel means land in gpurun_out/<tag>/<name>_sq.txt.
#   tools/pmc_sq.sh <tag> <name> [bench.py arguments...]
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
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/${NAME}_sq1 -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep > $OUT/${NAME}_sq1.json 2> $OUT/${NAME}_sq1.err
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/${NAME}_sq2 -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep > $OUT/${NAME}_sq2.json 2> $OUT/${NAME}_sq2.err
python3 $R/tools/pmc_summary.py "$OUT/${NAME}_sq1/*/*counter_collection.csv" "$OUT/${NAME}_sq2/*/*counter_collection.csv" > $OUT/${NAME}_sq.txt
rm -rf $OUT/${NAME}_sq1 $OUT/${NAME}_sq2
cat $OUT/${NAME}_sq.txt
