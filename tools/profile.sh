#!/bin/bash
# Runs on the GPU box: for one workload, the kernel-trace stats and the two HBM-traffic PMC passes
# (separate runs, as the counters do not fit one pass) of the benchmark command.
#   tools/profile.sh <tag> <name> [bench.py arguments...]
# Summaries land in gpurun_out/<tag>/<name>_*; tools/make_traffic.py turns the PMC summaries into
# profiles/traffic.json.  Copy what is to be judged into profiles/.
set -e
TAG=$1; NAME=$2; shift 2
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
# a sign of life every 30 s: a long profiler pass writes its files only when it ends
( while sleep 30; do date >> $OUT/heartbeat.log; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
echo "== $NAME: kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME}_trace -- python3 $R/bench.py "$@" --steps 100 --warmup 10 --regions 2 --no-cpu-baseline --no-sweep > $OUT/${NAME}_bench_under_trace.json 2> $OUT/${NAME}_trace.err
python3 $R/tools/kstats.py "$OUT/${NAME}_trace/*/*kernel_stats.csv" > $OUT/${NAME}_kernel_stats.txt
python3 $R/tools/koverlap.py $(ls $OUT/${NAME}_trace/*/*kernel_trace.csv | head -1) >> $OUT/${NAME}_kernel_stats.txt
echo "== $NAME: PMC fetch"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $OUT/${NAME}_pmc_fetch -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep > $OUT/${NAME}_pmc_fetch.json 2> $OUT/${NAME}_pmc_fetch.err
echo "== $NAME: PMC write"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum --output-format csv -d $OUT/${NAME}_pmc_write -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep > $OUT/${NAME}_pmc_write.json 2> $OUT/${NAME}_pmc_write.err
python3 $R/tools/pmc_summary.py "$OUT/${NAME}_pmc_fetch/*/*counter_collection.csv" "$OUT/${NAME}_pmc_write/*/*counter_collection.csv" > $OUT/${NAME}_pmc.txt
rm -rf $OUT/${NAME}_trace $OUT/${NAME}_pmc_fetch $OUT/${NAME}_pmc_write
cat $OUT/${NAME}_kernel_stats.txt $OUT/${NAME}_pmc.txt
