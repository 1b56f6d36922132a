#!/bin/bash
# Runs on the GPU box: kernel-trace stats and the HBM-traffic PMC pass of the default
# benchmark command; summaries land in gpurun_out/<tag>/ (copy what is to be judged into profiles/).
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
python3 $R/tools/kstats.py "$OUT/trace/*/*kernel_stats.csv" > $OUT/kernel_stats_summary.txt
python3 $R/tools/koverlap.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) k_detect_rows >> $OUT/kernel_stats_summary.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $R/tools/pmc_summary.py "$OUT/pmc_fetch/*/*counter_collection.csv" "$OUT/pmc_write/*/*counter_collection.csv" > $OUT/pmc_summary.txt
cat $OUT/kernel_stats_summary.txt $OUT/pmc_summary.txt
