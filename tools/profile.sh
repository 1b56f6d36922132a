#!/bin/bash
# Runs on the GPU box: for one workload, the kernel-trace stats and the two HBM-traffic PMC passes
# (separate runs, as the counters do not fit one pass) of the benchmark command.
#   tools/profile.sh <tag> <name> [bench.py arguments...]
# Summaries land in gpurun_out/<tag>/<name>_*; tools/make_traffic.py turns the PMC summaries into
# profiles/traffic.json.  Copy what is to be judged into profiles/.
#
# Every pass is a child process of its own that this script waits for; the next pass starts only when the
# pass before it has exited AND no process of this user holds the GPU device node any more (a profiler that
# has returned may leave helpers that are still closing their counter session; a counter session opened on
# top of that one is what twice did not get past its start-up within the time limit).  A pass that is killed
# at its limit keeps its stderr as <name>_<pass>.KILLED.err next to the last heartbeat, and nothing more is
# started on the GPU in this call.
set -u
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

gpu_holders() {                 # processes of this user, other than this script's own, with /dev/kfd open
    local n=0 p
    for p in /proc/[0-9]*; do
        [ -O "$p" ] || continue
        [ "${p#/proc/}" = "$$" ] && continue
        if ls -l $p/fd 2>/dev/null | grep -q '/dev/kfd'; then n=$((n + 1)); fi
    done
    echo $n
}

wait_gpu_free() {               # up to 60 s; reports what it saw
    local i
    for i in $(seq 1 60); do
        [ "$(gpu_holders)" = 0 ] && return 0
        sleep 1
    done
    echo "profile.sh: $(gpu_holders) process(es) still hold /dev/kfd after 60 s" | tee -a $OUT/${NAME}_handover.log
    return 1
}

run_pass() {                    # run_pass <pass name> <limit seconds> <stdout file> command...
    local pass=$1 limit=$2 so=$3; shift 3
    wait_gpu_free || exit 3
    echo "== $NAME: $pass"
    timeout -s KILL $limit "$@" > $so 2> $OUT/${NAME}_${pass}.err &
    local pid=$!
    wait $pid
    local rc=$?
    if [ $rc -ne 0 ]; then
        cp $OUT/${NAME}_${pass}.err $OUT/${NAME}_${pass}.KILLED.err
        { echo "pass $pass ended with status $rc after at most $limit s; last heartbeat:"; tail -1 $OUT/heartbeat.log 2>/dev/null; } >> $OUT/${NAME}_${pass}.KILLED.err
        echo "profile.sh: pass $pass failed (status $rc); its stderr is kept in ${NAME}_${pass}.KILLED.err; stopping" >&2
        exit $rc
    fi
}

run_pass trace 300 $OUT/${NAME}_bench_under_trace.json \
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME}_trace -- python3 $R/bench.py "$@" --steps 100 --warmup 10 --regions 2 --no-cpu-baseline --no-sweep
python3 $R/tools/kstats.py "$OUT/${NAME}_trace/*/*kernel_stats.csv" > $OUT/${NAME}_kernel_stats.txt
python3 $R/tools/koverlap.py $(ls $OUT/${NAME}_trace/*/*kernel_trace.csv | head -1) >> $OUT/${NAME}_kernel_stats.txt
run_pass pmc_fetch 120 $OUT/${NAME}_pmc_fetch.json \
    rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $OUT/${NAME}_pmc_fetch -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep
run_pass pmc_write 120 $OUT/${NAME}_pmc_write.json \
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum --output-format csv -d $OUT/${NAME}_pmc_write -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --regions 0 --no-cpu-baseline --no-sweep
python3 $R/tools/pmc_summary.py "$OUT/${NAME}_pmc_fetch/*/*counter_collection.csv" "$OUT/${NAME}_pmc_write/*/*counter_collection.csv" > $OUT/${NAME}_pmc.txt
rm -rf $OUT/${NAME}_trace $OUT/${NAME}_pmc_fetch $OUT/${NAME}_pmc_write
rm -f $OUT/${NAME}_trace.err $OUT/${NAME}_pmc_fetch.err $OUT/${NAME}_pmc_write.err
cat $OUT/${NAME}_kernel_stats.txt $OUT/${NAME}_pmc.txt
