#!/bin/bash
# Runs on the GPU box: the property tests of tests/test_gpu_properties.py as a soak, N fresh examples each
# (default 1500) from hypothesis seed SEED (default 0: the same draws again), log in gpurun_out/r03/.
# usage: tools/soak.sh [examples] [seconds] [seed] [pytest -k expression]
mkdir -p gpurun_out/r03
N=${1:-1500}
LIMIT=${2:-900}
SEED=${3:-0}
LOG=gpurun_out/r03/soak_${N}_seed${SEED}.log
( while sleep 60; do echo "soak: $(date +%T) still running"; done ) &      # a run that prints nothing for minutes is taken to be hung
HEARTBEAT=$!
YH_PROPERTY_EXAMPLES=$N timeout -k 10 $LIMIT python -u -m pytest tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider \
    --timeout $LIMIT --timeout-method thread --hypothesis-show-statistics --hypothesis-seed=$SEED ${4:+-k "$4"} > $LOG 2>&1
rc=$?
kill $HEARTBEAT
grep -v "Warning\|TypedDict\|^$\|class .*(TypedDict" $LOG | tail -60
exit $rc
