#!/bin/bash
# Runs on the GPU box: the property tests of tests/test_gpu_properties.py as a soak, N fresh examples each
# (default 1500), log in gpurun_out/r03/.   usage: tools/soak.sh [examples] [seconds]
mkdir -p gpurun_out/r03
N=${1:-1500}
LIMIT=${2:-900}
LOG=gpurun_out/r03/soak_${N}.log
YH_PROPERTY_EXAMPLES=$N timeout -k 10 $LIMIT python -u -m pytest tests/test_gpu_properties.py -m gpu -x -q -p no:cacheprovider \
    --timeout $LIMIT --timeout-method thread --hypothesis-show-statistics > $LOG 2>&1
rc=$?
tail -40 $LOG
exit $rc
