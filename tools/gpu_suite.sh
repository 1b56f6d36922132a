#!/bin/bash
# Runs on the GPU box: the whole GPU test suite (one process, per-test time limit), its log in gpurun_out/r03/.
# Stops the call at the first failure, so that nothing else touches the GPU after a fault.
set -e
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/pytest_${1:-gpu}.log
if timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $LOG 2>&1; then
    tail -3 $LOG
else
    rc=$?
    tail -60 $LOG
    exit $rc
fi
