#!/bin/bash
# Runs on the GPU box: the long consistency runs (tools/longrun.py, tools/longrun_loss.py), logs in gpurun_out/r03/.
# Stops at the first run that fails or is killed at its limit.
mkdir -p gpurun_out/r03
run() {                     # name, seconds, command...
    local name=$1 limit=$2; shift 2
    timeout -k 10 $limit "$@" > gpurun_out/r03/longrun_$name.txt 2>&1
    local rc=$?
    tail -2 gpurun_out/r03/longrun_$name.txt
    return $rc
}
run detect416 420 python -u tools/longrun.py detect416 ${1:-1000000} 3 &&
run detect608 200 python -u tools/longrun.py detect608 150000 3 &&
run detect_v2 200 python -u tools/longrun.py detect_v2 300000 2 &&
run stress 200 python -u tools/longrun.py stress 20000 3 &&
run train416 300 python -u tools/longrun_loss.py 60000
