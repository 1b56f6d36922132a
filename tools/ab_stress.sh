#!/bin/bash
# Runs on the GPU box: the stress line (kernel alone / per step) with several builds of the library.
#   tools/ab_stress.sh <lib name> ...     (yolo-tf_amd/lib/lib_<name>.so; "cur" = libyolohead.so)
set -e
mkdir -p gpurun_out/r03
for n in "$@"; do
    lib=yolo-tf_amd/lib/lib_$n.so; [ "$n" = cur ] && lib=yolo-tf_amd/lib/libyolohead.so
    timeout -k 10 200 python tools/bench_with_lib.py $lib --workload stress --steps 200 --regions 2 --no-cpu-baseline > gpurun_out/r03/ab_$n.json 2> gpurun_out/r03/ab_$n.err
    python - "$n" gpurun_out/r03/ab_$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print('%-8s step %8.1f us  kernel alone %7.1f us  serial %s  regions %s' % (sys.argv[1], 1e3 * d['ms_per_step'], 1e3 * d['roofline']['kernel_ms'],
      d['config'].get('serial_ms_per_step'), d['timing']['regions_ms_per_step']))
PY
done
