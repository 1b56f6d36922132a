#!/bin/bash
# Sweeps the stream kernel's workgroup count on the default benchmark workload.
for nb in 256 512 1024 1800 3600; do
  YH_STREAM_BLOCKS=$nb timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('blocks=$nb', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" || exit 1
done
