#!/usr/bin/env python3
"""Duration percentiles per kernel from a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
import numpy as np
for path in glob.glob(sys.argv[1]):
    rows = list(csv.DictReader(open(path)))
    by = {}
    for r in rows:
        by.setdefault(r['Kernel_Name'][:60], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        v = np.array(v)
        if len(v) < 5:
            continue
        print('%-60s n %4d  p10 %.2f p50 %.2f p90 %.2f mean %.2f us' % (
            k, len(v), *(np.percentile(v, [10, 50, 90]) / 1e3), v.mean() / 1e3))
