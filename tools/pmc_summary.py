#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 --pmc counter_collection.csv (yh:: kernels only)."""
import csv, sys, glob, collections
for pat in sys.argv[1:]:
    for f in glob.glob(pat):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'yh::' not in n:
                continue
            acc[n.split('(')[0][-28:]][r['Counter_Name']].append(float(r['Counter_Value']))
        print('#', f)
        for k, cs in acc.items():
            print(k, ' '.join('%s=%.4g(n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
