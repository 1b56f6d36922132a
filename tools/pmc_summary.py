#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 --pmc counter_collection.csv files (yh:: kernels only).
One line per kernel: name counter=mean(n=dispatches) ..."""
import csv, sys, glob, collections
for pat in sys.argv[1:]:
    for f in sorted(glob.glob(pat)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'yh::' not in n:
                continue
            short = n[n.index('yh::'):].split('(')[0]
            acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
        print('#', '/'.join(f.split('/')[-3:]))
        for k, cs in acc.items():
            print(k.replace(' ', ''), ' '.join('%s=%.6g(n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
