#!/usr/bin/env python3
"""Prints the yh:: kernel rows of a rocprofv3 kernel_stats.csv (names shortened)."""
import csv, sys, glob
for path in sys.argv[1:]:
    for f in glob.glob(path):
        print('#', f)
        for row in csv.DictReader(open(f)):
            n = row['Name']
            if 'yh::' in n:
                print('%-40s calls %5s avg_ns %10.1f min %8s max %8s pct %6s' % (
                    n.split('(')[0][-40:], row['Calls'], float(row['AverageNs']), row['MinNs'], row['MaxNs'], row['Percentage']))
