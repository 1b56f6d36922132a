#!/usr/bin/env python3
"""A window of a rocprofv3 kernel trace: every dispatch of the library's kernels that starts inside it, in start
order, with its queue: how the launches of several streams share the device.
usage: kwindow.py <kernel_trace.csv> [skip fraction 0..1] [window us]"""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'))
        for r in csv.DictReader(open(sys.argv[1])) if 'yh::' in r['Kernel_Name'] and 'probe' not in r['Kernel_Name']]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
win = float(sys.argv[3]) if len(sys.argv) > 3 else 4000.0
t0 = rows[int(len(rows) * skip)][0]
for s, e, n, q in rows:
    if t0 <= s < t0 + win * 1e3:
        short = n.split('yh::')[1].split('(')[0]
        print('%9.1f -> %9.1f  %7.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))
