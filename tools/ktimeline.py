#!/usr/bin/env python3
"""Start, end and queue of every dispatch of one kernel in a rocprofv3 kernel trace, in start order,
split into bursts (a gap of more than `gap_us` before a dispatch starts a new burst): what the short
timed regions of bench.py look like on the device.
usage: ktimeline.py <kernel_trace.csv> [kernel substring] [gap_us]"""
import csv, sys
path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else 'k_detect_rows'
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
rows = []
for r in csv.DictReader(open(path)):
    if name in r['Kernel_Name']:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?')))
rows.sort()
bursts, cur = [], []
for s, e, q in rows:
    if cur and (s - max(x[1] for x in cur)) > gap_us * 1e3:
        bursts.append(cur); cur = []
    cur.append((s, e, q))
if cur:
    bursts.append(cur)
for b in bursts:
    t0 = b[0][0]
    span = (max(x[1] for x in b) - t0) / 1e3
    print('burst of %4d dispatches, %9.1f us from first start to last end, %6.2f us per dispatch' % (len(b), span, span / len(b)))
    if len(b) <= 40:
        for s, e, q in b:
            print('    start %8.1f  end %8.1f  dur %6.1f  queue %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q))
