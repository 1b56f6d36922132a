"""Splits the dispatches of one kernel in a rocprofv3 kernel trace into those that started while an
earlier dispatch of the same kernel was still running (launches on several streams overlap even under
the tracer once the host enqueues fast enough) and those that ran alone.
usage: koverlap.py <kernel_trace.csv> <kernel name substring>"""
import csv
import statistics as st
import sys


def main(path, name):
    rows = [r for r in csv.DictReader(open(path)) if name in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    over, alone, prev_end = [], [], 0
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        (over if s < prev_end else alone).append((e - s) / 1e3)
        prev_end = max(prev_end, e)
    for what, d in (('alone', alone), ('overlapping an earlier dispatch', over)):
        if d:
            print('%s %-32s n %4d  mean %.2f  median %.2f us' % (name, what, len(d), st.mean(d), st.median(d)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
