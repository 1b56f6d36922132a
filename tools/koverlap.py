"""Splits the dispatches of the library's main kernels in a rocprofv3 kernel trace into those that
started while an earlier dispatch of the same kernel was still running (launches on several streams
overlap even under the tracer once the host enqueues fast enough) and those that ran alone; mean,
median and p10/p90 of the durations.
usage: koverlap.py <kernel_trace.csv> [kernel name substring ...]"""
import csv
import statistics as st
import sys

DEFAULT = ('k_detect_rows', 'k_stream_wave', 'k_stream<', 'k_loss_main', 'k_sort_nms', 'k_post', 'k_loss_finalize')


def pct(d, q):
    d = sorted(d)
    return d[min(len(d) - 1, int(q * len(d)))]


def main(path, names):
    allrows = list(csv.DictReader(open(path)))
    for name in names:
        rows = [r for r in allrows if name in r['Kernel_Name']]
        if not rows:
            continue
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        over, alone, prev_end = [], [], 0
        for r in rows:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            (over if s < prev_end else alone).append((e - s) / 1e3)
            prev_end = max(prev_end, e)
        for what, d in (('alone', alone), ('overlapping an earlier dispatch', over)):
            if d:
                print('%-16s %-32s n %5d  mean %8.2f  median %8.2f  p10 %8.2f  p90 %8.2f us' % (
                    name, what, len(d), st.mean(d), st.median(d), pct(d, 0.1), pct(d, 0.9)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:] or DEFAULT)
