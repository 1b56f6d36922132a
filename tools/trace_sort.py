"""Phases of k_sort_nms per segment (measurement build: make -C yolo-tf_amd/csrc trace).
Runs the config-5 shape a few times, reads the stamps of the last call and prints how long the
segments' workgroups spend in each phase (microseconds; the clock ticks at 100 MHz)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), 'libyolohead_trace.so')

dev = torch.device('cuda:0')
w = bench.WORKLOADS['stress']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = int(os.environ.get('B', w['batch'])), len(priors[0]), w['C']
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
lib = _abi.load()
lib.yh_trace_sort_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
sets = [y.synth.torch_heads(dev, 40 + i, B, grids, P, C, **w['dist']) for i in range(2)]
hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()
for i in range(4):
    if i == 3:
        torch.cuda.synchronize()
        assert lib.yh_trace_sort_read(None, 0, 1) == 0
    hd.detect(sets[i % 2], w['thr'], w['iou'], w['max_boxes'])
torch.cuda.synchronize()
tr = np.zeros((1 << 14, 8), np.uint64)
assert lib.yh_trace_sort_read(tr.ctypes.data, tr.nbytes, 0) == 0
tr = tr[tr[:, 7] > 0]
t = tr[:, :6].astype(np.float64) / 100.0
t0 = t[:, 0].min()
cnt, n, kept = (tr[:, 6] >> np.uint64(40)).astype(np.int64), ((tr[:, 6] >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.int64), (tr[:, 6] & np.uint64(0xfffff)).astype(np.int64)
def q(x):
    return 'min %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f  mean %7.2f' % (x.min(), *np.percentile(x, [10, 50, 90]), x.max(), x.mean())
print('%d segments that took the prefix path (of the first %d)' % (len(tr), 1 << 14))
print('  keys per segment   %s' % q(cnt.astype(np.float64)))
print('  selected           %s' % q(n.astype(np.float64)))
print('  kept               %s' % q(kept.astype(np.float64)))
print('  start at           %s' % q(t[:, 0] - t0))
print('  end at             %s' % q(t[:, 5] - t0))
for name, a, b in (('keys to LDS', 0, 1), ('select_prefix', 1, 2), ('sort_selected', 2, 3), ('boxes to LDS', 3, 4), ('walk', 4, 5), ('whole', 0, 5)):
    print('  %-18s %s' % (name, q(t[:, b] - t[:, a])))
hw = ((tr[:, 7] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
xcd = ((tr[:, 7] >> np.uint64(40)) & np.uint64(15)).astype(np.int64)
cu = (xcd << 8) | ((hw >> 8) & 0xff)
ids, c = np.unique(cu, return_counts=True)
print('  %d distinct CUs, segments per CU %s' % (len(ids), q(c.astype(np.float64))))
# how many workgroups of this kernel were resident at once on a CU (overlap of [start, end) intervals)
res = []
for u in ids[:64]:
    m = cu == u
    ev = sorted([(x, 1) for x in t[m, 0]] + [(x, -1) for x in t[m, 5]])
    cur = best = 0
    for _, d in ev:
        cur += d
        best = max(best, cur)
    res.append(best)
print('  most workgroups resident at once on a CU (64 CUs looked at): %s' % q(np.asarray(res, np.float64)))
