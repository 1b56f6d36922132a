"""Per-unit timeline of units that are read whole (measurement build: make -C yolo-tf_amd/csrc trace):
trace_whole.py [mu] [hot_image]   -- objectness ~ N(mu, 2); hot_image=1: only image 0 survives everywhere.
Prints, over the units read whole in the last launch: duration, time waiting for a chunk's loads, time scanning."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), 'libyolohead_trace.so')

dev = torch.device('cuda:0')
mu = float(sys.argv[1]) if len(sys.argv) > 1 else 9.0
hot = len(sys.argv) > 2 and sys.argv[2] == '1'
w = bench.WORKLOADS['detect416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
sets = [y.synth.torch_heads(dev, i, B, grids, P, C, obj_mean=(-4.0 if hot else mu), hot_mean=-6.0) for i in range(4)]
if hot:
    for hs in sets:
        for h in hs:
            h[0].reshape(-1, C + 5)[:, C + 4] = 9.0
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
lib = _abi.load()
lib.yh_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
cfg = hd.config_for(sets[0])
cap, mb = 512, w['max_boxes']
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
nbytes = ctypes.c_size_t(0)
_abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
for i in range(12):
    if i == 11:
        torch.cuda.synchronize()
        assert lib.yh_trace_clear() == 0
    hp = _abi.pointer_array([h.data_ptr() for h in sets[i % 4]])
    _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), hp, B, w['thr'], mb, cap, ws.data_ptr(), ws.numel(), st))
torch.cuda.synchronize()
tr = np.zeros((1 << 16, 8), np.uint64)
assert lib.yh_trace_read(tr.ctypes.data, tr.nbytes) == 0
whole = tr[tr[:, 7] == np.uint64(0xffffffffffffffff)]
other = tr[(tr[:, 7] != np.uint64(0xffffffffffffffff)) & (tr[:, 0] > 0)]
def q(x):
    return 'min %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f' % (x.min(), *np.percentile(x, [10, 50, 90]), x.max())
print('%d units read whole, %d records of wavefronts walking row by row' % (len(whole), len(other)))
if len(whole):
    t0 = min(whole[:, 0].min(), other[:, 0].min() if len(other) else whole[:, 0].min())
    us = lambda a: a.astype(np.float64) / 100.0
    print('  start       ', q(us(whole[:, 0] - t0)))
    print('  end         ', q(us(whole[:, 1] - t0)))
    print('  duration    ', q(us(whole[:, 1] - whole[:, 0])))
    print('  load waits  ', q(us(whole[:, 2])), ' (sum over the unit\'s chunks)')
    print('  scanning    ', q(us(whole[:, 3])))
    print('  chunks      ', q(whole[:, 4].astype(np.float64)))
