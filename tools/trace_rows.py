"""Per-wavefront timeline of k_detect_rows (measurement build: make -C yolo-tf_amd/csrc trace).
Launches the streaming pass of the config-2 shape a few times, reads the timestamps of the last
launch and prints where the wavefronts spend their time (microseconds since the first wavefront's
start; the clock ticks at 100 MHz)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), 'libyolohead_trace.so')

dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'detect416'
w = bench.WORKLOADS[wl]
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
sets = [y.synth.torch_heads(dev, i, B, grids, P, C) for i in range(9)]
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
lib = _abi.load()
lib.yh_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
cfg = hd.config_for(sets[0])
cap, mb = 512, w['max_boxes']
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for S in [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0'])]:
    _abi.set_tuning(row_steps=S)
    nbytes = ctypes.c_size_t(0)
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
    ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
    for i in range(12):
        if i == 11:
            torch.cuda.synchronize()
            assert lib.yh_trace_clear() == 0
        hp = _abi.pointer_array([h.data_ptr() for h in sets[i % 9]])
        _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), hp, B, w['thr'], mb, cap, ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
    tr = np.zeros((1 << 16, 8), np.uint64)
    assert lib.yh_trace_read(tr.ctypes.data, tr.nbytes) == 0
    tr = tr[(tr[:, 0] > 0) & (tr[:, 1] > 0)]
    t = tr[:, :5].astype(np.float64) / 100.0                 # microseconds
    t -= t[:, 0].min()
    queued, left = (tr[:, 6] >> np.uint64(32)).astype(np.int64), (tr[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
    tB, tC = tr[:, 7].astype(np.float64) / 100.0 - tr[:, 0].astype(np.float64).min() / 100.0, tr[:, 5].astype(np.float64) / 100.0 - tr[:, 0].astype(np.float64).min() / 100.0
    some = left > 0
    def q(x):
        return 'min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f' % (x.min(), *np.percentile(x, [10, 50, 90]), x.max())
    print('row_steps %d: %d wavefronts with rows, %d survivors queued (%.1f per wavefront, max %d)' % (
        S, len(t), queued.sum(), queued.mean(), queued.max()))
    print('  start                    ', q(t[:, 0]))
    print('  first column data at     ', q(t[:, 1]))
    print('  column walk done at      ', q(t[:, 2]))
    print('  last survivors done at   ', q(t[:, 3]))
    print('  stores acknowledged at   ', q(t[:, 4]))
    print('  first column round trip  ', q(t[:, 1] - t[:, 0]))
    print('  column walk              ', q(t[:, 2] - t[:, 0]))
    print('  final survivors          ', q(t[:, 3] - t[:, 2]), ' (%d wavefronts had some)' % int((left > 0).sum()))
    print('  flush + store acks       ', q(t[:, 4] - t[:, 3]))
    print('  last batch: gather wait  ', q((tB - t[:, 2])[some]))
    print('  last batch: arithmetic   ', q((tC - tB)[some]))
    late = t[:, 0] > 3.0
    print('  wavefronts starting after 3 us: %d' % late.sum(), flush=True)
