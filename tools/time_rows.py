"""Where a launch of the objectness-first kernel spends its time (config-2 shape): the streaming pass
alone (yh_detect_candidates), the select stage alone (yh_detect_select), both in one launch
(yh_detect_pipelined, one stream), back-to-back launches under one HIP event pair each."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi

dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'detect416'
w = bench.WORKLOADS[wl]
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
ROT = 9
sets = [y.synth.torch_heads(dev, i, B, grids, P, C) for i in range(ROT)]
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
lib = _abi.load()
cfg = hd.config_for(sets[0])
cap, mb = 512, w['max_boxes']
nbytes = ctypes.c_size_t(0)
_abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
outs = hd._alloc_outputs(B, mb, dev, False)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
ptrs = [_abi.pointer_array([h.data_ptr() for h in hs]) for hs in sets]


def timed(fn, n=200, runs=5):
    for i in range(20):
        fn(i)
    res = []
    for _ in range(runs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n):
            fn(i)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / n * 1e3)
    return np.median(res), res


def cand(i):
    _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), ptrs[i % ROT], B, w['thr'], mb, cap, ws.data_ptr(), ws.numel(), st))


def select(i):
    _abi.check(lib.yh_detect_select(ctypes.byref(cfg), B, w['iou'], mb, cap, ws.data_ptr(), ws.numel(),
                                    *hd._output_pointers(outs), st))


pipe = y.DetectPipeline(hd, w['thr'], w['iou'], mb)
print('streaming pass alone   %.2f us' % timed(cand)[0])
cand(0)
print('select stage alone     %.2f us (same candidates every time)' % timed(select)[0])
print('both in one launch     %.2f us' % timed(lambda i: pipe.submit(sets[i % ROT]))[0])
pipe.flush()
print('streaming, then select %.2f us (two launches)' % timed(lambda i: (cand(i), select(i)))[0])
