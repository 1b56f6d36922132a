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


steps_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for S in steps_list:
    _abi.set_tuning(row_steps=S)
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
    ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
    hd = y.DetectionHead(C, priors, box_mult, gt_rate)
    pipe = y.DetectPipeline(hd, w['thr'], w['iou'], mb)
    t_c = timed(cand)[0]
    cand(0)
    t_s = timed(select)[0]
    t_f = timed(lambda i: pipe.submit(sets[i % ROT]))[0]
    pipe.flush()
    lanes = y.DetectLanes(hd, w['thr'], w['iou'], mb, lanes=3, device=dev)
    for i in range(30):
        lanes.submit(sets[i % ROT])
    lanes.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(1000):
        lanes.submit(sets[i % ROT])
    lanes.synchronize()
    t_l = (time.perf_counter() - t0) / 1000 * 1e6
    ov = any(d.overflowed() for d in lanes.flush())
    print('row_steps %2d: streaming pass alone %6.2f us   select alone %6.2f   both in one launch %6.2f   three lanes %6.2f us/step   overflow %s'
          % (S, t_c, t_s, t_f, t_l, ov), flush=True)
