"""A/B timing of yh_loss (config-3 shape: YOLOv3-416, 100 GT rows per image, batch 32) between two
builds of the library, interleaved in one process: back-to-back calls on one stream under one HIP
event pair (the loss kernel plus its finalize launch).
usage: time_ab_loss.py libA.so libB.so"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi

dev = torch.device('cuda:0')
w = bench.WORKLOADS['loss416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = int(os.environ.get('AB_BATCH', w['batch'])), len(priors[0]), w['C']
cfg = y.head.make_config(grids, priors, box_mult, gt_rate, C)
sets = [y.synth.torch_heads(dev, 40 + i, B, grids, P, C) for i in range(12)]
gts = [torch.from_numpy(y.synth.make_ground_truth(77 + i, B, w['G'], w['pad'], w['size'], C)).to(dev) for i in range(12)]
ptrs = [_abi.pointer_array([h.data_ptr() for h in hs]) for hs in sets]
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
scales = (ctypes.c_float * 4)(1., 1., 5., 1.)
per, total = torch.empty(B, device=dev), torch.empty(2, device=dev)
GRAD = bool(int(os.environ.get('AB_GRAD', 0)))                   # time yh_loss_with_grad instead of yh_loss
gw = torch.full((B,), 0.5 / B, device=dev)
grads = [torch.empty_like(h) for h in sets[0]]
gptr = _abi.pointer_array([g.data_ptr() for g in grads])
libs = []
for path in sys.argv[1:3]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ('yh_loss_workspace_bytes', 'yh_loss', 'yh_loss_with_grad'):
        fn = getattr(lib, name); fn.restype, fn.argtypes = _abi.SIGNATURES[name]
    nbytes = ctypes.c_size_t(0)
    assert lib.yh_loss_workspace_bytes(ctypes.byref(cfg), B, w['pad'], ctypes.byref(nbytes)) == 0
    libs.append((os.path.basename(path), lib, torch.zeros(max(256, nbytes.value), dtype=torch.uint8, device=dev)))
res, vals = {}, {}
for rep in range(5):
    for name, lib, ws in libs:
        def go(i):
            if GRAD:
                rc = lib.yh_loss_with_grad(ctypes.byref(cfg), ptrs[i % 12], gts[i % 12].data_ptr(), B, w['pad'], scales, 0.5, 1,
                                           ws.data_ptr(), ws.numel(), per.data_ptr(), None, None, total.data_ptr(),
                                           gw.data_ptr(), gptr, st)
            else:
                rc = lib.yh_loss(ctypes.byref(cfg), ptrs[i % 12], gts[i % 12].data_ptr(), B, w['pad'], scales, 0.5, 1,
                                 ws.data_ptr(), ws.numel(), per.data_ptr(), None, None, total.data_ptr(), st)
            assert rc == 0, rc
        for i in range(12):
            go(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(120):
            go(i)
        b.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(a.elapsed_time(b) / 120 * 1e3)
        go(0); torch.cuda.synchronize()
        vals[name] = per.clone()
names = list(res)
print('   '.join('%s %7.2f us per call' % (n, np.median(v)) for n, v in res.items()),
      '  per-image losses identical: %s' % bool(torch.equal(vals[names[0]], vals[names[-1]])))
