"""A/B timing of the streaming pass (yh_detect_candidates alone, one stream, back to back) between
builds of the library (two or more), interleaved in one process so that all see the same chip and clocks:
config-2 shape at several row-survival rates (objectness ~ N(mu, 2), class logits too low for any
candidate except at the benchmark's own distribution).
usage: time_ab.py libA.so libB.so [libC.so ...]   (AB_STEPS=n: row_steps for all)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi

dev = torch.device('cuda:0')
w = bench.WORKLOADS[os.environ.get('AB_WORKLOAD', 'detect416')]
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = int(os.environ.get('AB_BATCH', w['batch'])), len(priors[0]), w['C']
cfg = y.head.make_config(grids, priors, box_mult, gt_rate, C)
libs = []
for k, path in enumerate(a for a in sys.argv[1:] if a.endswith('.so')):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ('yh_detect_workspace_bytes', 'yh_detect_candidates', 'yh_set_tuning'):
        fn = getattr(lib, name); fn.restype, fn.argtypes = _abi.SIGNATURES[name]
    steps = int(os.environ.get('AB_STEPS', 0))
    t = _abi.Tuning(row_steps=steps, stream_blocks=int(os.environ.get('AB_BLOCKS', 0)), rows_dense_min=int(os.environ.get('AB_DENSE_MIN', 0)),
                    rows_probe=int(os.environ.get('AB_PROBE', 0)))
    assert lib.yh_set_tuning(ctypes.byref(t)) == 0
    libs.append((os.path.basename(path) + ('/steps%d' % steps if steps else ''), lib))
cap, mb = int(os.environ.get('AB_CAP', 2048)), w['max_boxes']   # AB_CAP above the select stage's limit: the workgroup-tile variant
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
only = os.environ.get('AB_ONLY_BENCH')
for label, dist in (('benchmark distribution', {}),) if only else (('benchmark distribution', {}), ('mu -4, no candidates', dict(obj_mean=-4.0, hot_mean=-6.0)),
                    ('mu -2', dict(obj_mean=-2.0, hot_mean=-6.0)), ('mu 0', dict(obj_mean=0.0, hot_mean=-6.0)),
                    ('mu +2', dict(obj_mean=2.0, hot_mean=-6.0)), ('mu +9', dict(obj_mean=9.0, hot_mean=-6.0))):
    sets = [y.synth.torch_heads(dev, 40 + i, B, grids, P, C, **dict(w['dist'], **dist)) for i in range(6)]
    surv = bench.row_survival(sets[:1], C, w['thr'])
    ptrs = [_abi.pointer_array([h.data_ptr() for h in hs]) for hs in sets]
    res = {}
    for rep in range(3):
        for name, lib in libs:
            nbytes = ctypes.c_size_t(0)
            assert lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)) == 0
            ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
            def go(i):
                rc = lib.yh_detect_candidates(ctypes.byref(cfg), ptrs[i % 6], B, w['thr'], mb, cap, ws.data_ptr(), ws.numel(), st)
                assert rc == 0, rc
            for i in range(12):
                go(i)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(60):
                go(i)
            b.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(a.elapsed_time(b) / 60 * 1e3)
    model = bench.rows_traffic_model(B * sum(h * ww * P for h, ww in grids), surv, C + 5)
    print('%-24s survival %5.1f%%  bound %6.1f us  ' % (label, 100 * surv, max(14.0, 1.25 * model / 6e6)) +
          '   '.join('%s %7.2f us' % (n, np.median(v)) for n, v in res.items()), flush=True)
    del sets
