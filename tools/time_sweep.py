"""The rows kernel against the share of rows that pass the objectness prefilter, for several values of
yh_tuning.rows_dense_min (65 = never read a step whole):  time_sweep.py [workload] [dense_min,...] [hot_mean]
Microseconds per launch, launches back to back on one stream (the select stage of the previous batch rides
along), and whether every setting gives the detections of the first one."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi

dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'detect416'
mins = sys.argv[2].split(',') if len(sys.argv) > 2 else ['65', '0']       # dense_min[:probe]
hot = float(sys.argv[3]) if len(sys.argv) > 3 else -6.0
mus = [float(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else (-4.0, -2.0, -1.0, 0.0, 1.0, 2.0, 9.0)
if os.environ.get('YH_LIB'):
    _abi.LIB_PATH = os.path.abspath(os.environ['YH_LIB'])
w = bench.WORKLOADS[wl]
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
N = sum(h * ww * P for h, ww in grids)
for mu in mus:
    import math
    nominal = 0.5 * (1.0 + math.erf(mu / 2.0 / math.sqrt(2.0)))
    touched = bench.rows_traffic_model(B * N, nominal, C + 5)
    n_sets = bench.rotation_for(touched, floor=3, cap=10)
    sets = [y.synth.torch_heads(dev, 500 + i, B, grids, P, C, obj_mean=mu, hot_mean=hot) for i in range(n_sets)]
    if os.environ.get('SWEEP_HOT_IMAGE'):                   # image 0 of every set: every row survives (a few units read whole among units walked row by row)
        for hs in sets:
            for h in hs:
                h[0].reshape(-1, C + 5)[:, C + 4] = 9.0
    surv = bench.row_survival(sets[:1], C, w['thr'])
    line, ref = 'mu %5.1f survival %.3f sets %2d:' % (mu, surv, n_sets), None
    for dm in mins:
        _abi.set_tuning(rows_dense_min=int(dm.split(':')[0]), rows_probe=int(dm.split(':')[1]) if ':' in dm else 0)
        hd = y.DetectionHead(C, priors, box_mult, gt_rate)
        runs, last = bench.solo_kernel_ms(hd, w, sets, dev, per_run=60, runs=3)
        got = hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()
        key = [tuple(np.array(a).copy() for a in img) for img in got.per_image()]
        same = True
        if ref is None:
            ref = key
        else:
            same = all(np.array_equal(a, b) for ia, ib in zip(ref, key) for a, b in zip(ia, ib))
        line += '  dm %4s: %6.2f us%s' % (dm, float(np.median(runs)) * 1e3, '' if same else ' DIFFERS')
        del hd
    print(line + '  (detections %d)' % sum(len(i[0]) for i in ref), flush=True)
    del sets
    torch.cuda.empty_cache()
