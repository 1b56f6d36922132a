"""Where a short timed region of the detect416 benchmark spends its time on the host: regions of K
steps between device synchronisations, as bench.py runs them, with the duration of every submit."""
import gc, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
w = bench.WORKLOADS['detect416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
sets = [y.synth.torch_heads(dev, i, B, grids, P, C) for i in range(9)]
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()
pipe = y.DetectLanes(hd, w['thr'], w['iou'], w['max_boxes'], lanes=lanes, device=dev)
gc.collect(); gc.disable()
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 100):
    pipe.submit(sets[i % 9])
torch.cuda.synchronize()
for rep in range(8):
    t = np.zeros(K + 2)
    t[0] = time.perf_counter()
    for i in range(K):
        pipe.submit(sets[i % 9])
        t[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    t[K + 1] = time.perf_counter()
    d = np.diff(t) * 1e6
    print('region %d: %.1f us, %.2f us per step; submits: first %.1f, median %.1f, max %.1f (at %d); wait at the end %.1f us'
          % (rep, (t[-1] - t[0]) * 1e6, (t[-1] - t[0]) * 1e6 / K, d[0], np.median(d[:K]), d[:K].max(), int(d[:K].argmax()), d[K]))
    print('   ' + ' '.join('%.0f' % x for x in d[:K]), flush=True)
