"""Experiment: two-lane step time of the config-2 shape against the score threshold (how much of a
step is the survivors' work and how much the objectness column)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
dev = torch.device('cuda:0')
w = bench.WORKLOADS['detect416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P = w['batch'], len(priors[0])
sets = [y.synth.torch_heads(dev, i, B, grids, P, w['C']) for i in range(4)]
hd = y.DetectionHead(w['C'], priors, box_mult, gt_rate)
for thr in (0.5, 0.9, 0.99, 0.9999):
    for lanes in (1, 2):
        pipe = y.DetectLanes(hd, thr, 0.5, 10, lanes=lanes, device=dev)
        for i in range(100):
            pipe.submit(sets[i % 4])
        pipe.synchronize()
        K = 2000
        t0 = time.perf_counter()
        for i in range(K):
            pipe.submit(sets[i % 4])
        pipe.synchronize()
        dt = time.perf_counter() - t0
        n = sum(int(d.count.sum()) for d in pipe.flush())
        print('thr %.4f  %d lane(s): %.2f us/step  (%d detections in the last batches)' % (thr, lanes, dt / K * 1e6, n))
