"""Experiment: two DetectPipelines on two streams, batches alternating between them, so that the
latency-bound tail of one launch overlaps the bandwidth-bound start of the next."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
dev = torch.device('cuda:0')
wk = sys.argv[1] if len(sys.argv) > 1 else 'detect416'
w = bench.WORKLOADS[wk]
grids, priors, box_mult, gt_rate = bench.build_head(w)
B = w['batch']
P = len(priors[0])
sets = [y.synth.torch_heads(dev, i, B, grids, P, w['C'], **w.get('dist', {})) for i in range(4)]
hd0 = y.DetectionHead(w['C'], priors, box_mult, gt_rate)
hd0.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()
cap = hd0.default_max_candidates(sum(h * ww * P for h, ww in grids))
print('capacity', cap)
for nstreams in (1, 2, 3):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    heads_ = [y.DetectionHead(w['C'], priors, box_mult, gt_rate) for _ in range(nstreams)]
    pipes = [y.DetectPipeline(h, w['thr'], w['iou'], w['max_boxes'], max_candidates=cap, stream=s) for h, s in zip(heads_, streams)]
    for i in range(30):
        pipes[i % nstreams].submit(sets[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 600 if wk != 'stress' else 120
    for i in range(K):
        pipes[i % nstreams].submit(sets[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d stream(s): %.2f us/step  %.0f img/s' % (nstreams, dt / K * 1e6, B * K / dt))
    for p in pipes:
        p.flush()
