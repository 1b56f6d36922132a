"""Experiment: two DetectPipelines on two streams, batches alternating between them, so that the
latency-bound tail of one launch overlaps the bandwidth-bound start of the next."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
dev = torch.device('cuda:0')
w = bench.WORKLOADS['detect416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
sets = [y.synth.torch_heads(dev, i, 64, grids, 3, 80) for i in range(4)]
for nstreams in (1, 2, 3):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    heads_ = [y.DetectionHead(80, priors, box_mult, gt_rate) for _ in range(nstreams)]
    pipes = [y.DetectPipeline(h, 0.5, 0.5, 10, stream=s) for h, s in zip(heads_, streams)]
    for i in range(30):
        pipes[i % nstreams].submit(sets[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 600
    for i in range(K):
        pipes[i % nstreams].submit(sets[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d stream(s): %.2f us/step  %.0f img/s' % (nstreams, dt / K * 1e6, 64 * K / dt))
    for p in pipes:
        p.flush()
