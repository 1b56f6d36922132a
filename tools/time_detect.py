"""Times the pipelined detect step of the config-2 shape for several score thresholds
(a threshold near 1 leaves no surviving rows: the floor of the objectness pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
dev = torch.device('cuda:0')
w = bench.WORKLOADS['detect416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
hd = y.DetectionHead(80, priors, box_mult, gt_rate)
sets = [y.synth.torch_heads(dev, i, 64, grids, 3, 80) for i in range(4)]
for thr in [float(a) for a in sys.argv[1:]] or [0.5, 0.9, 0.999999]:
    pipe = y.DetectPipeline(hd, thr, 0.5, 10)
    for i in range(20):
        pipe.submit(sets[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        pipe.submit(sets[i % 4])
    e1.record(); torch.cuda.synchronize()
    last = pipe.flush()
    print('thr %-9g %.2f us/step   detections in last batch %d' % (thr, e0.elapsed_time(e1) * 1000 / 200, int(last.count.sum())))
