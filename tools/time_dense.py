"""How the detect kernels behave when many rows survive the objectness prefilter
(objectness logits ~ N(obj_mean, 2), threshold 0.5): objectness-first vs streaming."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi
dev = torch.device('cuda:0')
w = bench.WORKLOADS['detect416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
for obj_mean in (-4.0, -2.0, -1.0, 0.0, 2.0):
    sets = [y.synth.torch_heads(dev, i, 64, grids, 3, 80, obj_mean=obj_mean, hot_mean=-3.0) for i in range(4)]
    row = []
    for kind in ('1', '2'):
        _abi.set_tuning(detect_kernel=int(kind))
        hd = y.DetectionHead(80, priors, box_mult, gt_rate)
        pipe = y.DetectPipeline(hd, 0.5, 0.5, 10)
        for i in range(10):
            pipe.submit(sets[i % 4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            pipe.submit(sets[i % 4])
        e1.record(); torch.cuda.synchronize()
        last = pipe.flush()
        row.append('%s %.1f us' % ('rows' if kind == '1' else 'stream', e0.elapsed_time(e1) * 10))
    frac = float((sets[0][2][..., 84::85] >= 0).float().mean())
    print('obj_mean %+.0f  surviving rows %.1f%%   %s   overflow %s' % (obj_mean, 100 * frac, '   '.join(row), last.overflowed()))
