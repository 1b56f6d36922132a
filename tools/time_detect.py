#!/usr/bin/env python3
"""Times DetectionHead.detect on config 2 for several candidate capacities (GPU box)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import yolo_tf_amd as y
dev = torch.device('cuda:0')
s = [8., 16., 32.]
priors = [(np.asarray(y.synth.V3_PRIORS[i], np.float32) / np.float32(s[i])) for i in range(3)]
hd = y.DetectionHead(80, priors, (32., 32., 16.), (32., 16., 8.))
grids = y.synth.v3_grids(416)
sets = [y.synth.torch_heads(dev, i, 64, grids, 3, 80) for i in range(4)]
for cap in [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]:
    outs = hd._alloc_outputs(64, 10, dev, False)
    for i in range(20):
        hd.detect(sets[i % 4], 0.5, 0.5, 10, max_candidates=cap, outputs=outs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for i in range(n):
        r = hd.detect(sets[i % 4], 0.5, 0.5, 10, max_candidates=cap, outputs=outs)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('cap %6d  %.2f us/step  overflow %s  dets/img %.1f' % (cap, 1e6 * el / n, r.overflowed(), r.count.float().mean().item()))
