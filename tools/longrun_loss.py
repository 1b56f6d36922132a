"""Long run of the training path at the config-3 shape: loss with the fused gradient, every other step on a
prepared workspace, rotating ground truth and head tensors on one head object; every step's per-image losses and
gradient tensors are compared, bit for bit, with what a fresh head object gave for the same inputs before the run.
usage: longrun_loss.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device('cuda:0')
w = bench.WORKLOADS['train416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
nsets = 4
sets = [y.synth.torch_heads(dev, 800 + i, B, grids, P, C) for i in range(nsets)]
gts = [torch.from_numpy(y.synth.make_ground_truth(810 + i, B, w['G'], w['pad'], w['size'], C)).to(dev) for i in range(nsets)]
gw = torch.full((B,), 1.0 / B, device=dev)
args = (1., 1., 5., 1., 0.5, True)
want = []
for h, g in zip(sets, gts):
    o = y.DetectionHead(C, priors, box_mult, gt_rate).loss(h, g, *args, grad_per_image=gw)
    want.append((o['per_image'].clone(), [t.clone() for t in o['grads']]))
fwd = [y.DetectionHead(C, priors, box_mult, gt_rate).loss(h, g, *args)['per_image'].clone() for h, g in zip(sets, gts)]
torch.cuda.synchronize()
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
bad, t0 = 0, time.time()
for i in range(steps):
    k = i % nsets
    if i & 1:
        hd.loss_prepare(sets[k], gts[k])
    if i % 3 == 2:                                   # a forward-only call in between (the last workgroup finalizes)
        o = hd.loss(sets[k], gts[k], *args)
        ok = torch.equal(o['per_image'], fwd[k])
    else:
        o = hd.loss(sets[k], gts[k], *args, grad_per_image=gw)
        ok = torch.equal(o['per_image'], want[k][0]) and all(torch.equal(a, b) for a, b in zip(o['grads'], want[k][1]))
    if not ok:
        bad += 1
        print('step %d (set %d, prepared %s, gradient %s): outputs differ' % (i, k, bool(i & 1), i % 3 != 2), flush=True)
    if i % 5000 == 4999:
        print('%d steps, %d differ, %.0f s' % (i + 1, bad, time.time() - t0), flush=True)
print('done: train416, %d steps, %d differ' % (steps, bad))
sys.exit(1 if bad else 0)
