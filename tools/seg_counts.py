"""Candidates per (image, class) segment of the stress workload (diagnostic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
dev = torch.device('cuda:0')
w = bench.WORKLOADS['stress']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = 8, len(priors[0]), w['C']
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
heads = y.synth.torch_heads(dev, 1, B, grids, P, C, **(w.get('dist') or {}))
det = hd.detect(heads, w['thr'], w['iou'], w['max_boxes'], want_class_candidates=True).resolved()
cc = det.class_candidates.cpu().numpy()
print('segments', cc.shape, 'min', cc.min(), 'median', np.median(cc), 'max', cc.max(), 'over 2048:', (cc > 2048).sum())
print('detections per image', det.count.cpu().numpy())
