"""Generates the golden fixtures under tests/golden/ from the NumPy oracle
(oracle/yolo_head_oracle.py).  The reference holds no fixtures of its own and
cannot be run here (TensorFlow 1.12), so these vectors pin the oracle against
accidental change, not against the reference: parity stays "unpinned".

Run from the repo root:  python tests/golden/make_golden.py
Cases are small, seeded, and margin-safe: no confidence within 1e-4 (relative)
of the score threshold, no pairwise IoU within 1e-4 of the NMS threshold among
candidates, no equal scores inside a class, no ignore-mask IoU within 1e-4 of
its threshold."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import yolo_tf_amd as y                                   # noqa: E402
from oracle import yolo_head_oracle as O                  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
syn = y.synth


def margins_ok(bbox, conf, thr, iou_thr):
    if np.min(np.abs(conf.astype(np.float64) - thr)) / thr < 1e-4:
        return False
    for c in range(conf.shape[1]):
        rows = np.nonzero(conf[:, c] >= np.float32(thr))[0]
        s = conf[rows, c]
        if len(np.unique(s)) != len(s):
            return False
        for i in range(len(rows)):
            for j in range(i):
                bi, bj = bbox[rows[i]].astype(np.float64), bbox[rows[j]].astype(np.float64)
                iy = max(min(bi[2], bj[2]) - max(bi[0], bj[0]), 0.0)
                ix = max(min(bi[3], bj[3]) - max(bi[1], bj[1]), 0.0)
                a = (bi[2] - bi[0]) * (bi[3] - bi[1]) + (bj[2] - bj[0]) * (bj[3] - bj[1]) - iy * ix
                if a > 0 and abs(iy * ix / a - iou_thr) < 1e-4:
                    return False
    return True


def detect_case(name, grids, priors, mult, C, B, thr, iou_thr, mb, seed0, **dist):
    P = np.asarray(priors[0]).shape[0]
    seed = seed0
    while True:
        heads = syn.make_heads(seed, B, grids, P, C, **dist)
        dec = [O.decode([h[b] for h in heads], priors, mult, C) for b in range(B)]
        if all(margins_ok(bb, cf, thr, iou_thr) for bb, cf in dec):
            break
        seed += 1000
    out = dict(seed=np.int64(seed), num_classes=np.int32(C), score_threshold=np.float32(thr),
               iou_threshold=np.float32(iou_thr), max_boxes=np.int32(mb),
               grids=np.asarray(grids, np.int32), box_mult=np.asarray(mult, np.float32),
               priors=np.stack([np.asarray(p, np.float32) for p in priors]))
    for s, h in enumerate(heads):
        out['head%d' % s] = h
    out['bbox'] = np.stack([d[0] for d in dec]); out['conf'] = np.stack([d[1] for d in dec])
    counts, sc, bx, cl, ix = [], [], [], [], []
    for bb, cf in dec:
        s_, b_, c_, i_ = O.nms_per_class(bb, cf, thr, mb, iou_thr)
        counts.append(len(s_)); sc.append(s_); bx.append(b_); cl.append(c_); ix.append(i_)
    out['det_count'] = np.asarray(counts, np.int32)
    out['det_scores'] = np.concatenate(sc); out['det_bbox'] = np.concatenate(bx, axis=0)
    out['det_class'] = np.concatenate(cl); out['det_index'] = np.concatenate(ix)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'seed', seed, 'detections', counts)


def loss_case(name, version, grids, priors, rates, C, B, G, pad, size, seed0):
    P = np.asarray(priors[0]).shape[0]
    thr = 0.5 if version == 3 else 0.6
    seed = seed0
    while True:
        heads = syn.make_heads(seed, B, grids, P, C)
        gt = syn.make_ground_truth(seed + 1, B, G, pad, size, C)
        for b in range(B):                                  # ragged GT counts
            gt[b, max(1, G - 2 * b):] = -1.0
        ious = [O.noobj_iou_max([h[b] for h in heads], gt[b], priors, rates, C) for b in range(B)]
        if all(np.nanmin(np.abs(i.astype(np.float64) - thr)) > 1e-4 for i in ious):
            break
        seed += 1000
    if version == 3:
        per, mean = O.loss_v3_batch(heads, gt, priors, C)
        parts = [O.loss_v3_image([h[b] for h in heads], gt[b], priors, C, return_parts=True)[1] for b in range(B)]
    else:
        per, mean = O.loss_v2_batch(heads[0], gt, priors[0], C)
        parts = [O.loss_v2_image(heads[0][b], gt[b], priors[0], C, return_parts=True)[1] for b in range(B)]
    out = dict(seed=np.int64(seed), version=np.int32(version), num_classes=np.int32(C),
               grids=np.asarray(grids, np.int32), gt_rate=np.asarray(rates, np.float32),
               priors=np.stack([np.asarray(p, np.float32) for p in priors]), ground_truth=gt,
               loss_scales=np.asarray([1., 1., 5., 1.], np.float32), ignore_threshold=np.float32(thr),
               loss_per_image=per, loss_mean=np.float32(mean),
               loss_parts=np.asarray([[p['coord'], p['cls'], p['obj'], p['noobj']] for p in parts], np.float32),
               iou_max=np.stack(ious))
    for s, h in enumerate(heads):
        out['head%d' % s] = h
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'seed', seed, 'loss', per)


if __name__ == '__main__':
    v3p = O.v3_grid_priors(syn.V3_PRIORS)
    v2p = [np.asarray(syn.V2_PRIORS, np.float32)]
    # config 1 of the benchmark list: YOLOv2 VOC-20 416x416 batch 1 (+1 image)
    detect_case('detect_v2_416_voc', [(13, 13)], v2p, O.V2_BOX_MULT, 20, 2, 0.5, 0.5, 10, 1, obj_mean=0.0)
    # YOLOv3 geometry at 160x160 (grids 5/10/20), 12 classes
    detect_case('detect_v3_160', syn.v3_grids(160), v3p, O.V3_BOX_MULT, 12, 2, 0.3, 0.45, 8, 2, obj_mean=-1.0)
    loss_case('loss_v3_160', 3, syn.v3_grids(160), v3p, O.V3_GT_RATE, 12, 3, 9, 12, 160, 3)
    loss_case('loss_v2_416_voc', 2, [(13, 13)], v2p, O.V2_GT_RATE, 20, 3, 7, 60, 416, 4)
