"""ctypes wrapper of the C restatement (oracle/yolo_head_oracle.c).

TEST INFRASTRUCTURE ONLY (parity unpinned, see yolo_head_oracle.c): used by
tests/, the smoke check and bench.py's cpu_baseline leg, never by the shipped
package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_build', 'libyolo_head_oracle.so')

MAX_SCALES, MAX_PRIORS = 4, 16


class OracleConfig(ctypes.Structure):
    _fields_ = [
        ('num_scales', ctypes.c_int32), ('num_priors', ctypes.c_int32), ('num_classes', ctypes.c_int32),
        ('grid_h', ctypes.c_int32 * MAX_SCALES), ('grid_w', ctypes.c_int32 * MAX_SCALES),
        ('box_mult', ctypes.c_float * MAX_SCALES), ('gt_rate', ctypes.c_float * MAX_SCALES),
        ('prior_hw', ((ctypes.c_float * 2) * MAX_PRIORS) * MAX_SCALES),
    ]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, 'yolo_head_oracle.c')):
        subprocess.run(['make', '-C', _HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.yo_boxes_per_image.restype = ctypes.c_int64
        _lib.yo_max_threads.restype = ctypes.c_int
    return _lib


def make_config(grids, priors, box_mult, gt_rate, num_classes):
    cfg = OracleConfig()
    cfg.num_scales = len(grids)
    cfg.num_priors = len(priors[0])
    cfg.num_classes = int(num_classes)
    for s in range(len(grids)):
        cfg.grid_h[s], cfg.grid_w[s] = int(grids[s][0]), int(grids[s][1])
        cfg.box_mult[s] = float(box_mult[s]); cfg.gt_rate[s] = float(gt_rate[s])
        pr = np.asarray(priors[s], dtype=np.float32).reshape(-1, 2)
        for p in range(pr.shape[0]):
            cfg.prior_hw[s][p][0] = float(pr[p, 0]); cfg.prior_hw[s][p][1] = float(pr[p, 1])
    return cfg


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _heads_ptr(heads):
    hs = [np.ascontiguousarray(h, dtype=np.float32) for h in heads]
    arr = (ctypes.c_void_p * len(hs))(*[h.ctypes.data for h in hs])
    return hs, arr


def decode(cfg, heads, want_conf=True, threads=1):
    lib = load()
    hs, arr = _heads_ptr(heads)
    B = hs[0].shape[0]; N = lib.yo_boxes_per_image(ctypes.byref(cfg)); C = cfg.num_classes
    boxes = np.empty((B, N, 4), np.float32)
    conf = np.empty((B, N, C), np.float32) if want_conf else None
    lib.yo_decode(ctypes.byref(cfg), arr, ctypes.c_int(B), _fp(boxes), _fp(conf) if want_conf else None,
                  ctypes.c_int(threads))
    return boxes, conf


def _unpack(count, scores, boxes, cls, idx):
    out = []
    for b in range(count.shape[0]):
        k = int(count[b])
        out.append((scores[b, :k].copy(), boxes[b, :k].copy(), cls[b, :k].copy(), idx[b, :k].copy()))
    return out


def detect(cfg, heads, score_thr, iou_thr, max_boxes, threads=1, unpack=True):
    lib = load()
    hs, arr = _heads_ptr(heads)
    B = hs[0].shape[0]; C = cfg.num_classes; cap = C * max_boxes
    count = np.zeros((B,), np.int32); scores = np.zeros((B, cap), np.float32)
    boxes = np.zeros((B, cap, 4), np.float32); cls = np.zeros((B, cap), np.int32); idx = np.zeros((B, cap), np.int32)
    rc = lib.yo_detect(ctypes.byref(cfg), arr, ctypes.c_int(B), ctypes.c_float(score_thr), ctypes.c_float(iou_thr),
                       ctypes.c_int(max_boxes), _fp(count), _fp(scores), _fp(boxes), _fp(cls), _fp(idx),
                       ctypes.c_int(threads))
    if rc != 0:
        raise ValueError('yo_detect failed: %d' % rc)
    return _unpack(count, scores, boxes, cls, idx) if unpack else (count, scores, boxes, cls, idx)


def nms(boxes, conf, score_thr, iou_thr, max_boxes, threads=1):
    lib = load()
    boxes = np.ascontiguousarray(boxes, np.float32); conf = np.ascontiguousarray(conf, np.float32)
    B, N, C = conf.shape; cap = C * max_boxes
    count = np.zeros((B,), np.int32); scores = np.zeros((B, cap), np.float32)
    oboxes = np.zeros((B, cap, 4), np.float32); cls = np.zeros((B, cap), np.int32); idx = np.zeros((B, cap), np.int32)
    rc = lib.yo_nms(ctypes.c_int(B), ctypes.c_int64(N), ctypes.c_int(C), _fp(boxes), _fp(conf),
                    ctypes.c_float(score_thr), ctypes.c_float(iou_thr), ctypes.c_int(max_boxes),
                    _fp(count), _fp(scores), _fp(oboxes), _fp(cls), _fp(idx), ctypes.c_int(threads))
    if rc != 0:
        raise ValueError('yo_nms failed: %d' % rc)
    return _unpack(count, scores, oboxes, cls, idx)


def loss(cfg, heads, gt, scales, ignore_thr, normalize, want_iou=False, threads=1):
    """scales = (coord, noobj, obj, class).  Returns per_image [B], parts [B,4],
    iou_max [B,N] or None, mean."""
    lib = load()
    hs, arr = _heads_ptr(heads)
    gt = np.ascontiguousarray(gt, np.float32)
    B, pad = gt.shape[0], gt.shape[1]
    N = lib.yo_boxes_per_image(ctypes.byref(cfg))
    per = np.zeros((B,), np.float32); parts = np.zeros((B, 4), np.float32)
    iou = np.zeros((B, N), np.float32) if want_iou else None
    mean = ctypes.c_float(0)
    w = (ctypes.c_float * 4)(*[float(x) for x in scales])
    lib.yo_loss(ctypes.byref(cfg), arr, _fp(gt), ctypes.c_int(B), ctypes.c_int(pad), w, ctypes.c_float(ignore_thr),
                ctypes.c_int(1 if normalize else 0), _fp(per), _fp(parts), _fp(iou) if want_iou else None,
                ctypes.byref(mean), ctypes.c_int(threads))
    return per, parts, iou, np.float32(mean.value)
