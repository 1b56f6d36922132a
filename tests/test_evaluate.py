"""mAP evaluator (the consumer after test_one_image): hand-computed PASCAL VOC cases."""
import numpy as np
import pytest

import yolo_tf_amd as y
from yolo_tf_amd import evaluate as E


def _case():
    # image 0: ground truth A, B of class 0 and D of class 1; image 1: C of class 0
    gt = y.wire.encode_ground_truth_batch(
        [[[0, 0, 10, 10], [20, 20, 30, 30], [40, 40, 60, 60]], [[0, 0, 10, 10]]], [[0, 0, 1], [0]], (100, 100), 5)
    det0 = (np.array([.9, .8, .6, .5], np.float32),
            np.array([[0, 0, 10, 10], [1, 1, 10, 10], [50, 50, 60, 60], [20, 20, 30, 30]], np.float32),
            np.array([0, 0, 0, 0], np.int32))
    det1 = (np.array([.7], np.float32), np.array([[0, 0, 10, 5]], np.float32), np.array([0], np.int32))
    return [det0, det1], gt


def test_iou():
    iou = E.box_iou([0, 0, 10, 10], [[0, 0, 10, 10], [1, 1, 10, 10], [0, 0, 10, 5], [20, 20, 30, 30], [5, 5, 5, 5]])
    np.testing.assert_allclose(iou, [1.0, 0.81, 0.5, 0.0, 0.0])


def test_class_ap_by_hand():
    """Score order: TP (A), FP (A again), TP (C, IoU exactly 0.5), FP (no overlap), TP (B): recall 1/3 1/3 2/3 2/3 1,
    precision 1 1/2 2/3 1/2 3/5; envelope area (1 + 2/3 + 3/5) / 3; 11-point mean (4*1 + 3*2/3 + 4*3/5) / 11."""
    dets, gt = _case()
    gts = [y.wire.decode_ground_truth(b) for b in gt]
    ap, rec, prec, n = E.class_average_precision(dets, gts, 0)
    assert n == 3
    np.testing.assert_allclose(rec, [1 / 3, 1 / 3, 2 / 3, 2 / 3, 1.0])
    np.testing.assert_allclose(prec, [1.0, 0.5, 2 / 3, 0.5, 0.6])
    assert ap == pytest.approx((1 + 2 / 3 + 0.6) / 3)
    ap07, _, _, _ = E.class_average_precision(dets, gts, 0, use_07_metric=True)
    assert ap07 == pytest.approx((4 * 1 + 3 * 2 / 3 + 4 * 0.6) / 11)
    # a stricter overlap turns the IoU-0.5 match into a false positive: TP FP FP FP TP -> recall 1/3 .. 2/3
    ap75, rec75, _, _ = E.class_average_precision(dets, gts, 0, iou_threshold=0.75)
    np.testing.assert_allclose(rec75, [1 / 3, 1 / 3, 1 / 3, 1 / 3, 2 / 3])
    assert ap75 == pytest.approx(1 / 3 * 1.0 + 1 / 3 * 0.4)


def test_map_leaves_out_classes_without_ground_truth():
    dets, gt = _case()
    res = E.mean_average_precision(dets, gt, num_classes=3)
    assert set(res['ap']) == {0, 1} and res['num_gt'] == {0: 3, 1: 1}
    assert res['ap'][1] == 0.0                                 # ground truth D, never detected
    assert res['map'] == pytest.approx(((1 + 2 / 3 + 0.6) / 3 + 0.0) / 2)
    perfect = [(np.ones(len(c), np.float32), b, c) for b, c in (y.wire.decode_ground_truth(blk) for blk in gt)]
    assert E.mean_average_precision(perfect, gt, 3)['map'] == pytest.approx(1.0)
    with pytest.raises(ValueError):
        E.mean_average_precision(dets[:1], gt, 3)
    empty = E.mean_average_precision([(np.zeros(0), np.zeros((0, 4)), np.zeros(0, np.int32))] * 2, gt, 3)
    assert empty['map'] == 0.0


def test_equal_scores_keep_image_and_detection_order():
    gt = y.wire.encode_ground_truth_batch([[[0, 0, 10, 10]], [[0, 0, 10, 10]]], [[0], [0]], (50, 50), 3)
    dets = [(np.array([.5, .5], np.float32), np.array([[30, 30, 40, 40], [0, 0, 10, 10]], np.float32), np.zeros(2, np.int32)),
            (np.array([.5], np.float32), np.array([[0, 0, 10, 10]], np.float32), np.zeros(1, np.int32))]
    _, rec, prec, _ = E.class_average_precision(dets, [y.wire.decode_ground_truth(b) for b in gt], 0)
    np.testing.assert_allclose(rec, [0.0, 0.5, 1.0])
    np.testing.assert_allclose(prec, [0.0, 0.5, 2 / 3])


# ---- the consumers on the HIP path's own outputs (GPU) ---------------------------------------------------
@pytest.mark.gpu
def test_map_of_the_hip_detections_equals_map_of_the_oracle_detections(gpu, c_oracle):
    """Sixteen synthetic images through YOLOv3.test_one_image (the HIP path behind the class API) and through
    the oracle's decode + NMS; ground truth built from a few of each image's own detections (shifted) plus
    random boxes, in the wire format.  The evaluator must give identical AP for every class: the kept rows are
    bit-exact and a 1e-5 difference in a box or score moves no detection across the IoU 0.5 line nor past a
    neighbour in the score order."""
    import torch
    from tests import cases
    C, size, n_img = 6, 160, 16
    case = cases.v3(size=size, C=C)
    cfg = dict(mode='test', data_shape=[size, size, 3], num_classes=C, weight_decay=5e-4, keep_prob=0.5,
               data_format='channels_last', batch_size=1, coord_scale=1, noobj_scale=1, obj_scale=5., class_scale=1.,
               num_priors=3, nms_score_threshold=0.3, nms_max_boxes=10, nms_iou_threshold=0.5, priors=y.synth.V3_PRIORS)
    net = y.YOLOv3(cfg, None)
    heads = case.heads(4711, n_img, obj_mean=-1.5)
    want = c_oracle.detect(case.oracle_cfg(c_oracle), heads, 0.3, 0.5, 10)
    got = []
    for i in range(n_img):
        pred = net.test_one_image([torch.from_numpy(h[i:i + 1]).to(gpu) for h in heads])
        assert np.array_equal(pred[2], want[i][2]) and len(pred[0]) == len(want[i][0])
        got.append(pred)
    rng = np.random.Generator(np.random.PCG64(4712))
    boxes, classes = [], []
    for i in range(n_img):
        take = rng.permutation(len(want[i][0]))[:4]
        bx = [np.clip(want[i][1][k] + rng.uniform(-6, 6, size=4), 1, size - 2) for k in take]
        cl = [int(want[i][2][k]) for k in take]
        for _ in range(3):                                     # ground truth nothing was detected at
            y1, x1 = rng.uniform(1, size - 40, size=2)
            bx.append(np.array([y1, x1, y1 + rng.uniform(8, 38), x1 + rng.uniform(8, 38)]))
            cl.append(int(rng.integers(0, C)))
        boxes.append(bx)
        classes.append(cl)
    gt = y.wire.encode_ground_truth_batch(boxes, classes, (size, size), 12)
    m_hip = E.mean_average_precision(got, gt, C)
    m_ora = E.mean_average_precision([(w[0], w[1], w[2]) for w in want], gt, C)
    assert sum(len(g[0]) for g in got) > 40 and len(m_hip['ap']) == C
    assert m_hip['num_gt'] == m_ora['num_gt']
    for c in range(C):
        assert m_hip['ap'][c] == m_ora['ap'][c], 'class %d: AP %r from the HIP path, %r from the oracle' % (c, m_hip['ap'][c], m_ora['ap'][c])
    assert 0.0 < m_hip['map'] < 1.0
    # the same detections through the record / class-name consumer
    rec = y.wire.detections_to_records(*got[0], names={i: 'c%d' % i for i in range(C)})
    assert len(rec) == len(got[0][0]) and rec[0]['name'] == 'c%d' % int(got[0][2][0])


@pytest.mark.gpu
def test_wire_format_blocks_feed_the_loss_kernel(gpu, c_oracle):
    """encode_ground_truth_batch (utils/image_augmentor.py:205-244: centre filter, clamp, -1 padding, ragged
    counts per image) -> yh_loss on the GPU against the oracle's loss on the same blocks, 1e-5."""
    import torch
    from tests import cases
    from tests.test_gpu_parity import assert_close
    C, size, B = 6, 160, 5
    case = cases.v3(size=size, C=C)
    rng = np.random.Generator(np.random.PCG64(99))
    boxes, classes = [], []
    for b in range(B):
        n = int(rng.integers(1, 9))
        y1x1 = rng.uniform(-20, size - 10, size=(n, 2))        # some centres fall outside and are dropped, some boxes are clamped
        hw = rng.uniform(6, 90, size=(n, 2))
        bx = np.concatenate([y1x1, y1x1 + hw], axis=1)
        bx[0] = [20 + b, 30, 70 + b, 95]                        # at least one box survives in every image
        boxes.append(bx)
        classes.append(rng.integers(0, C, size=n))
    gt = y.wire.encode_ground_truth_batch(boxes, classes, (size, size), 10)
    counts = [(blk[:, 0] >= 0).sum() for blk in gt]
    assert min(counts) >= 1 and len(set(counts)) > 1            # ragged
    heads = case.heads(100, B)
    out = case.head_obj().loss([torch.from_numpy(h).to(gpu) for h in heads], torch.from_numpy(gt).to(gpu),
                               1., 1., 5., 1., 0.5, True, want_parts=True)
    per, parts, _, mean = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, (1., 1., 5., 1.), 0.5, True)
    assert_close(out['parts'].cpu().numpy(), parts, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, what='per-image loss')
    assert np.all(np.isfinite(per))
