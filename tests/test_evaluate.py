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
