"""Mean average precision of detections against ground truth: the consumer that
follows `test_one_image` in a detection workflow.  The reference stops at
plotting (testYOLOv3.py:81-97) and holds no evaluation code; this module follows
the published PASCAL VOC protocol on the reference's own data formats:

  detections    per image what test_one_image returns (YOLOv3.py:365-368):
                scores [K], bbox [K, 4] as (y1, x1, y2, x2) pixels, class_id [K];
  ground truth  per image the [pad, 5] block of (yc, xc, h, w, class) rows,
                -1 padded (utils/image_augmentor.py:205-244).

Protocol, per class: all images' detections of the class in descending score
order (equal scores: image order, then detection order); a detection is a true
positive when the ground-truth box of its image and class that overlaps it most
has IoU >= iou_threshold and has not been claimed by an earlier detection,
otherwise a false positive; precision and recall follow the cumulative counts;
AP is the area under the precision envelope (VOC2010+) or the 11-point
interpolation (VOC2007).  Classes without ground truth are left out of the mean.
Coordinates are continuous (no +1 pixel convention); the wire format carries no
"difficult" flag, so every ground-truth box counts."""
import numpy as np

from .wire import decode_ground_truth


def box_iou(box, boxes):
    """IoU of one (y1, x1, y2, x2) box with an array [G, 4] of them (intersection clamped at 0)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    box = np.asarray(box, dtype=np.float64)
    ih = np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0])
    iw = np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1])
    inter = np.clip(ih, 0.0, None) * np.clip(iw, 0.0, None)
    union = (box[2] - box[0]) * (box[3] - box[1]) + (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.where(union > 0.0, inter / union, 0.0)
    return iou


def average_precision(recall, precision, use_07_metric=False):
    """Area under a precision/recall curve given as cumulative arrays in detection order."""
    recall = np.asarray(recall, dtype=np.float64)
    precision = np.asarray(precision, dtype=np.float64)
    if use_07_metric:                                         # mean of the best precision at recall >= 0, 0.1, ..., 1
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            reached = recall >= t - 1e-12
            ap += (precision[reached].max() if reached.any() else 0.0) / 11.0
        return float(ap)
    mrec = np.concatenate([[0.0], recall, [1.0]])
    mpre = np.concatenate([[0.0], precision, [0.0]])
    for i in range(mpre.size - 2, -1, -1):                    # the envelope: best precision at this recall or beyond
        mpre[i] = max(mpre[i], mpre[i + 1])
    step = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def class_average_precision(detections, ground_truth, class_id, iou_threshold=0.5, use_07_metric=False):
    """AP of one class.  detections: list over images of (scores, bbox, class_id); ground_truth: list
    over images of (boxes [G, 4], classes [G]).  Returns (ap, recall, precision, number of ground-truth
    boxes); ap is None when the class has no ground truth."""
    gt_boxes, claimed, total = [], [], 0
    for boxes, classes in ground_truth:
        sel = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)[np.asarray(classes).reshape(-1) == class_id]
        gt_boxes.append(sel)
        claimed.append(np.zeros(sel.shape[0], dtype=bool))
        total += sel.shape[0]
    recs = []
    for img, (scores, bbox, cls) in enumerate(detections):
        scores, bbox, cls = np.asarray(scores).reshape(-1), np.asarray(bbox).reshape(-1, 4), np.asarray(cls).reshape(-1)
        for k in np.nonzero(cls == class_id)[0]:
            recs.append((-float(scores[k]), img, int(k)))
    recs.sort()
    tp = np.zeros(len(recs))
    for i, (_, img, k) in enumerate(recs):
        cand = gt_boxes[img]
        if cand.shape[0] == 0:
            continue
        iou = box_iou(detections[img][1][k], cand)
        j = int(np.argmax(iou))
        if iou[j] >= iou_threshold and not claimed[img][j]:
            claimed[img][j] = True
            tp[i] = 1.0
    if total == 0:
        return None, np.zeros(0), np.zeros(0), 0
    ctp = np.cumsum(tp)
    cfp = np.cumsum(1.0 - tp)
    recall = ctp / total
    precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
    return average_precision(recall, precision, use_07_metric), recall, precision, total


def mean_average_precision(detections, ground_truth_blocks, num_classes, iou_threshold=0.5, use_07_metric=False):
    """mAP over the classes that have ground truth.  detections: list over images of test_one_image's
    [scores, bbox, class_id]; ground_truth_blocks: [B, pad, 5] wire-format blocks of the same images.
    Returns dict(map, ap: {class_id: ap}, num_gt: {class_id: count})."""
    if len(detections) != len(ground_truth_blocks):
        raise ValueError('%d images of detections but %d of ground truth' % (len(detections), len(ground_truth_blocks)))
    gts = [decode_ground_truth(block) for block in ground_truth_blocks]
    aps, counts = {}, {}
    for c in range(int(num_classes)):
        ap, _, _, n = class_average_precision(detections, gts, c, iou_threshold, use_07_metric)
        if ap is not None:
            aps[c], counts[c] = ap, n
    return dict(map=float(np.mean(list(aps.values()))) if aps else float('nan'), ap=aps, num_gt=counts)
