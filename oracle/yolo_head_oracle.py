"""CPU restatement (NumPy, fp32) of the YOLO-TF detection-head path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module: it is the checker the HIP path is compared with, never the thing that
is measured or shipped.

PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
this path and its arithmetic runs inside TensorFlow 1.12, which is not
installed here.  This file therefore restates the reference's op sequence from
its source text, and is cross-checked only against a second, independent C
restatement (yolo_head_oracle.c).  The TensorFlow op semantics relied upon
(NonMaxSuppressionV3, sigmoid_cross_entropy_with_logits, argmax/argmin
first-index rule, boolean_mask order, one_hot out of range) are the published
ones, restated in the functions below.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  All arithmetic is float32, evaluated op by op in the
reference's order; boxes are (y1, x1, y2, x2).
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# TensorFlow op restatements
# --------------------------------------------------------------------------

def sigmoid(x):
    """tf.sigmoid on float32: 1 / (1 + exp(-x))."""
    x = np.asarray(x, dtype=F32)
    with np.errstate(over='ignore'):
        return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def sigmoid_ce(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits, the documented formula
    max(x, 0) - x * z + log(1 + exp(-abs(x)))."""
    x = np.asarray(logits, dtype=F32)
    z = np.asarray(labels, dtype=F32)
    return (np.maximum(x, F32(0)) - x * z + np.log1p(np.exp(-np.abs(x)))).astype(F32)


def one_hot(idx, depth):
    """tf.one_hot: an index outside [0, depth) gives an all-zero row."""
    idx = np.asarray(idx, dtype=np.int64)
    out = np.zeros((idx.shape[0], depth), dtype=F32)
    ok = (idx >= 0) & (idx < depth)
    out[np.nonzero(ok)[0], idx[ok]] = F32(1)
    return out


def _nms_suppresses(boxes, i, j, iou_threshold):
    """NonMaxSuppressionV3's overlap test (tensorflow/core/kernels/
    non_max_suppression_op.cc, TF 1.12): corner order is normalised with
    min/max, a box of non-positive area never suppresses nor is suppressed,
    the intersection is clamped at zero and the comparison is strict."""
    ymin_i = min(boxes[i, 0], boxes[i, 2]); xmin_i = min(boxes[i, 1], boxes[i, 3])
    ymax_i = max(boxes[i, 0], boxes[i, 2]); xmax_i = max(boxes[i, 1], boxes[i, 3])
    ymin_j = min(boxes[j, 0], boxes[j, 2]); xmin_j = min(boxes[j, 1], boxes[j, 3])
    ymax_j = max(boxes[j, 0], boxes[j, 2]); xmax_j = max(boxes[j, 1], boxes[j, 3])
    area_i = F32(F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i))
    area_j = F32(F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return False
    iy = max(F32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), F32(0))
    ix = max(F32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), F32(0))
    inter = F32(iy * ix)
    with np.errstate(invalid='ignore', divide='ignore'):
        iou = F32(inter / F32(F32(area_i + area_j) - inter))
    return bool(iou > F32(iou_threshold))


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression(boxes, scores, max_output_size,
    iou_threshold) as called at YOLOv3.py:358-361 / YOLOv2.py:200-203.

    Greedy: candidates in descending score; a candidate is selected unless it
    overlaps an already selected box by more than iou_threshold; stops at
    max_output_size.  Equal scores: lower input index first (TF 1.12 leaves
    the order of ties to its heap; this build declares lower-index-first).
    Returns int32 indices into the input list, in selection order."""
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F32).reshape(-1)
    if not (0.0 <= iou_threshold <= 1.0):
        raise ValueError('iou_threshold must be in [0, 1]')
    order = np.argsort(-scores, kind='stable')
    out_size = min(int(max_output_size), boxes.shape[0])
    selected = []
    for cand in order:
        if len(selected) >= out_size:
            break
        if not (scores[cand] > -np.inf):      # score_threshold = -inf, strict
            continue
        keep = True
        for j in reversed(selected):
            if _nms_suppresses(boxes, cand, j, iou_threshold):
                keep = False
                break
        if keep:
            selected.append(int(cand))
    return np.asarray(selected, dtype=np.int32)


# --------------------------------------------------------------------------
# head split and priors
# --------------------------------------------------------------------------

def split_head(pred, num_priors, num_classes):
    """YOLOv3.py:92-110 / YOLOv2.py:94-100: [..,H,W,P*(C+5)] -> [..,H,W,P,C+5]
    and the four channel slices class[:C], yx[C:C+2], hw[C+2:C+4], obj[C+4:]."""
    C = num_classes
    pred = np.asarray(pred, dtype=F32)
    pred = pred.reshape(pred.shape[:-1] + (num_priors, -1))
    return pred[..., :C], pred[..., C:C + 2], pred[..., C + 2:C + 4], pred[..., C + 4:]


def get_priors(H, W, priors):
    """_get_priors (YOLOv3.py:419-433, YOLOv2.py:230-244).  priors: [P,2]
    (h, w) in grid units.  Returns abbox_yx, abbox_hw, abbox_y1x1, abbox_y2x2,
    each [H,W,P,2]."""
    priors = np.asarray(priors, dtype=F32).reshape(1, 1, -1, 2)
    P = priors.shape[2]
    tl_y = np.arange(H, dtype=F32).reshape(-1, 1, 1, 1)
    tl_x = np.arange(W, dtype=F32).reshape(1, -1, 1, 1)
    tl_y_ = np.tile(tl_y, [1, W, 1, 1])
    tl_x_ = np.tile(tl_x, [H, 1, 1, 1])
    tl = np.concatenate([tl_y_, tl_x_], -1)
    abbox_yx = (tl + F32(0.5)).astype(F32)
    abbox_yx = np.tile(abbox_yx, [1, 1, P, 1])
    abbox_hw = np.tile(priors, [H, W, 1, 1]).astype(F32)
    abbox_y1x1 = (abbox_yx - abbox_hw / F32(2)).astype(F32)
    abbox_y2x2 = (abbox_yx + abbox_hw / F32(2)).astype(F32)
    return abbox_yx, abbox_hw, abbox_y1x1, abbox_y2x2


def v3_grid_priors(config_priors):
    """YOLOv3.py:37-41: priors[i] / stride[i] with stride = [8, 16, 32]; list
    entry 0 is the one paired with the first (coarsest) grid at
    YOLOv3.py:111-113."""
    stride = [F32(8.), F32(16.), F32(32.)]
    return [(np.asarray(config_priors[i], dtype=F32) / stride[i]).astype(F32)
            for i in range(len(config_priors))]


V3_BOX_MULT = (32.0, 32.0, 16.0)   # YOLOv3.py:346-348: stride[-1], stride[-1], stride[-2]
V3_GT_RATE = (32.0, 16.0, 8.0)     # YOLOv3.py:118-120: stride[-1], stride[-2], stride[-3]
V2_BOX_MULT = (32.0,)              # YOLOv2.py:191, 302
V2_GT_RATE = (32.0,)               # YOLOv2.py:106, 302


# --------------------------------------------------------------------------
# decode + per-class NMS (test mode)
# --------------------------------------------------------------------------

def decode(heads, priors, box_mult, num_classes):
    """YOLOv3.py:320-350 (three scales) / YOLOv2.py:183-192 (one scale) for one
    image.  heads: list of [H,W,P*(C+5)]; priors: list of [P,2] grid units;
    box_mult: per-scale multiplier.  Returns bbox [N,4] and confidence [N,C];
    row order = scales concatenated, (y*W + x)*P + p within a scale."""
    C = num_classes
    cls_l, obj_l, box_l = [], [], []
    for pred, pr, m in zip(heads, priors, box_mult):
        pr = np.asarray(pr, dtype=F32)
        P = pr.shape[0]
        H, W = pred.shape[0], pred.shape[1]
        pclass, pyx, phw, pobj = split_head(pred, P, C)
        a_yx, a_hw, _, _ = get_priors(H, W, pr)
        pclass = pclass.reshape(-1, C); pobj = pobj.reshape(-1, 1)
        pyx = pyx.reshape(-1, 2); phw = phw.reshape(-1, 2)
        a_yx = a_yx.reshape(-1, 2); a_hw = a_hw.reshape(-1, 2)
        bbox_yx = (a_yx + sigmoid(pyx)).astype(F32)
        with np.errstate(over='ignore'):
            bbox_hw = (a_hw + np.exp(phw)).astype(F32)
        box = np.concatenate([bbox_yx - bbox_hw / F32(2), bbox_yx + bbox_hw / F32(2)],
                             axis=-1).astype(F32) * F32(m)
        cls_l.append(pclass); obj_l.append(pobj); box_l.append(box.astype(F32))
    pclasst = sigmoid(np.concatenate(cls_l, axis=0))
    pobjt = sigmoid(np.concatenate(obj_l, axis=0))
    bbox = np.concatenate(box_l, axis=0).astype(F32)
    confidence = (pclasst * pobjt).astype(F32)
    return bbox, confidence


def nms_per_class(bbox, confidence, score_threshold, max_boxes, iou_threshold):
    """YOLOv3.py:351-368 / YOLOv2.py:193-210 on an already decoded image.
    Returns scores [K], bbox [K,4], class_id [K] int32 and, in addition to the
    reference's outputs, the global box index [K] int32 of every detection."""
    bbox = np.asarray(bbox, dtype=F32); confidence = np.asarray(confidence, dtype=F32)
    C = confidence.shape[1]
    filter_mask = confidence >= F32(score_threshold)
    scores, boxes, class_id, index = [], [], [], []
    for i in range(C):
        rows = np.nonzero(filter_mask[:, i])[0]            # boolean_mask keeps order
        scoresi = confidence[rows, i]
        bboxi = bbox[rows]
        sel = non_max_suppression(bboxi, scoresi, max_boxes, iou_threshold)
        scores.append(scoresi[sel]); boxes.append(bboxi[sel].reshape(-1, 4))
        class_id.append(np.full(sel.shape, i, dtype=np.int32))
        index.append(rows[sel].astype(np.int32))
    return (np.concatenate(scores).astype(F32), np.concatenate(boxes, axis=0).astype(F32),
            np.concatenate(class_id).astype(np.int32), np.concatenate(index).astype(np.int32))


def detect(heads, priors, box_mult, num_classes, score_threshold, max_boxes, iou_threshold):
    """test_one_image's graph for one image: decode then per-class NMS."""
    bbox, conf = decode(heads, priors, box_mult, num_classes)
    return nms_per_class(bbox, conf, score_threshold, max_boxes, iou_threshold)


# --------------------------------------------------------------------------
# training loss
# --------------------------------------------------------------------------

def normalized_gn(ground_truth_i, downsampling_rate):
    """_get_normlized_gn (YOLOv3.py:435-442, YOLOv2.py:246-253) for one image's
    [pad,5] rows (yc, xc, h, w, class), padded with -1 rows."""
    gt = np.asarray(ground_truth_i, dtype=F32)
    slice_index = int(np.argmin(gt, axis=0)[0])            # first index of the minimum of column 0
    gt = gt[:slice_index]
    r = F32(downsampling_rate)
    scale = np.array([r, r, r, r, 1], dtype=F32).reshape(1, 5)
    gn = (gt / scale).astype(F32)
    return gn[..., :2], gn[..., 2:4], gn[..., 4].astype(np.int32)


def _gather_nd(t, idx):
    """tf.gather_nd(t[H,W,...], idx[G,2])."""
    return t[idx[:, 0], idx[:, 1]]


def _scale_positive_part(pred_i, priors, gn_yx, gn_hw, num_classes):
    """Per scale, YOLOv3.py:123-184 / YOLOv2.py:107-129: gathers at the GT
    cells and the GT x cell-prior IoU (intersection not clamped)."""
    C = num_classes
    priors = np.asarray(priors, dtype=F32)
    P = priors.shape[0]
    H, W = pred_i.shape[0], pred_i.shape[1]
    pclass, pyx, phw, pobj = split_head(pred_i, P, C)
    a_yx, a_hw, a_y1x1, a_y2x2 = get_priors(H, W, priors)
    gn_floor = np.floor(gn_yx).astype(np.int64)
    nogn_mask = np.ones((H, W), dtype=bool)
    nogn_mask[gn_floor[:, 0], gn_floor[:, 1]] = False       # sparse_to_dense, then (1 - x) > 0
    rp_yx = _gather_nd(pyx, gn_floor); rp_hw = _gather_nd(phw, gn_floor)
    ra_hw = _gather_nd(a_hw, gn_floor)
    rp_class = _gather_nd(pclass, gn_floor); rp_obj = _gather_nd(pobj, gn_floor)
    ra_y1x1 = _gather_nd(a_y1x1, gn_floor); ra_y2x2 = _gather_nd(a_y2x2, gn_floor)
    gn_y1x1 = np.expand_dims(gn_yx - gn_hw / F32(2.), axis=1).astype(F32)
    gn_y2x2 = np.expand_dims(gn_yx + gn_hw / F32(2.), axis=1).astype(F32)
    iou_y1x1 = np.maximum(gn_y1x1, ra_y1x1)
    iou_y2x2 = np.minimum(gn_y2x2, ra_y2x2)
    inter = np.prod((iou_y2x2 - iou_y1x1).astype(F32), axis=-1).astype(F32)
    garea = np.prod((gn_y2x2 - gn_y1x1).astype(F32), axis=-1).astype(F32)
    aarea = np.prod(ra_hw, axis=-1).astype(F32)
    with np.errstate(invalid='ignore', divide='ignore'):
        iou = (inter / ((aarea + garea).astype(F32) - inter)).astype(F32)     # [G,P]
    if iou.shape[0]:
        index = np.argmax(iou, axis=-1).astype(np.int32)
        iou_max = np.max(iou, axis=-1).astype(F32)
    else:
        index = np.zeros((0,), np.int32); iou_max = np.zeros((0,), F32)
    return dict(P=P, H=H, W=W, pobj=pobj, a_yx=a_yx, a_hw=a_hw, nogn_mask=nogn_mask,
                rp_yx=rp_yx, rp_hw=rp_hw, ra_hw=ra_hw, rp_class=rp_class, rp_obj=rp_obj,
                gn_y1x1=gn_y1x1, gn_y2x2=gn_y2x2, iou=iou, index=index, iou_max=iou_max)


def _scale_losses(sp, mask, gn_yx, gn_hw, gn_label, num_classes):
    """YOLOv3.py:190-241 / YOLOv2.py:128-145 for one scale: select the GT rows
    of `mask`, take their best prior, build targets, sum the four loss terms."""
    C = num_classes
    idx = sp['index'][mask]
    rng = np.arange(idx.shape[0])
    rp_yx = sp['rp_yx'][mask][rng, idx].reshape(-1, 2)
    rp_hw = sp['rp_hw'][mask][rng, idx].reshape(-1, 2)
    rp_class = sp['rp_class'][mask][rng, idx].reshape(-1, C)
    rp_obj = sp['rp_obj'][mask][rng, idx].reshape(-1, 1)
    ra_hw = sp['ra_hw'][mask][rng, idx].reshape(-1, 2)
    g_yx = gn_yx[mask]; g_hw = gn_hw[mask]
    label = one_hot(gn_label[mask], C)
    yx_target = (g_yx - np.floor(g_yx)).astype(F32)
    with np.errstate(invalid='ignore', divide='ignore'):
        hw_target = np.log((g_hw / ra_hw).astype(F32)).astype(F32)
    yx_loss = np.sum(sigmoid_ce(rp_yx, yx_target), dtype=F32)
    hw_loss = F32(0.5) * np.sum(np.square((rp_hw - hw_target).astype(F32)), dtype=F32)
    class_loss = np.sum(sigmoid_ce(rp_class, label), dtype=F32)
    obj_loss = np.sum(sigmoid_ce(rp_obj, np.ones_like(rp_obj)), dtype=F32)
    return F32(yx_loss), F32(hw_loss), F32(class_loss), F32(obj_loss)


def _scale_noobj(sp, ignore_threshold, return_iou=False):
    """YOLOv3.py:242-307 / YOLOv2.py:146-171 for one scale.  The no-object box
    is built from L = a_yx - a_hw/2 and U = a_yx + a_hw/2 used as if they were
    (centre, size): y1x1 = L - U/2, y2x2 = L + U/2.  IoU against every GT of the
    image with the intersection not clamped; max over GT; weight = (max <= thr)."""
    P = sp['P']
    nogn = sp['nogn_mask'].reshape(-1)
    yx_nb = np.expand_dims((sp['a_yx'] - sp['a_hw'] / F32(2.)).astype(F32).reshape(-1, P, 2)[nogn], 1)
    hw_nb = np.expand_dims((sp['a_yx'] + sp['a_hw'] / F32(2.)).astype(F32).reshape(-1, P, 2)[nogn], 1)
    a_y1x1 = (yx_nb - hw_nb / F32(2.)).astype(F32)          # [A,1,P,2]
    a_y2x2 = (yx_nb + hw_nb / F32(2.)).astype(F32)
    pobj_nb = sp['pobj'].reshape(-1, P)[nogn]                # [A,P]
    g1 = np.expand_dims(sp['gn_y1x1'], 0)                    # [1,G,1,2]
    g2 = np.expand_dims(sp['gn_y2x2'], 0)
    iou_y1x1 = np.maximum(g1, a_y1x1)                        # broadcast == the reference's tiles
    iou_y2x2 = np.minimum(g2, a_y2x2)
    inter = np.prod((iou_y2x2 - iou_y1x1).astype(F32), axis=-1).astype(F32)   # [A,G,P]
    aarea = np.prod((a_y2x2 - a_y1x1).astype(F32), axis=-1).astype(F32)       # [A,1,P]
    garea = np.prod((g2 - g1).astype(F32), axis=-1).astype(F32)               # [1,G,1]
    with np.errstate(invalid='ignore', divide='ignore'):
        iou = (inter / ((aarea + garea).astype(F32) - inter)).astype(F32)
    if iou.shape[1] == 0:                                    # reduce_max over no GT: -inf
        iou = np.full((iou.shape[0], P), -np.inf, dtype=F32)
    else:
        iou = np.max(iou, axis=1).astype(F32)                # [A,P]; NaN propagates -> weight 0
    with np.errstate(invalid='ignore'):
        w = (iou <= F32(ignore_threshold)).astype(F32)
    loss = np.sum((sigmoid_ce(pobj_nb, np.zeros_like(pobj_nb)) * w).astype(F32), dtype=F32)
    if return_iou:
        full = np.full((nogn.shape[0], P), np.nan, dtype=F32)
        full[nogn] = iou
        return F32(loss), full
    return F32(loss)


def loss_v3_image(heads_i, ground_truth_i, priors, num_classes,
                  coord_scale=1., noobj_scale=1., obj_scale=5., class_scale=1.,
                  return_parts=False):
    """YOLOv3.py:117-310 for one image.  heads_i: three [H,W,P*(C+5)] arrays in
    the reference's order (pred1 coarsest); priors: three [P,2] arrays in grid
    units, priors[0] paired with heads_i[0]."""
    C = num_classes
    gn, sp = [], []
    for s in range(3):
        gn.append(normalized_gn(ground_truth_i, V3_GT_RATE[s]))
    num_g = gn[1][2].shape[0]
    with np.errstate(invalid='ignore', divide='ignore'):
        num_gf = F32(num_g)
    for s in range(3):
        sp.append(_scale_positive_part(heads_i[s], priors[s], gn[s][0], gn[s][1], C))
    m1f = (sp[0]['iou_max'] > sp[1]['iou_max']).astype(F32) * (sp[0]['iou_max'] > sp[2]['iou_max']).astype(F32)
    m2f = (sp[1]['iou_max'] > sp[0]['iou_max']).astype(F32) * (sp[1]['iou_max'] > sp[2]['iou_max']).astype(F32)
    m3 = (F32(1.) - (m1f + m2f)) > 0.
    masks = [m1f > 0., m2f > 0., m3]
    yx = hw = cl = ob = F32(0)
    yx_l, hw_l, cl_l, ob_l = [], [], [], []
    for s in range(3):
        a, b, c, d = _scale_losses(sp[s], masks[s], gn[s][0], gn[s][1], gn[s][2], C)
        yx_l.append(a); hw_l.append(b); cl_l.append(c); ob_l.append(d)
    coord_loss = F32(F32(F32(F32(F32(yx_l[0] + yx_l[1]) + yx_l[2]) + hw_l[0]) + hw_l[1]) + hw_l[2])
    class_loss = F32(F32(cl_l[0] + cl_l[1]) + cl_l[2])
    obj_loss = F32(F32(ob_l[0] + ob_l[1]) + ob_l[2])
    nb = [_scale_noobj(sp[s], 0.5) for s in range(3)]
    noobj_loss = F32(F32(nb[0] + nb[1]) + nb[2])
    with np.errstate(invalid='ignore', divide='ignore'):
        pos_loss = F32(F32(F32(F32(coord_scale) * coord_loss + F32(class_scale) * class_loss)
                           + F32(obj_scale) * obj_loss) / num_gf)
        neg_loss = F32(F32(F32(noobj_scale) * noobj_loss) / num_gf)
        total = F32(pos_loss + neg_loss)
    if return_parts:
        return total, dict(coord=coord_loss, cls=class_loss, obj=obj_loss, noobj=noobj_loss,
                           num_g=num_g, masks=masks, index=[p['index'] for p in sp],
                           iou=[p['iou'] for p in sp])
    return total


def loss_v2_image(head_i, ground_truth_i, priors, num_classes,
                  coord_scale=1., noobj_scale=1., obj_scale=5., class_scale=1.,
                  return_parts=False):
    """YOLOv2.py:105-173 for one image: one scale, rate 32, every GT takes its
    best prior, ignore threshold 0.6, no division by the GT count."""
    C = num_classes
    g_yx, g_hw, g_label = normalized_gn(ground_truth_i, V2_GT_RATE[0])
    sp = _scale_positive_part(head_i, priors, g_yx, g_hw, C)
    allm = np.ones((g_label.shape[0],), dtype=bool)
    yx_loss, hw_loss, class_loss, obj_loss = _scale_losses(sp, allm, g_yx, g_hw, g_label, C)
    coord_loss = F32(yx_loss + hw_loss)
    noobj_loss = _scale_noobj(sp, 0.6)
    loss = F32(F32(F32(F32(coord_scale) * coord_loss + F32(class_scale) * class_loss)
                   + F32(obj_scale) * obj_loss) + F32(noobj_scale) * noobj_loss)
    if return_parts:
        return loss, dict(coord=coord_loss, cls=class_loss, obj=obj_loss, noobj=noobj_loss,
                          num_g=g_label.shape[0], index=[sp['index']], iou=[sp['iou']])
    return loss


def loss_v3_batch(heads, ground_truth, priors, num_classes, **kw):
    """YOLOv3.py:116-311: per-image losses and their reduce_mean.
    heads: three [B,H,W,P*(C+5)] arrays."""
    B = heads[0].shape[0]
    per = np.array([loss_v3_image([h[i] for h in heads], ground_truth[i], priors, num_classes, **kw)
                    for i in range(B)], dtype=F32)
    return per, F32(np.mean(per, dtype=F32))


def loss_v2_batch(head, ground_truth, priors, num_classes, **kw):
    """YOLOv2.py:104-174."""
    B = head.shape[0]
    per = np.array([loss_v2_image(head[i], ground_truth[i], priors, num_classes, **kw)
                    for i in range(B)], dtype=F32)
    return per, F32(np.mean(per, dtype=F32))


def noobj_iou_max(heads_i, ground_truth_i, priors, gt_rate, num_classes):
    """The ignore-mask IoU maximum of every box of one image (NaN inside cells
    that hold a GT centre, which the reference removes before this step).
    Returned concatenated over scales in box order, for IoU parity checks."""
    out = []
    for s, (h, pr) in enumerate(zip(heads_i, priors)):
        g_yx, g_hw, _ = normalized_gn(ground_truth_i, gt_rate[s])
        sp = _scale_positive_part(h, pr, g_yx, g_hw, num_classes)
        _, full = _scale_noobj(sp, 0.5, return_iou=True)
        out.append(full.reshape(-1))
    return np.concatenate(out).astype(F32)


# --------------------------------------------------------------------------
# gradient of the loss w.r.t. the head tensors
# --------------------------------------------------------------------------

def _loss_grad_scales(heads_i, gn, sp, masks, priors, num_classes, weights, inv_g, ignore_threshold):
    """Closed-form d(loss_i)/d(head) for one image, the derivative TF's autodiff takes
    of YOLOv3.py:190-310 / YOLOv2.py:128-172 (optimizer.minimize, YOLOv3.py:312-318):
    the GT selection, the best-prior index and the ignore mask do not depend on the
    head, so only the cross-entropy and squared-error terms differentiate:
    d CE(x, z)/dx = sigmoid(x) - z, d 0.5 (t - u)^2/dt = t - u.  float64 arithmetic on
    the fp32 inputs.  Returns one [H, W, P*(C+5)] float64 array per scale."""
    C = num_classes
    K = C + 5
    w_coord, w_noobj, w_obj, w_class = [float(w) for w in weights]
    sig = lambda x: 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
    grads = []
    for s, h in enumerate(heads_i):
        pr = np.asarray(priors[s], F32)
        P = pr.shape[0]
        H, W = h.shape[0], h.shape[1]
        hv = np.asarray(h, np.float64).reshape(H, W, P, K)
        g = np.zeros((H, W, P, K), np.float64)
        # no-object term (YOLOv3.py:242-307): boxes outside GT cells whose IoU maximum <= threshold
        _, full = _scale_noobj(sp[s], ignore_threshold, return_iou=True)
        with np.errstate(invalid='ignore'):
            wmask = (full <= F32(ignore_threshold)).reshape(H, W, P)
        g[..., C + 4] = np.where(wmask, w_noobj * inv_g * sig(hv[..., C + 4]), 0.0)
        # assigned rows (YOLOv3.py:190-241)
        g_yx, g_hw, g_label = gn[s]
        for j in np.nonzero(masks[s])[0]:
            cy, cx = int(np.floor(g_yx[j, 0])), int(np.floor(g_yx[j, 1]))
            p = int(sp[s]['index'][j])
            v = hv[cy, cx, p]
            d = np.zeros((K,), np.float64)
            onehot = np.zeros((C,), np.float64); onehot[int(g_label[j])] = 1.0
            d[:C] = w_class * (sig(v[:C]) - onehot)
            frac = (g_yx[j] - np.floor(g_yx[j])).astype(F32).astype(np.float64)
            d[C:C + 2] = w_coord * (sig(v[C:C + 2]) - frac)
            with np.errstate(invalid='ignore', divide='ignore'):
                target = np.log((g_hw[j] / pr[p]).astype(F32)).astype(np.float64)
            d[C + 2:C + 4] = w_coord * (v[C + 2:C + 4] - target)
            d[C + 4] = w_obj * (sig(v[C + 4]) - 1.0)
            g[cy, cx, p] += d * inv_g                         # several GT rows may share a box
        grads.append(g.reshape(H, W, P * K))
    return grads


def loss_v3_grad_image(heads_i, ground_truth_i, priors, num_classes,
                       coord_scale=1., noobj_scale=1., obj_scale=5., class_scale=1.):
    """d loss_v3_image / d heads_i."""
    _, parts = loss_v3_image(heads_i, ground_truth_i, priors, num_classes, coord_scale, noobj_scale,
                             obj_scale, class_scale, return_parts=True)
    gn = [normalized_gn(ground_truth_i, V3_GT_RATE[s]) for s in range(3)]
    sp = [_scale_positive_part(heads_i[s], priors[s], gn[s][0], gn[s][1], num_classes) for s in range(3)]
    inv_g = 1.0 / float(parts['num_g'])
    return _loss_grad_scales(heads_i, gn, sp, parts['masks'], priors, num_classes,
                             (coord_scale, noobj_scale, obj_scale, class_scale), inv_g, 0.5)


def loss_v2_grad_image(head_i, ground_truth_i, priors, num_classes,
                       coord_scale=1., noobj_scale=1., obj_scale=5., class_scale=1.):
    """d loss_v2_image / d head_i (one scale, every GT assigned, no division by the GT count)."""
    gn = [normalized_gn(ground_truth_i, V2_GT_RATE[0])]
    sp = [_scale_positive_part(head_i, priors, gn[0][0], gn[0][1], num_classes)]
    masks = [np.ones((gn[0][2].shape[0],), dtype=bool)]
    return _loss_grad_scales([head_i], gn, sp, masks, [priors], num_classes,
                             (coord_scale, noobj_scale, obj_scale, class_scale), 1.0, 0.6)[0]
