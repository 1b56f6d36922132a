"""CPU: the two restatements of the reference (NumPy, C) against each other and
against the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

import yolo_tf_amd as y
from oracle import yolo_head_oracle as O
from tests import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def golden_heads(g):
    return [g['head%d' % s] for s in range(g['grids'].shape[0])]


@pytest.mark.parametrize('name', ['detect_v2_416_voc', 'detect_v3_160'])
def test_detect_golden(c_oracle, name):
    g = load(name)
    heads = golden_heads(g)
    C = int(g['num_classes']); thr = float(g['score_threshold']); iou = float(g['iou_threshold']); mb = int(g['max_boxes'])
    priors = list(g['priors'])
    # NumPy restatement reproduces the fixture exactly
    off = 0
    for b in range(heads[0].shape[0]):
        bb, cf = O.decode([h[b] for h in heads], priors, g['box_mult'], C)
        assert np.array_equal(bb, g['bbox'][b]) and np.array_equal(cf, g['conf'][b])
        s, bx, cl, ix = O.nms_per_class(bb, cf, thr, mb, iou)
        k = int(g['det_count'][b])
        assert np.array_equal(ix, g['det_index'][off:off + k]) and np.array_equal(cl, g['det_class'][off:off + k])
        assert np.array_equal(s, g['det_scores'][off:off + k])
        off += k
    # C restatement: same kept rows, values within 1e-5
    cfg = c_oracle.make_config([tuple(x) for x in g['grids']], priors, g['box_mult'], [1.0] * len(priors), C)
    got = c_oracle.detect(cfg, heads, thr, iou, mb)
    off = 0
    for b, (s, bx, cl, ix) in enumerate(got):
        k = int(g['det_count'][b])
        assert np.array_equal(ix, g['det_index'][off:off + k]) and np.array_equal(cl, g['det_class'][off:off + k])
        np.testing.assert_allclose(s, g['det_scores'][off:off + k], rtol=1e-5)
        np.testing.assert_allclose(bx, g['det_bbox'][off:off + k], rtol=1e-5, atol=1e-3)
        off += k


@pytest.mark.parametrize('name', ['loss_v3_160', 'loss_v2_416_voc'])
def test_loss_golden(c_oracle, name):
    g = load(name)
    heads = golden_heads(g)
    C = int(g['num_classes']); priors = list(g['priors']); v = int(g['version'])
    if v == 3:
        per, mean = O.loss_v3_batch(heads, g['ground_truth'], priors, C)
    else:
        per, mean = O.loss_v2_batch(heads[0], g['ground_truth'], priors[0], C)
    assert np.array_equal(per, g['loss_per_image']) and mean == g['loss_mean']
    cfg = c_oracle.make_config([tuple(x) for x in g['grids']], priors, [1.0] * len(priors), g['gt_rate'], C)
    cper, cparts, ciou, cmean = c_oracle.loss(cfg, heads, g['ground_truth'], g['loss_scales'],
                                              float(g['ignore_threshold']), v == 3, want_iou=True)
    np.testing.assert_allclose(cper, g['loss_per_image'], rtol=1e-5)
    np.testing.assert_allclose(cparts, g['loss_parts'], rtol=1e-5)
    np.testing.assert_allclose(cmean, g['loss_mean'], rtol=1e-5)
    # IoU is + - * / min max only: both restatements give the same bits
    assert np.array_equal(ciou.view(np.uint32), g['iou_max'].view(np.uint32))


def test_restatements_agree_on_v3_416(c_oracle):
    case = cases.v3()
    heads = case.heads(7, 2)
    cfg = case.oracle_cfg(c_oracle)
    cb, cc = c_oracle.decode(cfg, heads)
    for b in range(2):
        nb, nc = O.decode([h[b] for h in heads], case.priors, case.box_mult, case.C)
        np.testing.assert_allclose(cb[b], nb, rtol=1e-5, atol=1e-5 * 416)
        np.testing.assert_allclose(cc[b], nc, rtol=1e-5, atol=1e-30)
    got = c_oracle.detect(cfg, heads, 0.5, 0.5, 10)
    for b in range(2):
        s, bx, cl, ix = O.detect([h[b] for h in heads], case.priors, case.box_mult, case.C, 0.5, 10, 0.5)
        assert np.array_equal(ix, got[b][3]) and np.array_equal(cl, got[b][2]) and len(ix) > 0
    gt = case.gt(8, 2, 100, 101)
    per, parts, iou, mean = c_oracle.loss(cfg, heads, gt, (1., 1., 5., 1.), 0.5, True, want_iou=True)
    nper, nmean = O.loss_v3_batch(heads, gt, case.priors, case.C)
    np.testing.assert_allclose(per, nper, rtol=1e-5)
    np.testing.assert_allclose(mean, nmean, rtol=1e-5)
    ni = O.noobj_iou_max([h[0] for h in heads], gt[0], case.priors, case.gt_rate, case.C)
    assert np.array_equal(ni.view(np.uint32), iou[0].view(np.uint32))


def test_restatements_agree_on_v2_and_c_threads(c_oracle):
    case = cases.v2(size=480)
    heads = case.heads(9, 4)
    gt = case.gt(10, 4, 9, 60)
    cfg = case.oracle_cfg(c_oracle)
    per, _, _, mean = c_oracle.loss(cfg, heads, gt, (1., 1., 5., 1.), 0.6, False)
    per4, _, _, _ = c_oracle.loss(cfg, heads, gt, (1., 1., 5., 1.), 0.6, False, threads=4)
    assert np.array_equal(per, per4)
    nper, nmean = O.loss_v2_batch(heads[0], gt, case.priors[0], case.C)
    np.testing.assert_allclose(per, nper, rtol=1e-5)
    a = c_oracle.detect(cfg, heads, 0.4, 0.5, 10)
    b = c_oracle.detect(cfg, heads, 0.4, 0.5, 10, threads=4)
    cases.same_detections(a, b)


def test_nms_semantics():
    """Hand-made boxes: strict '>' on the IoU threshold, suppression only by
    SELECTED boxes, max_output_size, corner normalisation, zero-area boxes,
    lower index first among equal scores."""
    b = np.array([[0, 0, 10, 10],      # 0: best
                  [0, 0, 10, 5],       # 1: IoU with 0 = 0.5 exactly -> not suppressed at thr 0.5
                  [0, 1, 10, 10],      # 2: IoU with 0 = 0.9 -> suppressed
                  [10, 0.5, 0, 0],     # 3: flipped corners; IoU with 0 = 0.05; with 1 = 0.1
                  [20, 20, 20, 30],    # 4: zero area: never suppressed
                  [0, 1, 10, 10]], np.float32)   # 5: same as 2 -> suppressed by 0
    s = np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.7], np.float32)
    sel = O.non_max_suppression(b, s, 10, 0.5)
    assert sel.tolist() == [0, 1, 3, 4]
    assert O.non_max_suppression(b, s, 2, 0.5).tolist() == [0, 1]
    # 2 and 5 tie: with 0 removed they are taken in index order, 5 suppressed by 2
    assert O.non_max_suppression(b[1:], s[1:], 10, 0.5).tolist() == [0, 1, 2, 3]
    # a suppressed box does not suppress: chain a-b-c with IoU(a,b), IoU(b,c) high, IoU(a,c) low
    c = np.array([[0, 0, 10, 10], [0, 4, 10, 14], [0, 8, 10, 18]], np.float32)
    assert O.non_max_suppression(c, np.array([.9, .8, .7], np.float32), 10, 0.3).tolist() == [0, 2]
    with pytest.raises(ValueError):
        O.non_max_suppression(b, s, 10, 1.5)


def test_loss_formula_pieces():
    x = np.array([-3., 0., 2.5], np.float32); z = np.array([0., 1., 0.3], np.float32)
    want = -(z * np.log(O.sigmoid(x).astype(np.float64)) + (1 - z) * np.log(1 - O.sigmoid(x).astype(np.float64)))
    np.testing.assert_allclose(O.sigmoid_ce(x, z), want, rtol=1e-5)
    import torch
    t = torch.nn.functional.binary_cross_entropy_with_logits(torch.from_numpy(x), torch.from_numpy(z), reduction='none')
    np.testing.assert_allclose(O.sigmoid_ce(x, z), t.numpy(), rtol=1e-6)
    assert O.one_hot(np.array([1, 7, -1]), 3).tolist() == [[0, 1, 0], [0, 0, 0], [0, 0, 0]]
    gt = np.array([[5, 6, 7, 8, 2], [9, 9, 9, 9, 1], [-1, -1, -1, -1, -1], [-1, -1, -1, -1, -1]], np.float32)
    yx, hw, lab = O.normalized_gn(gt, 2.0)
    assert yx.shape == (2, 2) and lab.tolist() == [2, 1] and hw[0].tolist() == [3.5, 4.0]
    # priors: centre + 0.5, (y, x) order
    a_yx, a_hw, a1, a2 = O.get_priors(2, 3, [[1., 2.]])
    assert a_yx[1, 2, 0].tolist() == [1.5, 2.5] and a1[0, 0, 0].tolist() == [0.0, -0.5]


def test_v3_quirks():
    """Scale pairing and multipliers the reference really uses."""
    p = O.v3_grid_priors(y.synth.V3_PRIORS)
    np.testing.assert_allclose(p[0][0], [10 / 8., 13 / 8.])          # coarse grid <- small priors / 8
    np.testing.assert_allclose(p[2][2], [373 / 32., 326 / 32.])
    assert O.V3_BOX_MULT == (32.0, 32.0, 16.0) and O.V3_GT_RATE == (32.0, 16.0, 8.0)
    # additive hw decode: zero logits give hw = prior + 1
    h = [np.zeros((1, 1, 3 * 7), np.float32)] * 3
    bb, cf = O.decode(h, p, O.V3_BOX_MULT, 2)
    np.testing.assert_allclose(bb[0], [(1.0 - (10 / 8. + 1) / 2) * 32, (1.0 - (13 / 8. + 1) / 2) * 32,
                                       (1.0 + (10 / 8. + 1) / 2) * 32, (1.0 + (13 / 8. + 1) / 2) * 32], rtol=1e-6)
    assert np.all(cf == np.float32(0.25))


def _torch_loss_image(heads_t, gt_i, case, version, weights=(1., 1., 5., 1.)):
    """The loss of one image as a differentiable torch float64 expression.  The discrete
    decisions (GT selection per scale, best prior, ignore mask) come from the NumPy
    restatement; the expression itself is written with torch's own loss primitives."""
    import torch
    import torch.nn.functional as F
    C, K = case.C, case.C + 5
    w_coord, w_noobj, w_obj, w_class = weights
    heads_np = [h.detach().numpy().astype(np.float32) for h in heads_t]
    S = len(heads_t)
    rates = O.V3_GT_RATE if version == 3 else O.V2_GT_RATE
    thr = 0.5 if version == 3 else 0.6
    gn = [O.normalized_gn(gt_i, rates[s]) for s in range(S)]
    sp = [O._scale_positive_part(heads_np[s], case.priors[s], gn[s][0], gn[s][1], C) for s in range(S)]
    if version == 3:
        _, parts = O.loss_v3_image(heads_np, gt_i, case.priors, C, return_parts=True)
        masks, num_g = parts['masks'], parts['num_g']
    else:
        masks, num_g = [np.ones((gn[0][2].shape[0],), bool)], 1
    coord = cls = obj = noobj = 0.0
    for s in range(S):
        P = case.priors[s].shape[0]
        H, W = heads_np[s].shape[:2]
        hv = heads_t[s].reshape(H, W, P, K)
        _, full = O._scale_noobj(sp[s], thr, return_iou=True)
        with np.errstate(invalid='ignore'):
            w = torch.from_numpy((full <= np.float32(thr)).reshape(H, W, P).astype(np.float64))
        noobj = noobj + (F.softplus(hv[..., C + 4]) * w).sum()          # CE(x, 0) = softplus(x)
        g_yx, g_hw, g_label = gn[s]
        for j in np.nonzero(masks[s])[0]:
            cy, cx, p = int(np.floor(g_yx[j, 0])), int(np.floor(g_yx[j, 1])), int(sp[s]['index'][j])
            v = hv[cy, cx, p]
            onehot = torch.zeros(C, dtype=torch.float64); onehot[int(g_label[j])] = 1.0
            cls = cls + F.binary_cross_entropy_with_logits(v[:C], onehot, reduction='sum')
            frac = torch.from_numpy((g_yx[j] - np.floor(g_yx[j])).astype(np.float32).astype(np.float64))
            coord = coord + F.binary_cross_entropy_with_logits(v[C:C + 2], frac, reduction='sum')
            target = torch.from_numpy(np.log((g_hw[j] / case.priors[s][p]).astype(np.float32)).astype(np.float64))
            coord = coord + 0.5 * ((v[C + 2:C + 4] - target) ** 2).sum()
            obj = obj + F.softplus(-v[C + 4])                                # CE(x, 1) = softplus(-x)
    return ((w_coord * coord + w_class * cls + w_obj * obj) + w_noobj * noobj) / num_g


@pytest.mark.parametrize('version', [3, 2])
def test_gradient_restatement_against_torch_autograd(version):
    """Pins oracle.loss_v*_grad_image: torch autograd of an independent float64 expression of
    the same loss (whose value is first checked against the fp32 restatement)."""
    import torch
    case = cases.v3(size=160, C=7) if version == 3 else cases.v2(size=160, C=6)
    heads = case.heads(5, 1, obj_mean=-1.0)
    gt = case.gt(6, 1, 9, 12)
    gt[0, 3, :2] = gt[0, 2, :2]                              # two GT rows in one cell
    heads_i = [h[0] for h in heads]
    heads_t = [torch.from_numpy(h.astype(np.float64)).requires_grad_(True) for h in heads_i]
    loss = _torch_loss_image(heads_t, gt[0], case, version)
    if version == 3:
        want_loss = O.loss_v3_image(heads_i, gt[0], case.priors, case.C)
        got = O.loss_v3_grad_image(heads_i, gt[0], case.priors, case.C)
    else:
        want_loss = O.loss_v2_image(heads_i[0], gt[0], case.priors[0], case.C)
        got = [O.loss_v2_grad_image(heads_i[0], gt[0], case.priors[0], case.C)]
    assert abs(loss.item() - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    loss.backward()
    for g, t in zip(got, heads_t):
        ref = t.grad.numpy()
        assert np.count_nonzero(ref) > 0
        assert np.allclose(g, ref, rtol=1e-9, atol=1e-12)


def test_sigmoid_and_cross_entropy_against_torch():
    """A third witness for the recalled TensorFlow semantics that is importable here: torch's
    sigmoid and binary_cross_entropy_with_logits (the same documented formula,
    max(x, 0) - x * z + log(1 + exp(-|x|))) on CPU against the restatements, 1e-6.  It pins
    nothing of TensorFlow's, it catches a misreading shared by the two restatements."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.normal(0, 4, size=4000), [-100., -30., -1e-8, 0., 1e-8, 30., 100.]]).astype(np.float32)
    z = np.concatenate([rng.uniform(0, 1, size=2000), rng.integers(0, 2, size=2000), [0, 1, 0, 1, .5, 1, 0]]).astype(np.float32)
    want_s = torch.sigmoid(torch.from_numpy(x).double()).numpy()
    want_ce = F.binary_cross_entropy_with_logits(torch.from_numpy(x).double(), torch.from_numpy(z).double(),
                                                 reduction='none').numpy()
    np.testing.assert_allclose(O.sigmoid(x), want_s, rtol=1e-6, atol=1e-30)
    # fp32 evaluation of max(x, 0) - x * z + log1p(exp(-|x|)): each term rounds at the scale of |x|
    bound = 1e-6 * np.abs(want_ce) + 2.4e-7 * (1.0 + np.abs(x.astype(np.float64)))
    assert np.all(np.abs(O.sigmoid_ce(x, z).astype(np.float64) - want_ce) <= bound)
    f32 = F.binary_cross_entropy_with_logits(torch.from_numpy(x), torch.from_numpy(z), reduction='none').numpy()
    assert np.all(np.abs(O.sigmoid_ce(x, z).astype(np.float64) - f32) <= 2 * bound)


def test_restatements_agree_on_random_sizes_and_ground_truth(c_oracle):
    """The C restatement and the NumPy restatement, written independently from the reference's text, on random
    image sizes, class counts and ground truth (duplicated rows, rows sharing a cell, boxes of a few pixels and
    of the whole image): decode 1e-5, the same kept rows and classes, per-image losses 1e-5, ignore-mask IoU
    maxima bit for bit; v3 and v2."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=int(__import__('os').environ.get('YH_PROPERTY_EXAMPLES', '0')) or 40, deadline=None, derandomize='YH_PROPERTY_EXAMPLES' not in __import__('os').environ, suppress_health_check=list(HealthCheck))
    @given(version=st.sampled_from([2, 3]), cells=st.integers(1, 5), C=st.sampled_from([1, 3, 20, 80]),
           B=st.integers(1, 2), G=st.integers(1, 12), seed=st.integers(0, 10 ** 6), thr=st.sampled_from([0.1, 0.3, 0.5]))
    def check(version, cells, C, B, G, seed, thr):
        size = 32 * cells
        case = cases.v3(size=size, C=C) if version == 3 else cases.v2(size=size, C=C)
        heads = case.heads(seed, B, obj_mean=0.0, cold_mean=-2.0)
        cfg = case.oracle_cfg(c_oracle)
        cb, cc = c_oracle.decode(cfg, heads)
        got = c_oracle.detect(cfg, heads, thr, 0.5, 5)
        for b in range(B):
            nb, nc = O.decode([h[b] for h in heads], case.priors, case.box_mult, case.C)
            np.testing.assert_allclose(cb[b], nb, rtol=1e-5, atol=1e-5 * size)
            np.testing.assert_allclose(cc[b], nc, rtol=1e-5, atol=1e-30)
            # the walk on the C restatement's own decoded arrays: the same comparisons, so the same rows
            s, bx, cl, ix = O.nms_per_class(cb[b], cc[b], thr, 5, 0.5)
            assert np.array_equal(ix, got[b][3]) and np.array_equal(cl, got[b][2])
        rng = np.random.Generator(np.random.PCG64(seed + 1))
        gt = np.full((B, G + 2, 5), -1.0, np.float32)
        for b in range(B):
            yx = rng.uniform(0.0, size, size=(G, 2))
            hw = np.exp(rng.uniform(np.log(2.0), np.log(size), size=(G, 2)))
            for g in range(1, G):
                if rng.integers(0, 4) == 0:                  # a copy of an earlier row, or a row in its cell
                    j = int(rng.integers(0, g))
                    yx[g] = yx[j]
                    if rng.integers(0, 2):
                        hw[g] = hw[j]
            gt[b, :G, 0:2], gt[b, :G, 2:4] = yx, hw
            gt[b, :G, 4] = rng.integers(0, C, size=G)
        scales = (1., 1., 5., 1.)
        if version == 3:
            per, parts, iou, mean = c_oracle.loss(cfg, heads, gt, scales, 0.5, True, want_iou=True)
            nper, nmean = O.loss_v3_batch(heads, gt, case.priors, case.C)
            for b in range(B):
                ni = O.noobj_iou_max([h[b] for h in heads], gt[b], case.priors, case.gt_rate, case.C)
                assert np.array_equal(ni.view(np.uint32), iou[b].view(np.uint32))
        else:
            per, parts, iou, mean = c_oracle.loss(cfg, heads, gt, scales, 0.6, False)
            nper, nmean = O.loss_v2_batch(heads[0], gt, case.priors[0], case.C)
        np.testing.assert_allclose(per, nper, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mean, nmean, rtol=1e-5, atol=1e-6)
    check()
