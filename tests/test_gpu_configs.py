"""The benchmark configurations at their own shapes (BASELINE.json configs 2, 4, 5), through the
C ABI: against the C oracle at batches the oracle finishes in seconds, and fused == staged
(full decode, then yh_nms) at the full per-GPU batch."""
import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from tests import cases
from tests.test_gpu_parity import to_dev, assert_close, end_to_end_against_oracle, RTOL

pytestmark = pytest.mark.gpu


def _end_to_end(case, heads, thr, iou, mb, gpu, c_oracle, **kw):
    return end_to_end_against_oracle(case, heads, thr, iou, mb, gpu, c_oracle, **kw)


def _nms_stage_bit_exact(case, dheads, thr, iou, mb, c_oracle, **kw):
    hd = case.head_obj()
    boxes, conf = hd.decode(dheads)
    res = hd.nms(boxes, conf, thr, iou, mb, want_class_candidates=True, **kw)
    got = res.per_image()
    hb, hc = boxes.cpu().numpy(), conf.cpu().numpy()
    want = c_oracle.nms(hb, hc, thr, iou, mb)
    cases.same_detections(got, want)
    for g, w in zip(got, want):
        assert np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32)), 'scores differ in bits'
        assert np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)), 'boxes differ in bits'
    cc = res.resolved().class_candidates.cpu().numpy()
    assert np.array_equal(cc, (hc >= np.float32(thr)).sum(axis=1).astype(np.int32))
    return got, cc


def _fused_equals_staged(case, heads, thr, iou, mb):
    hd = case.head_obj()
    fused = hd.detect(heads, thr, iou, mb, want_class_candidates=True).resolved()
    boxes, conf = hd.decode(heads)
    staged = hd.nms(boxes, conf, thr, iou, mb, want_class_candidates=True).resolved()
    f, s = fused.per_image(), staged.per_image()
    cases.same_detections(f, s)
    for a, b in zip(f, s):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert torch.equal(fused.class_candidates, staged.class_candidates)
    assert int(fused.count.sum()) > 0
    return fused


def test_config4_yolov3_608_against_the_oracle(gpu, c_oracle):
    """YOLOv3-608 COCO-80, 19/38/76 grids, 22743 boxes per image (YOLOv3.py:319-368)."""
    case = cases.v3(size=608)
    assert case.N == 22743
    heads = case.heads(608, 3)
    _end_to_end(case, heads, 0.5, 0.5, 10, gpu, c_oracle)
    got, _ = _nms_stage_bit_exact(case, to_dev(heads, gpu), 0.5, 0.5, 10, c_oracle)
    assert sum(len(g[0]) for g in got) > 0


def test_config4_yolov3_608_fused_equals_staged_at_batch_64(gpu):
    case = cases.v3(size=608)
    heads = y.synth.torch_heads(gpu, 6080, 64, case.grids, case.P, case.C)
    _fused_equals_staged(case, heads, 0.5, 0.5, 10)


def test_config2_yolov3_416_fused_equals_staged_at_batch_64(gpu):
    case = cases.v3()
    heads = y.synth.torch_heads(gpu, 4160, 64, case.grids, case.P, case.C)
    fused = _fused_equals_staged(case, heads, 0.5, 0.5, 10)
    # the software pipeline over three lanes gives the same batch results as the plain call
    lanes = y.DetectLanes(case.head_obj(), 0.5, 0.5, 10, lanes=3)
    for _ in range(4):
        lanes.submit(heads, wait_current=True)
    for d in lanes.flush():
        cases.same_detections(d.per_image(), fused.per_image())


def test_config5_nms_stress_at_full_size(gpu, c_oracle):
    """100 x 100 grid x 10 priors = 1e5 boxes per image, 80 classes, threshold 0.001, 100 boxes per
    class: about 1e5 candidates per image.  Candidate counts and kept rows bit-exact against the
    oracle's NMS on the GPU's own decoded inputs, and the fused path (from the head tensors) equal to
    that staged path."""
    case = cases.stress(grid=100)
    assert case.N == 100000
    B = 2
    heads = case.heads(5005, B, **y.synth.STRESS)
    dheads = to_dev(heads, gpu)
    got, cc = _nms_stage_bit_exact(case, dheads, 0.001, 0.5, 100, c_oracle, max_candidates=200000)
    assert 80000 < cc.sum(axis=1).min() and cc.sum(axis=1).max() < 200000
    assert min(len(g[0]) for g in got) > 5000                 # most classes select their 100 boxes
    fused = case.head_obj().detect(dheads, 0.001, 0.5, 100, max_candidates=200000, want_class_candidates=True)
    assert not fused.overflowed()
    cases.same_detections(fused.per_image(), got)
    assert np.array_equal(fused.class_candidates.cpu().numpy(), cc)
    # the default capacity overflows, is reported, and the rerun grows it to the same answer
    auto = case.head_obj().detect(dheads, 0.001, 0.5, 100)
    assert auto.overflowed()
    cases.same_detections(auto.per_image(), got)


def test_config5_fused_equals_staged_at_batch_8(gpu):
    case = cases.stress(grid=100)
    heads = y.synth.torch_heads(gpu, 5050, 8, case.grids, case.P, case.C, **y.synth.STRESS)
    hd = case.head_obj()
    fused = hd.detect(heads, 0.001, 0.5, 100, max_candidates=200000, want_class_candidates=True)
    boxes, conf = hd.decode(heads)
    staged = hd.nms(boxes, conf, 0.001, 0.5, 100, max_candidates=200000, want_class_candidates=True)
    assert not fused.overflowed() and not staged.overflowed()
    cases.same_detections(fused.per_image(), staged.per_image())
    assert torch.equal(fused.class_candidates, staged.class_candidates)


def test_config3_loss_416_at_batch_32_against_the_oracle_sample(gpu, c_oracle):
    """Config 3's per-GPU shape (B = 32, G = 100): four of the images against the oracle, all of
    them against a call of their own (per-image values do not depend on the batch)."""
    case = cases.v3()
    B = 32
    heads = case.heads(3003, B)
    gt = case.gt(3004, B, 100, 101)
    hd = case.head_obj()
    out = hd.loss(to_dev(heads, gpu), torch.from_numpy(gt).to(gpu), 1., 1., 5., 1., 0.5, True, want_parts=True)
    pick = [0, 9, 17, 31]
    per, parts, _, _ = c_oracle.loss(case.oracle_cfg(c_oracle), [h[pick] for h in heads], gt[pick], (1., 1., 5., 1.), 0.5, True)
    assert_close(out['per_image'].cpu().numpy()[pick], per, what='per-image loss')
    assert_close(out['parts'].cpu().numpy()[pick], parts, what='loss parts')
    pair = out['reduce_pair'].cpu().numpy()
    assert pair[1] == B and abs(pair[0] - out['per_image'].double().sum().item()) <= 1e-9 * abs(pair[0])
