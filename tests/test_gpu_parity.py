"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on identical seeded inputs.

Bars: kept box rows / class ids / candidate counts bit-exact; decoded boxes,
confidences, IoU and loss within 1e-5 relative (fp32), the tolerance
BASELINE.json's north_star states."""
import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from oracle import yolo_head_oracle as O
from tests import cases

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def to_dev(arrs, dev):
    return [torch.from_numpy(a).to(dev) for a in arrs]


def assert_close(got, want, rtol=RTOL, atol=0.0, what=''):
    """Non-finite values are compared first and exactly: NaN only where the oracle has NaN,
    +-inf only where the oracle has the same infinity; the tolerance applies to the rest."""
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    assert got.shape == want.shape, '%s: shape %s against %s' % (what, got.shape, want.shape)
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), '%s: NaN at %d places, the oracle has %d (first mismatch at %s)' % (
        what, nan_g.sum(), nan_w.sum(), np.argwhere(nan_g != nan_w)[0])
    inf = np.isinf(want) | np.isinf(got)
    assert np.array_equal(got[inf], want[inf]), '%s: infinities differ (%d places)' % (what, inf.sum())
    fin = ~(nan_w | inf)
    g, w = got[fin], want[fin]
    if g.size == 0:
        return
    err = np.abs(g - w)
    tol = atol + rtol * np.abs(w)
    bad = err > tol
    assert not bad.any(), '%s: %d of %d beyond rtol %g (max err %g at want %g)' % (
        what, bad.sum(), bad.size, rtol, err.max(), w[np.argmax(err)])


@pytest.mark.parametrize('case_fn,B', [(cases.v3, 3), (cases.v2, 5), (cases.stress, 2)])
def test_decode_matches_oracle(gpu, c_oracle, case_fn, B):
    case = case_fn()
    heads = case.heads(11, B)
    boxes, conf = case.head_obj().decode(to_dev(heads, gpu))
    wb, wc = c_oracle.decode(case.oracle_cfg(c_oracle), heads)
    # boxes are differences of O(size) terms: tolerance is relative to the image size
    assert_close(boxes.cpu().numpy(), wb, rtol=RTOL, atol=RTOL * case.size, what='boxes')
    assert_close(conf.cpu().numpy(), wc, rtol=RTOL, atol=1e-30, what='conf')
    # NumPy restatement on image 0 as a second witness
    nb, nc = O.decode([h[0] for h in heads], case.priors, case.box_mult, case.C)
    assert_close(boxes[0].cpu().numpy(), nb, rtol=RTOL, atol=RTOL * case.size, what='boxes vs numpy')
    assert_close(conf[0].cpu().numpy(), nc, rtol=RTOL, atol=1e-30, what='conf vs numpy')


@pytest.mark.parametrize('case_fn', [cases.v3, cases.v2])
def test_decode_at_extreme_logits(gpu, c_oracle, case_fn):
    """Every channel kind (class, t_yx, t_hw, objectness) swept over +-{10, 20, 40, 60, 80, 87} and
    magnitudes in between: the hardware exp2 must stay within 1e-5 of the oracle's expf however
    large the logit (exp(87) = 6e37 is the last power below the fp32 range; the boxes it yields
    overflow to +-inf after the multiplier, and must do so on both sides).  YOLOv3.py:338-350."""
    case = case_fn(size=160)
    K = case.C + 5
    heads = case.heads(17, 2)
    rng = np.random.Generator(np.random.PCG64(18))
    mags = np.array([10, 20, 40, 60, 80, 87], np.float32)
    for h in heads:
        rows = h.reshape(-1, K)                                 # a view: NHWC rows of one box each
        n = rows.shape[0]
        vals = (rng.choice(mags, size=(n, K)) * rng.choice([-1.0, 1.0], size=(n, K))).astype(np.float32)
        vals += rng.uniform(-0.5, 0.5, size=(n, K)).astype(np.float32) * (np.abs(vals) < 87)
        pick = rng.random((n, K)) < 0.35
        rows[pick] = vals[pick]
        rows[:len(mags), case.C + 2] = mags                     # t_h = each magnitude at least once ...
        rows[:len(mags), case.C + 3] = -mags                    # ... and t_w its negative
    boxes, conf = case.head_obj().decode(to_dev(heads, gpu))
    wb, wc = c_oracle.decode(case.oracle_cfg(c_oracle), heads)
    assert np.isinf(wb).any() and (np.abs(wb[np.isfinite(wb)]) > 1e30).any()
    assert_close(boxes.cpu().numpy(), wb, rtol=RTOL, atol=RTOL * case.size, what='boxes')
    assert_close(conf.cpu().numpy(), wc, rtol=RTOL, atol=1e-30, what='conf')


@pytest.mark.parametrize('grid,P', [((7, 1), 1), ((1, 9), 1), ((1, 1), 3), ((5, 4), 1), ((3, 1), 5)])
def test_decode_and_detect_on_degenerate_grids(gpu, grid, P):
    """One prior, one column, one cell: the row -> (cell, prior) -> (y, x) arithmetic of the
    kernels has no division, so its divisors of 1 are cases of their own."""
    C = 4
    rng = np.random.default_rng(5)
    priors = [rng.uniform(0.5, 3.0, size=(P, 2)).astype(np.float32)]
    heads = y.synth.make_heads(15, 2, [grid], P, C, obj_mean=1.0)
    hd = y.DetectionHead(C, priors, (32.0,), (32.0,))
    boxes, conf = hd.decode(to_dev(heads, gpu))
    for b in range(2):
        nb, nc = O.decode([h[b] for h in heads], priors, (32.0,), C)
        assert_close(boxes[b].cpu().numpy(), nb, rtol=RTOL, atol=RTOL * 32 * max(grid), what='boxes')
        assert_close(conf[b].cpu().numpy(), nc, rtol=RTOL, atol=1e-30, what='conf')
    got = hd.detect(to_dev(heads, gpu), 0.3, 0.5, 10)
    staged = hd.nms(boxes, conf, 0.3, 0.5, 10)
    cases.same_detections(got.per_image(), staged.per_image())
    for g, s in zip(got.per_image(), staged.per_image()):
        assert np.array_equal(g[1], s[1])                      # the fused path's boxes are the decode kernel's
    assert int(got.count.sum()) > 0


def test_decode_without_conf(gpu, c_oracle):
    case = cases.v2()
    heads = case.heads(12, 2)
    boxes, conf = case.head_obj().decode(to_dev(heads, gpu), want_conf=False)
    assert conf is None
    wb, _ = c_oracle.decode(case.oracle_cfg(c_oracle), heads, want_conf=False)
    assert_close(boxes.cpu().numpy(), wb, rtol=RTOL, atol=RTOL * case.size, what='boxes')


@pytest.mark.parametrize('case_fn,B,thr,iou,mb,dist', [
    (cases.v3, 4, 0.5, 0.5, 10, {}),
    (cases.v2, 6, 0.5, 0.5, 10, dict(obj_mean=0.0)),
    (cases.v3, 2, 0.05, 0.3, 100, dict(obj_mean=-1.0, cold_mean=-3.0)),
    (cases.stress, 3, 0.001, 0.5, 100, y.synth.STRESS),
])
def test_nms_stage_bit_exact_on_identical_decoded_inputs(gpu, c_oracle, case_fn, B, thr, iou, mb, dist):
    """K3+K4 on the GPU's own decoded boxes/conf vs the oracle NMS on the same arrays."""
    case = case_fn()
    heads = case.heads(21, B, **dist)
    hd = case.head_obj()
    boxes, conf = hd.decode(to_dev(heads, gpu))
    res = hd.nms(boxes, conf, thr, iou, mb, want_class_candidates=True)
    got = res.per_image()
    want = c_oracle.nms(boxes.cpu().numpy(), conf.cpu().numpy(), thr, iou, mb)
    cases.same_detections(got, want)
    for g, w in zip(got, want):
        assert np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32)), 'scores differ in bits'
        assert np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)), 'boxes differ in bits'
    cc = res.resolved().class_candidates.cpu().numpy()
    assert np.array_equal(cc, (conf.cpu().numpy() >= np.float32(thr)).sum(axis=1).astype(np.int32))
    assert sum(len(g[0]) for g in got) > 0


def test_timed_entry_points_give_the_untimed_results(gpu):
    """yh_detect_timed / yh_nms_timed: the same call with an event recorded on either side of the
    pass over the input; results equal the plain entry points', the events measure a positive time."""
    case = cases.v3()
    heads = to_dev(case.heads(33, 3, obj_mean=-1.0), gpu)
    hd = case.head_obj()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in evs:
        e.record()                                             # creates the hipEvent_t handles
    plain = hd.detect(heads, 0.3, 0.5, 10).per_image()
    timed = hd.detect(heads, 0.3, 0.5, 10, stream_events=(evs[0], evs[1])).per_image()
    boxes, conf = hd.decode(heads)
    plain_n = hd.nms(boxes, conf, 0.3, 0.5, 10).per_image()
    timed_n = hd.nms(boxes, conf, 0.3, 0.5, 10, stream_events=(evs[2], evs[3])).per_image()
    torch.cuda.synchronize()
    for got, want in ((timed, plain), (timed_n, plain_n), (timed_n, plain)):
        cases.same_detections(got, want)
        for g, w in zip(got, want):
            assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
    assert evs[0].elapsed_time(evs[1]) > 0 and evs[2].elapsed_time(evs[3]) > 0
    assert sum(len(g[0]) for g in plain) > 0


def _fragile(wbox, wconf_c, thr, iou_thr, tol=1e-5):
    """True when the oracle's own decisions for one (image, class) sit within
    float noise of a boundary: a confidence at the score threshold, two
    candidates with (nearly) equal scores, or a candidate pair at the IoU
    threshold.  There a 1-ulp difference in sigmoid/exp may legitimately flip
    the outcome."""
    near = np.nonzero(wconf_c >= thr * (1 - tol))[0]
    s = wconf_c[near].astype(np.float64)
    if np.any(np.abs(s - thr) <= tol * thr):
        return True
    o = np.sort(s)[::-1]
    if o.size > 1 and np.any((o[:-1] - o[1:]) <= tol * o[:-1]):
        return True
    b = wbox[near].astype(np.float64)
    y1, x1 = np.minimum(b[:, 0], b[:, 2]), np.minimum(b[:, 1], b[:, 3])
    y2, x2 = np.maximum(b[:, 0], b[:, 2]), np.maximum(b[:, 1], b[:, 3])
    area = (y2 - y1) * (x2 - x1)
    iy = np.maximum(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]), 0)
    ix = np.maximum(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]), 0)
    inter = iy * ix
    with np.errstate(invalid='ignore', divide='ignore'):
        iou = inter / (area[:, None] + area[None] - inter)
    return bool(np.any(np.abs(iou - iou_thr) <= tol))


@pytest.mark.parametrize('case_fn,B,thr,iou,mb,dist', [
    (cases.v3, 4, 0.5, 0.5, 10, {}),
    (cases.v2, 6, 0.5, 0.5, 10, dict(obj_mean=0.0)),
    (cases.v3, 2, 0.05, 0.3, 100, dict(obj_mean=-1.0, cold_mean=-3.0)),
    (cases.stress, 3, 0.001, 0.5, 100, y.synth.STRESS),
])
def test_detect_end_to_end(gpu, c_oracle, case_fn, B, thr, iou, mb, dist):
    """yh_detect (fused, from the head tensors) vs the oracle's decode + NMS.
    Kept rows must be identical per (image, class); the only excuse accepted for
    a class is that the oracle itself sits within 1e-5 of a decision boundary
    there (sigmoid/exp differ by an ulp between libm and the device), and such
    classes must stay rare.  Scores and boxes of matching classes: 1e-5."""
    case = case_fn()
    heads = case.heads(31, B, **dist)
    end_to_end_against_oracle(case, heads, thr, iou, mb, gpu, c_oracle)


def end_to_end_against_oracle(case, heads, thr, iou, mb, gpu, c_oracle, **kw):
    """The comparison behind every end-to-end detect test.  Prints `excused/checked` and the
    (image, class) pairs it excused; an excused pair must be one `_fragile` can name."""
    B = heads[0].shape[0]
    want = c_oracle.detect(case.oracle_cfg(c_oracle), heads, thr, iou, mb)
    wbox, wconf = c_oracle.decode(case.oracle_cfg(c_oracle), heads)
    got = case.head_obj().detect(to_dev(heads, gpu), thr, iou, mb, **kw).per_image()
    checked, excused = 0, []
    for b in range(B):
        for c in range(case.C):
            gm, wm = got[b][2] == c, want[b][2] == c
            if not wm.any() and not gm.any():
                continue
            checked += 1
            if np.array_equal(got[b][3][gm], want[b][3][wm]):
                assert_close(got[b][0][gm], want[b][0][wm], what='scores')
                assert_close(got[b][1][gm], want[b][1][wm], rtol=RTOL, atol=RTOL * case.size, what='bbox')
            else:
                assert _fragile(wbox[b], wconf[b, :, c], thr, iou), \
                    'image %d class %d: kept rows differ away from any decision boundary' % (b, c)
                excused.append((b, c))
    print('end to end %s B=%d thr=%g: excused/checked = %d/%d %s' % (case.name, B, thr, len(excused), checked, excused))
    assert checked > 0 and len(excused) <= max(1, checked // 50), '%d of %d classes excused: %s' % (
        len(excused), checked, excused)
    return len(excused), checked


def test_detect_equals_decode_then_nms_at_full_size(gpu):
    """Config-2 shape (416, C=80, 10647 boxes/img) at batch 16: the fused path
    must give exactly what its own staged path (full decode, then yh_nms) gives."""
    case = cases.v3()
    B = 16
    heads = y.synth.torch_heads(gpu, 5, B, case.grids, case.P, case.C)
    hd = case.head_obj()
    fused = hd.detect(heads, 0.5, 0.5, 10, want_class_candidates=True)
    boxes, conf = hd.decode(heads)
    staged = hd.nms(boxes, conf, 0.5, 0.5, 10, want_class_candidates=True)
    f, s = fused.per_image(), staged.per_image()
    cases.same_detections(f, s)
    for a, b in zip(f, s):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert torch.equal(fused.class_candidates, staged.class_candidates)
    assert int(fused.count.sum()) > 0


def test_both_launch_shapes_agree(gpu, c_oracle):
    """Capacity <= 2048 takes the single-launch candidates -> detections stage,
    a larger one the three-launch stage: identical results, and the same
    workspace can be reused call after call."""
    case = cases.v3()
    heads = case.heads(55, 3)
    dheads = to_dev(heads, gpu)
    hd = case.head_obj()
    want = c_oracle.nms(*[t.cpu().numpy() for t in hd.decode(dheads)], 0.5, 0.5, 10)
    for cap in (512, 513, 2048, 2049, 512, 100000):
        for _ in range(2):
            got = hd.detect(dheads, 0.5, 0.5, 10, max_candidates=cap, want_class_candidates=True)
            assert not got.overflowed()
            cases.same_detections(got.per_image(), want)


@pytest.mark.parametrize('cap,kernel', [(None, '1'), (2048, '1'), (5000, '1'), (None, '2'), (2048, '2')])
def test_pipelined_submit_matches_detect(gpu, tuning, cap, kernel):
    """yh_detect_pipelined: batch i streams while batch i-1 is selected inside the same
    launch (capacity 512), or right before it (larger capacities): detect()'s results,
    batch after batch, on alternating workspaces; with the objectness-first kernel (1) and
    with the streaming one (2)."""
    tuning(detect_kernel=int(kernel))
    case = cases.v3()
    hd = case.head_obj()
    sets = [y.synth.torch_heads(gpu, 70 + i, 4, case.grids, case.P, case.C) for i in range(5)]
    want = []
    for h in sets:
        w = hd.detect(h, 0.5, 0.5, 10, max_candidates=cap, want_class_candidates=True)
        want.append((w.per_image(), w.class_candidates.clone()))
    pipe = y.DetectPipeline(case.head_obj(), 0.5, 0.5, 10, max_candidates=cap, want_class_candidates=True)
    got = []
    for h in sets:
        d = pipe.submit(h)
        if d is not None:
            got.append((d.per_image(), d.class_candidates.clone(), d.overflowed()))
    d = pipe.flush()
    got.append((d.per_image(), d.class_candidates.clone(), d.overflowed()))
    assert pipe.flush() is None and len(got) == len(want)
    for (gi, gc, ov), (wi, wc) in zip(got, want):
        assert not ov
        cases.same_detections(gi, wi)
        for a, b in zip(gi, wi):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert torch.equal(gc, wc)
    assert sum(len(g[0][0][0]) for g in got) > 0


def test_pipeline_keeps_up_with_changing_batches(gpu):
    """A pipeline prepares its call once per slot and reuses it while batches keep their
    shape; batches of another size (after a flush), a return to the first size, and a
    strided head tensor in between all give detect()'s results."""
    case = cases.v3()
    hd = case.head_obj()
    pipe = y.DetectPipeline(case.head_obj(), 0.5, 0.5, 10)
    seed = 300
    for B, n in ((4, 4), (2, 1), (4, 3), (3, 4)):
        sets = [y.synth.torch_heads(gpu, seed + i, B, case.grids, case.P, case.C) for i in range(n)]
        seed += n
        if n >= 3:                                   # same values behind a non-contiguous view
            wide = torch.zeros(tuple(sets[2][1].shape[:3]) + (2 * sets[2][1].shape[3],), device=gpu)
            wide[..., ::2] = sets[2][1]
            sets[2] = [sets[2][0], wide[..., ::2], sets[2][2]]
            assert not sets[2][1].is_contiguous()
        want = [hd.detect(h, 0.5, 0.5, 10).per_image() for h in sets]
        got = []
        for h in sets:
            d = pipe.submit(h)
            if d is not None:
                got.append(d.per_image())
        got.append(pipe.flush().per_image())
        assert len(got) == len(want)
        for g, w in zip(got, want):
            cases.same_detections(g, w)
            for a, b in zip(g, w):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize('lanes', [1, 2, 3])
def test_lanes_on_several_streams_match_detect(gpu, lanes):
    """DetectLanes: pipelines on streams of their own, fed in turn; every batch's result is
    detect()'s, in submission order, whatever the number of lanes."""
    case = cases.v3()
    hd = case.head_obj()
    sets = [y.synth.torch_heads(gpu, 170 + i, 4, case.grids, case.P, case.C) for i in range(7)]
    want = [hd.detect(h, 0.5, 0.5, 10).per_image() for h in sets]
    torch.cuda.synchronize()
    pipe = y.DetectLanes(case.head_obj(), 0.5, 0.5, 10, lanes=lanes)
    got = []
    for h in sets:
        d = pipe.submit(h, wait_current=True)
        if d is not None:
            got.append(d.per_image())
    got += [d.per_image() for d in pipe.flush()]
    assert len(got) == len(want) and pipe.flush() == []
    for g, w in zip(got, want):
        cases.same_detections(g, w)


def test_candidates_then_select_calls(gpu):
    """The two halves of yh_detect called one after the other through the C ABI."""
    import ctypes
    from yolo_tf_amd import _abi
    case = cases.v3(size=160)
    hd = case.head_obj()
    heads = y.synth.torch_heads(gpu, 81, 3, case.grids, case.P, case.C, obj_mean=-1.0)
    want = hd.detect(heads, 0.4, 0.5, 10, want_class_candidates=True)
    (cfg, B, N, ws), = hd._plans.values()
    cap = hd.default_max_candidates(N)
    outs = hd._alloc_outputs(B, 10, gpu, True)
    lib = _abi.load()
    st = ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    for _ in range(2):
        _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), _abi.pointer_array([h.data_ptr() for h in heads]), B,
                                            0.4, 10, cap, ws.data_ptr(), ws.numel(), st))
        _abi.check(lib.yh_detect_select(ctypes.byref(cfg), B, 0.5, 10, cap, ws.data_ptr(), ws.numel(),
                                        *[t.data_ptr() for t in outs], st))
    got = y.Detections(*outs, None)
    assert int(got.count.sum()) > 0 and not got.overflowed()
    cases.same_detections(got.per_image(), want.per_image())
    assert torch.equal(got.class_candidates, want.class_candidates)


def test_select_finds_its_streaming_call_among_many_workspaces(gpu):
    """With a dense candidate capacity yh_detect_select decodes the boxes it looks at from the head tensors
    of the yh_detect_candidates call on its workspace, which the library keeps on record for the 64
    workspaces streamed into most recently.  A workspace that was on record long ago and streams again
    keeps its record while a few other workspaces stream for the first time (a table that replaced entries
    in turn, whatever their age, lost it: 'no streaming call on this workspace is on record'); after 64
    others it is gone and the select call says so."""
    import ctypes
    from yolo_tf_amd import _abi
    case = cases.stress(grid=20, P=10, C=3)                     # 4000 boxes x 3 classes
    hd = case.head_obj()
    B, mb = 2, 5
    heads = y.synth.torch_heads(gpu, 83, B, case.grids, case.P, case.C, obj_mean=0.0)
    cap = 4096                                                  # beyond the single-launch form: no boxes in the workspace
    want = hd.detect(heads, 0.05, 0.5, mb, max_candidates=cap)
    (cfg, _, N, ws0), = hd._plans.values()
    lib = _abi.load()
    st = ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    hp = _abi.pointer_array([h.data_ptr() for h in heads])
    spaces = [torch.zeros_like(ws0) for _ in range(64 + 70)]
    outs = hd._alloc_outputs(B, mb, gpu, False)

    def candidates(ws):
        _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), hp, B, 0.05, mb, cap, ws.data_ptr(), ws.numel(), st))

    def select(ws):
        _abi.check(lib.yh_detect_select(ctypes.byref(cfg), B, 0.5, mb, cap, ws.data_ptr(), ws.numel(),
                                        *[t.data_ptr() if t is not None else None for t in outs], st))
        cases.same_detections(y.Detections(*outs, None).per_image(), want.per_image())

    for ws in spaces[:64]:                                      # the table is full, in this order
        candidates(ws)
        select(ws)
    mine = spaces[1]
    candidates(mine)                                            # on record already, and now the most recent
    for ws in spaces[64:67]:
        candidates(ws)
        select(ws)
    select(mine)
    for ws in spaces[67:67 + 64]:                               # 64 others since: its record is gone
        candidates(ws)
        select(ws)
    with pytest.raises(_abi.YhError, match='on record'):
        select(mine)


@pytest.mark.parametrize('obj_mean,hot_mean', [(-1.0, 2.0), (0.0, 0.0), (2.0, -1.0), (9.0, -2.0)])
def test_streaming_variants_agree(gpu, tuning, obj_mean, hot_mean):
    """The objectness-first kernel (the default; one to many steps per work unit), the
    wave-private streaming kernel (4 and 8 KiB tiles, 2..4 buffers, few and many work units)
    and the workgroup-tile kernel find the same candidates, from a few percent of the rows
    surviving the objectness prefilter to all of them."""
    case = cases.v3(size=160)
    heads = case.heads(77, 3, obj_mean=obj_mean, hot_mean=hot_mean)
    dheads = to_dev(heads, gpu)
    cap = 2048
    tuning(detect_kernel=2, stream_tile_kib=1)                # workgroup tiles: the kernel of yh_nms and of dense candidate sets
    ref = case.head_obj().detect(dheads, 0.4, 0.5, 10, max_candidates=cap, want_class_candidates=True)
    assert not ref.overflowed()
    want, want_cc = ref.per_image(), ref.class_candidates.clone()
    assert int(ref.count.sum()) > 0
    for steps in (0, 1, 2, 3, 5, 50):
        tuning(row_steps=steps)
        got = case.head_obj().detect(dheads, 0.4, 0.5, 10, max_candidates=cap, want_class_candidates=True)
        assert not got.overflowed()
        cases.same_detections(got.per_image(), want)
        for a, b in zip(got.per_image(), want):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert torch.equal(got.class_candidates, want_cc)
    for ni, nbuf, units in ((8, 3, 8192), (8, 2, 64), (4, 2, 300), (4, 3, 7), (4, 4, 100000)):
        tuning(detect_kernel=2, stream_tile_kib=ni, stream_buffers=nbuf, stream_units=units)
        got = case.head_obj().detect(dheads, 0.4, 0.5, 10, max_candidates=cap, want_class_candidates=True)
        assert not got.overflowed()
        cases.same_detections(got.per_image(), want)
        assert torch.equal(got.class_candidates, want_cc)


@pytest.mark.parametrize('C,P,grid,share,cold', [(80, 3, 30, 0.01, 0.0), (97, 5, 24, 0.006, 1.0), (130, 3, 20, 0.015, -1.0),
                                                 (20, 9, 26, 0.015, 0.5), (80, 3, 30, 0.1, -3.0), (80, 3, 30, 0.6, -3.0)])
def test_rows_kernel_with_many_candidates_per_row(gpu, tuning, C, P, grid, share, cold):
    """Few to many rows pass the objectness prefilter and most of their classes pass the score threshold: a
    batch of sixteen rows then holds several hundred class logits for the exact test (more than its pending
    list takes at once), a wavefront stages more candidates than its unit's private slots hold, and the
    image's list takes the rest.  The objectness-first kernel (several unit lengths) against the staged path
    (full decode, then the tile kernel over the confidence matrix) and against the oracle's candidate counts."""
    rng = np.random.Generator(np.random.PCG64(7 + C))
    priors = [rng.uniform(0.5, 3.0, size=(P, 2)).astype(np.float32)]
    heads = y.synth.make_heads(300 + C, 2, [(grid, grid)], P, C, hot_mean=2.0, cold_mean=cold, cold_std=1.5)
    K = C + 5
    rows = heads[0].reshape(-1, K)
    alive = rng.random(rows.shape[0]) < share
    rows[:, K - 1] = np.where(alive, rng.normal(3.0, 1.0, rows.shape[0]), -12.0).astype(np.float32)
    hd = y.DetectionHead(C, priors, (32.0,), (32.0,))
    dheads = to_dev(heads, gpu)
    thr, cap = 0.3, 2048
    boxes, conf = hd.decode(dheads)
    want_cc = (conf.cpu().numpy() >= np.float32(thr)).sum(axis=1).astype(np.int32)
    per_image = want_cc.sum(axis=1)
    assert per_image.max() > 16 * 4
    staged = hd.nms(boxes, conf, thr, 0.5, 20, max_candidates=max(cap, int(per_image.max()) + 1), want_class_candidates=True)
    assert not staged.overflowed()
    want = staged.per_image()
    for steps in (0, 1, 3, 16):
        tuning(row_steps=steps)
        got = y.DetectionHead(C, priors, (32.0,), (32.0,)).detect(dheads, thr, 0.5, 20, max_candidates=cap, want_class_candidates=True)
        if per_image.max() > cap:
            assert got.overflowed()                            # the capacity is the caller's; what was seen is reported
            continue
        assert not got.overflowed()
        cases.same_detections(got.per_image(), want)
        for a, b in zip(got.per_image(), want):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(got.class_candidates.cpu().numpy(), want_cc)


def test_tile_kernel_with_a_fixed_number_of_workgroups(gpu, tuning):
    """yh_tuning.stream_resident: the workgroup-tile kernel launched as one workgroup per compute unit,
    each walking several runs of tiles (446 runs here), against one workgroup per run."""
    case = cases.stress(grid=40)
    dheads = to_dev(case.heads(78, 2, **y.synth.STRESS), gpu)
    cap = 40000
    ref = case.head_obj().detect(dheads, 0.001, 0.5, 100, max_candidates=cap, want_class_candidates=True)
    assert not ref.overflowed() and int(ref.count.sum()) > 0
    want, want_cc = ref.per_image(), ref.class_candidates.clone()
    for blocks, resident in ((100000, 1), (100000, 2), (300, 1)):
        tuning(stream_blocks=blocks, stream_resident=resident)
        got = case.head_obj().detect(dheads, 0.001, 0.5, 100, max_candidates=cap, want_class_candidates=True)
        assert not got.overflowed()
        cases.same_detections(got.per_image(), want)
        for a, b in zip(got.per_image(), want):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert torch.equal(got.class_candidates, want_cc)


def test_rows_wider_than_a_wave_tile(gpu):
    """2100 classes: one row (8420 B) does not fit the wave-private tiles, the
    workgroup-tile kernel takes over; 600 classes: several 64-lane passes per row."""
    for C, thr in ((2100, 0.3), (600, 0.3)):
        grids = [(3, 2), (2, 2)]
        rng = np.random.Generator(np.random.PCG64(5))
        priors = [rng.uniform(0.5, 3.0, size=(2, 2)).astype(np.float32) for _ in grids]
        case = cases.Case('wide', grids, priors, (32.0, 16.0), (32.0, 16.0), C, 64, 3)
        dheads = to_dev(case.heads(78, 2, obj_mean=1.0, hot_mean=1.0), gpu)
        hd = case.head_obj()
        fused = hd.detect(dheads, thr, 0.5, 5, want_class_candidates=True).resolved()
        staged = hd.nms(*hd.decode(dheads), thr, 0.5, 5, want_class_candidates=True).resolved()
        assert int(fused.count.sum()) > 0
        cases.same_detections(fused.per_image(), staged.per_image())
        assert torch.equal(fused.class_candidates, staged.class_candidates)


def test_detect_is_deterministic(gpu):
    case = cases.stress(grid=40)
    heads = y.synth.torch_heads(gpu, 6, 4, case.grids, case.P, case.C, **y.synth.STRESS)
    hd = case.head_obj()
    a = hd.detect(heads, 0.001, 0.5, 100).per_image()
    for _ in range(3):
        b = hd.detect(heads, 0.001, 0.5, 100).per_image()
        for x, z in zip(a, b):
            assert all(np.array_equal(p, q) for p, q in zip(x, z))


def test_candidate_overflow_is_reported_and_rerun(gpu, c_oracle):
    case = cases.v2()
    heads = case.heads(41, 2, obj_mean=2.0)
    hd = case.head_obj()
    res = hd.detect(to_dev(heads, gpu), 0.3, 0.5, 10, max_candidates=4)
    assert res.overflowed()
    # the status words say how many candidates the overflowed images had, in units of 256
    conf = case.head_obj().decode(to_dev(heads, gpu))[1].cpu().numpy()
    most = int((conf >= np.float32(0.3)).sum(axis=(1, 2)).max())
    assert res.needed_candidates() == (most + 255) // 256 * 256
    want = c_oracle.detect(case.oracle_cfg(c_oracle), heads, 0.3, 0.5, 10)
    cases.same_detections(res.per_image(), want)


@pytest.mark.parametrize('cap', [40000, 9000])
def test_long_candidate_lists_are_binned_in_slices(gpu, c_oracle, cap):
    """About 16000 candidates per image: the class binning runs over several slices of the
    list (capacity beyond one slice), twice on the same workspace; with a capacity below
    the count the overflow is flagged and the rerun gives the full answer."""
    case = cases.stress(grid=40, P=10, C=80)
    heads = case.heads(61, 2, **y.synth.STRESS)
    dheads = to_dev(heads, gpu)
    hd = case.head_obj()
    boxes, conf = hd.decode(dheads)
    want = c_oracle.nms(boxes.cpu().numpy(), conf.cpu().numpy(), 0.001, 0.5, 100)
    assert (conf.cpu().numpy() >= 0.001).sum(axis=(1, 2)).min() > 9000
    for _ in range(2):
        res = hd.detect(dheads, 0.001, 0.5, 100, max_candidates=cap, want_class_candidates=True)
        assert res.overflowed() == (cap == 9000)
        cases.same_detections(res.per_image(), want)
    staged = hd.nms(boxes, conf, 0.001, 0.5, 100, max_candidates=cap)
    cases.same_detections(staged.per_image(), want)


@pytest.mark.parametrize('C', [1100, 300])
def test_dense_candidates_over_many_classes(gpu, c_oracle, C):
    """A dense candidate set (capacity beyond the single-launch stage) with more classes than the slice-sorting
    scatter keeps tables for (1024) and with fewer: the class counts come from the kernel that lists the
    candidates, the scatter is the plain one above 1024 classes and the sorting one below.  Selections, class
    counts and the second call on the same workspace against the oracle."""
    rng = np.random.default_rng(C)
    B, N = 2, 500
    yx = rng.uniform(20, 380, size=(B, N, 2)).astype(np.float32)
    hw = rng.uniform(10, 90, size=(B, N, 2)).astype(np.float32)
    boxes = np.concatenate([yx - hw / 2, yx + hw / 2], axis=-1).astype(np.float32)
    conf = rng.uniform(0.0, 1.0, size=(B, N, C)).astype(np.float32)
    conf[rng.uniform(size=conf.shape) < 0.97] = 0.0                 # about 3 % of the pairs are candidates
    per_class = (conf >= np.float32(0.05)).sum(axis=1)
    assert per_class.sum(axis=1).min() > 2048
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    want = c_oracle.nms(boxes, conf, 0.05, 0.4, 5)
    dboxes, dconf = torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu)
    for _ in range(2):
        res = hd.nms(dboxes, dconf, 0.05, 0.4, 5, max_candidates=20480, want_class_candidates=True)
        assert not res.overflowed()
        cases.same_detections(res.per_image(), want)
        assert np.array_equal(res.class_candidates.cpu().numpy(), per_class)


def test_long_segment_sorts_in_global_memory(gpu, c_oracle):
    """One class takes every box: a 4000-candidate segment (beyond the LDS sort)."""
    case = cases.stress(grid=20, P=10, C=3)
    heads = case.heads(51, 1, obj_mean=3.0, hot_mean=3.0, cold_mean=3.0)
    hd = case.head_obj()
    boxes, conf = hd.decode(to_dev(heads, gpu))
    got = hd.nms(boxes, conf, 0.01, 0.5, 200).per_image()
    want = c_oracle.nms(boxes.cpu().numpy(), conf.cpu().numpy(), 0.01, 0.5, 200)
    cases.same_detections(got, want)
    assert (conf.cpu().numpy() >= 0.01).sum(axis=1).max() > 2048


@pytest.mark.parametrize('N,C,clusters,mb,iou', [
    (1500, 2, 6, 20, 0.5),      # ~750 per class around 6 objects: every tile is contested, the sorted prefix runs out
    (6000, 3, 40, 100, 0.3),    # ~2000 per class: LDS-sized segments, selection beyond the first tile
    (9000, 2, 15, 200, 0.4),    # > 2048 per class: segments sorted in global memory
    (300, 4, 5, 60, 0.3),       # short segments: single-launch stage
])
def test_nms_of_clustered_boxes(gpu, c_oracle, N, C, clusters, mb, iou):
    """Boxes jittered around a few objects, as a detector's candidates are: most candidates are
    suppressed, tiles of the walk are full of mutual overlaps, and max_boxes is often not reached
    inside the sorted prefix.  Selections bit-identical to the oracle's greedy walk."""
    rng = np.random.default_rng(N + C)
    B = 2
    centre = rng.uniform(60, 340, size=(B, clusters, 2)).astype(np.float32)
    size = rng.uniform(30, 120, size=(B, clusters, 2)).astype(np.float32)
    which = rng.integers(0, clusters, size=(B, N))
    c = np.take_along_axis(centre, which[..., None], axis=1) + rng.normal(0, 6, size=(B, N, 2)).astype(np.float32)
    s = np.take_along_axis(size, which[..., None], axis=1) * rng.uniform(0.8, 1.25, size=(B, N, 2)).astype(np.float32)
    boxes = np.concatenate([c - s / 2, c + s / 2], axis=-1).astype(np.float32)
    conf = rng.uniform(0.0, 1.0, size=(B, N, C)).astype(np.float32)
    conf[rng.uniform(size=conf.shape) < 0.5 / C] = 0.0         # some pairs below any threshold
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    got = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), 0.05, iou, mb).per_image()
    want = c_oracle.nms(boxes, conf, 0.05, iou, mb)
    cases.same_detections(got, want)
    for g, w in zip(got, want):
        assert np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32))
        assert np.array_equal(g[3], w[3])
    assert 0 < sum(len(g[0]) for g in got) < B * C * mb       # suppression, not max_boxes, ended most walks


@pytest.mark.parametrize('iou', [0.5, 0.3, 0.7])
def test_overlaps_at_the_iou_threshold(gpu, c_oracle, iou):
    """Pairs whose IoU sits on the threshold or a few ulps around it: the overlap test decides most
    pairs without the division, these take it; the comparison is strict (an IoU equal to the
    threshold does not suppress).  One pair per class; selections as the oracle's."""
    C = 96
    rng = np.random.default_rng(17)
    boxes = np.zeros((1, 2 * C, 4), np.float32)
    conf = np.zeros((1, 2 * C, C), np.float32)
    t = np.float32(iou)
    for c in range(C):
        h, w = np.float32(rng.uniform(3, 300)), np.float32(rng.uniform(3, 300))
        y0, x0 = np.float32(rng.uniform(0, 50)), np.float32(rng.uniform(0, 50))
        frac = t                                       # B inside A, IoU = area(B) / area(A) ~ frac
        for _ in range(c % 9):                         # 0..8 ulps above, then below, the threshold
            frac = np.nextafter(frac, np.float32(2.0 if (c // 9) % 2 == 0 else 0.0), dtype=np.float32)
        boxes[0, 2 * c] = (y0, x0, y0 + h, x0 + w)
        boxes[0, 2 * c + 1] = (y0, x0, y0 + h, x0 + w * frac)
        conf[0, 2 * c, c], conf[0, 2 * c + 1, c] = 0.9, 0.8
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    got = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), 0.5, iou, 10).per_image()
    want = c_oracle.nms(boxes, conf, 0.5, iou, 10)
    cases.same_detections(got, want)
    kept = len(got[0][0])
    assert C < kept < 2 * C                            # some second boxes suppressed, some not


def test_tied_scores_take_lower_row_first(gpu, c_oracle):
    rng = np.random.default_rng(3)
    B, N, C = 2, 300, 4
    boxes = rng.uniform(0, 200, size=(B, N, 2)).astype(np.float32)
    boxes = np.concatenate([boxes, boxes + rng.uniform(5, 60, size=(B, N, 2)).astype(np.float32)], axis=-1)
    conf = rng.choice(np.array([0.2, 0.6, 0.7, 0.9], np.float32), size=(B, N, C))
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    got = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), 0.5, 0.4, 50).per_image()
    want = c_oracle.nms(boxes, conf, 0.5, 0.4, 50)
    cases.same_detections(got, want)


def test_empty_result_and_empty_batch(gpu):
    case = cases.v2()
    hd = case.head_obj()
    heads = to_dev(case.heads(61, 2, obj_mean=-30.0), gpu)
    out = hd.detect(heads, 0.5, 0.5, 10).per_image()
    assert all(len(o[0]) == 0 and o[1].shape == (0, 4) for o in out)
    empty = [torch.empty((0, 13, 13, 125), dtype=torch.float32, device=gpu)]
    assert hd.detect(empty, 0.5, 0.5, 10).per_image() == []


def test_threshold_edges(gpu, c_oracle):
    """thr <= 0 passes every pair (max_candidates = N*C); thr > 1 passes none."""
    case = cases.v2(C=3)
    heads = case.heads(71, 1)
    hd = case.head_obj()
    got = hd.detect(to_dev(heads, gpu), 0.0, 0.5, 5, want_class_candidates=True)
    assert np.all(got.resolved().class_candidates.cpu().numpy() == case.N)
    assert all(len(o[0]) == 0 for o in hd.detect(to_dev(heads, gpu), 1.5, 0.5, 5).per_image())


@pytest.mark.parametrize('thr', [0.0, -1.0])
def test_non_positive_threshold_on_the_rows_kernel(gpu, c_oracle, thr):
    """thr <= 0 turns the objectness prefilter into -inf; with N * C <= 2048 the call stays on the
    objectness-first kernel, whose lanes beyond a unit's rows must not pass that test: every
    (box, class) pair of the image is a candidate, none beyond it, detections as the oracle's."""
    case = cases.v2(C=2)                                       # 845 boxes x 2 classes = 1690 pairs
    heads = case.heads(73, 3)
    hd = case.head_obj()
    res = hd.detect(to_dev(heads, gpu), thr, 0.5, 7, max_candidates=case.N * case.C, want_class_candidates=True)
    assert not res.overflowed()
    assert np.all(res.class_candidates.cpu().numpy() == case.N)
    want = c_oracle.detect(case.oracle_cfg(c_oracle), heads, thr, 0.5, 7)
    got = res.per_image()
    cases.same_detections(got, want)
    for g, w in zip(got, want):
        assert_close(g[0], w[0], what='scores')
        assert g[3].max() < case.N


@pytest.mark.parametrize('mb', [512, 400])
def test_nms_prefix_selection_at_the_largest_max_boxes(gpu, c_oracle, mb):
    """nms_max_boxes 512 and class segments of 1640..2048 candidates: the radix-selected prefix of
    the score order may not outgrow its LDS buffer (1024 keys); selections as the oracle's."""
    rng = np.random.default_rng(23)
    B, N, C = 2, 2048, 2
    centre = rng.uniform(0, 3000, size=(B, N, 2)).astype(np.float32)
    size = rng.uniform(20, 60, size=(B, N, 2)).astype(np.float32)
    boxes = np.concatenate([centre - size / 2, centre + size / 2], axis=-1).astype(np.float32)
    conf = rng.uniform(0.2, 1.0, size=(B, N, C)).astype(np.float32)
    conf[1, 1700:, 1] = 0.0                                    # one segment of 1700
    # scores bunched so that the first radix level's boundary bin overshoots the target
    conf[0, :, 0] = (0.5 + 0.001 * rng.uniform(size=N)).astype(np.float32)
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    got = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), 0.1, 0.5, mb,
                 max_candidates=N * C).per_image()
    want = c_oracle.nms(boxes, conf, 0.1, 0.5, mb)
    cases.same_detections(got, want)
    assert max(len(g[0]) for g in got) > mb


@pytest.mark.parametrize('group_of_wave', [0, 1, 2, 3])
@pytest.mark.parametrize('left', [1, 2, 3])
def test_tile_kernel_reads_the_partial_last_slot_of_a_tensor(gpu, left, group_of_wave):
    """A confidence matrix whose float count is not a multiple of four ends in a partial 16-byte slot, which the
    workgroup-tile kernel stores into its LDS tile with a plain store of wavefront 0, while the window's last
    group of whole slots is an LDS-DMA load of whichever wavefront the group falls to, with its lanes past the
    end switched off (a lane that wrote there too would race with that store: seen once in some 70000 first
    calls as a count that was one short, the partial slot holding a copy of the slot before it).  The only
    candidates are the partial slot's floats, then only the last whole slot's; either way the counts and the
    kept rows are exact, with the group on each of the four wavefronts."""
    B, C = 1, 1
    whole = 64 * (12 + group_of_wave) + 37                      # whole slots: the last group is the tile's 13th..16th
    N = 4 * whole + left
    boxes = np.zeros((B, N, 4), np.float32)
    boxes[0, :, 0] = boxes[0, :, 1] = 100.0 * np.arange(N, dtype=np.float32)      # nothing overlaps
    boxes[0, :, 2:] = boxes[0, :, :2] + 10.0
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    for which in ('partial', 'before'):
        conf = np.full((B, N, C), 0.01, np.float32)
        rows = np.arange(4 * whole, N) if which == 'partial' else np.arange(4 * whole - 4, 4 * whole)
        conf[0, rows, 0] = np.linspace(0.9, 0.6, len(rows), dtype=np.float32)
        for _ in range(3):
            res = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), 0.3, 0.5, 20,
                         max_candidates=N * C, want_class_candidates=True)
            assert not res.overflowed()
            assert res.class_candidates.cpu().numpy().tolist() == [[len(rows)]], which
            got = res.per_image()[0]
            assert got[3].tolist() == rows.tolist() and np.array_equal(got[0], conf[0, rows, 0]), which


@pytest.mark.parametrize('dist', ['ties', 'one_value_band', 'log_uniform', 'two_bands', 'half_zero_bits'])
def test_prefix_selection_over_awkward_score_distributions(gpu, c_oracle, dist):
    """Segments of about 1500 candidates (sorted in LDS through a selected prefix that is grouped by two
    score bytes and ranked inside its groups) with scores that make those groups degenerate: a handful of
    distinct values (ties resolved by row, groups of hundreds of equal scores), one value band narrower than
    a group, magnitudes over three decades (many top bytes), two far-apart bands, and scores whose low
    mantissa bits are all zero.  Overlapping boxes, so the walk often needs more than the prefix."""
    rng = np.random.default_rng(len(dist))
    B, N, C = 2, 1500, 2
    centre = rng.uniform(0, 900, size=(B, N, 2)).astype(np.float32)
    size = rng.uniform(20, 160, size=(B, N, 2)).astype(np.float32)
    boxes = np.concatenate([centre - size / 2, centre + size / 2], axis=-1).astype(np.float32)
    u = rng.uniform(size=(B, N, C))
    if dist == 'ties':
        conf = rng.choice(np.array([0.3, 0.55, 0.5500001, 0.8, 0.95], np.float32), size=(B, N, C))
    elif dist == 'one_value_band':
        conf = (0.75 + 1e-5 * u).astype(np.float32)
    elif dist == 'log_uniform':
        conf = np.exp(np.log(2e-3) + u * (np.log(1.0) - np.log(2e-3))).astype(np.float32)
    elif dist == 'two_bands':
        conf = np.where(u < 0.1, 0.9 + 0.05 * u, 0.01 + 0.001 * u).astype(np.float32)
    else:
        conf = (np.floor(u * 64) / 64 + 1 / 128).astype(np.float32)
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    for mb in (20, 100):
        got = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), 0.001, 0.45, mb,
                     max_candidates=N * C).per_image()
        want = c_oracle.nms(boxes, conf, 0.001, 0.45, mb)
        cases.same_detections(got, want)


def test_loss_at_the_largest_ground_truth_block(gpu, c_oracle):
    """pad 2048 (the documented bound): 90 KiB of LDS for the ground-truth block, beyond the
    default limit of dynamic LDS."""
    case = cases.v3(size=96, C=6)
    B, pad, G = 2, 2048, 2040
    heads = case.heads(83, B)
    gt = case.gt(84, B, G, pad)
    out = case.head_obj().loss(to_dev(heads, gpu), torch.from_numpy(gt).to(gpu), 1., 1., 5., 1., 0.5, True,
                               want_parts=True, want_iou_max=True)
    per, parts, iou, _ = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, (1., 1., 5., 1.), 0.5, True, want_iou=True)
    assert np.array_equal(out['iou_max'].cpu().numpy().view(np.uint32), iou.view(np.uint32))
    assert_close(out['parts'].cpu().numpy(), parts, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, what='per-image loss')
    fast = case.head_obj().loss(to_dev(heads, gpu), torch.from_numpy(gt).to(gpu), 1., 1., 5., 1., 0.5, True)
    assert torch.equal(fast['per_image'], out['per_image'])
    with pytest.raises(y._abi.YhError) as e:
        case.head_obj().loss(to_dev(heads, gpu), torch.from_numpy(case.gt(84, B, 5, 2049)).to(gpu), 1., 1., 5., 1., 0.5, True)
    assert e.value.code == y._abi.YH_ERR_LIMIT


@pytest.mark.parametrize('G,B', [(100, 4), (7, 3), (1, 2)])
def test_loss_v3_matches_oracle(gpu, c_oracle, G, B):
    case = cases.v3()
    heads = case.heads(81, B)
    gt = case.gt(82, B, G, G + 5)
    out = case.head_obj().loss(to_dev(heads, gpu), torch.from_numpy(gt).to(gpu), 1., 1., 5., 1., 0.5, True,
                               want_parts=True, want_iou_max=True)
    per, parts, iou, mean = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, (1., 1., 5., 1.), 0.5, True, want_iou=True)
    # IoU: + - * / min max only, same order on both sides -> identical bits
    assert np.array_equal(out['iou_max'].cpu().numpy().view(np.uint32), iou.view(np.uint32))
    assert_close(out['parts'].cpu().numpy(), parts, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, what='per-image loss')
    assert_close(out['sum'].item() / B, mean, what='mean loss')
    assert np.float32(out['mean'].item()) == np.float32(out['sum'].item()) / np.float32(B)     # the kernel's own fp32 mean
    # NumPy restatement, image 0
    n0 = O.loss_v3_image([h[0] for h in heads], gt[0], case.priors, case.C)
    assert_close(out['per_image'][0].item(), n0, what='loss vs numpy')


@pytest.mark.parametrize('thr', [0.0, 0.3, 0.5, 0.9, 1.0])
def test_loss_mask_decided_without_division_is_the_same_mask(gpu, thr):
    """Without iou_max the ignore mask is decided by a margin test with the division as a
    fallback: the no-object sums (same summation order) must be the very same bits."""
    case = cases.v3()
    heads = to_dev(case.heads(91, 3), gpu)
    gt = case.gt(92, 3, 60, 64)
    gt[0, :5, 2:4] = 0.0                                     # zero-area boxes
    gt[1, 0, :4] = (1e30, 1e30, 1e30, 1e30)                  # products overflow: non-finite IoU
    dgt = torch.from_numpy(gt).to(gpu)
    hd = case.head_obj()
    exact = hd.loss(heads, dgt, 1., 1., 5., 1., thr, True, want_parts=True, want_iou_max=True)
    fast = hd.loss(heads, dgt, 1., 1., 5., 1., thr, True, want_parts=True)
    a, b = exact['parts'].cpu().numpy(), fast['parts'].cpu().numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(exact['per_image'].cpu().numpy().view(np.uint32), fast['per_image'].cpu().numpy().view(np.uint32))
    # the mask is not trivial at this threshold
    m = exact['iou_max'].cpu().numpy()
    assert 0 < np.count_nonzero(m > thr) or thr >= 0.9


@pytest.mark.parametrize('version,B,G', [(3, 3, 40), (3, 2, 1), (2, 4, 12)])
def test_loss_backward_matches_oracle(gpu, version, B, G):
    """Gradient w.r.t. the head tensors through the autograd Function (yh_loss_forward +
    yh_loss_backward) against the closed-form NumPy restatement (float64), 1e-5 relative."""
    case = cases.v3(size=224, C=20) if version == 3 else cases.v2(size=224, C=20)
    heads = case.heads(95, B, obj_mean=-1.0)
    gt = case.gt(96, B, G, G + 3)
    if G > 2:
        gt[0, 1, :2] = gt[0, 0, :2]                          # two GT rows share a cell
    dheads = [h.requires_grad_(True) for h in to_dev(heads, gpu)]
    thr, norm = (0.5, True) if version == 3 else (0.6, False)
    hd = case.head_obj()
    total, per = hd.loss_with_grad(dheads, torch.from_numpy(gt).to(gpu), 1., 1., 5., 1., thr, norm)
    weights = torch.linspace(0.5, 1.5, B, device=gpu)
    ((total / B).sum() * 0.5 + (per * weights).sum()).backward()    # uses both outputs
    for b in range(B):
        hi = [h[b] for h in heads]
        if version == 3:
            want = O.loss_v3_grad_image(hi, gt[b], case.priors, case.C)
        else:
            want = [O.loss_v2_grad_image(hi[0], gt[b], case.priors[0], case.C)]
        scale = 0.5 / B + float(weights[b])
        for s, w in enumerate(want):
            got = dheads[s].grad[b].cpu().numpy()
            assert np.count_nonzero(w) > 0
            assert_close(got, w * scale, rtol=RTOL, atol=1e-7, what='grad image %d scale %d' % (b, s))
            assert np.array_equal(got == 0, w == 0) or np.count_nonzero((got == 0) != (w == 0)) <= 2


def test_gradient_of_a_box_assigned_to_several_rows_is_reproducible(gpu):
    """Five ground-truth rows of one image with the same centre and nearly the same size are assigned
    the same box: the box's gradient is the sum of five terms, added in the order of the rows
    whichever workgroup runs first.  The fused pass and the two-call path give the same bits, call
    after call (atomic additions gave two or three different roundings), and the closed-form
    float64 restatement agrees to 1e-5."""
    case = cases.v3(size=224, C=20)
    B, G, pad = 3, 40, 43
    heads = case.heads(195, B, obj_mean=-1.0)
    gt = case.gt(196, B, G, pad)
    for j, f in ((7, 1.02), (18, 0.97), (19, 1.04), (33, 0.99)):      # rows of several workgroups (16 rows each)
        gt[1, j, :2] = gt[1, 2, :2]
        gt[1, j, 2:4] = gt[1, 2, 2:4] * f
    dheads = to_dev(heads, gpu)
    dgt = torch.from_numpy(gt).to(gpu)
    hd = case.head_obj()
    gw = torch.linspace(0.5, 1.5, B, device=gpu)
    fwd = hd.loss(dheads, dgt, 1., 1., 5., 1., 0.5, True, want_mask=True)
    first = None
    for rep in range(6):
        fused = hd.loss(dheads, dgt, 1., 1., 5., 1., 0.5, True, grad_per_image=gw)['grads']
        two = hd.loss_backward(dheads, dgt, 1., 1., 5., 1., True, fwd['noobj_mask'], gw)
        if first is None:
            first = [g.clone() for g in fused]
        for s in range(len(first)):
            assert torch.equal(fused[s], first[s]) and torch.equal(two[s], first[s]), (rep, s)
    want = O.loss_v3_grad_image([h[1] for h in heads], gt[1], case.priors, case.C)
    for s, w in enumerate(want):
        assert_close(first[s][1].cpu().numpy(), w * float(gw[1]), rtol=RTOL, atol=1e-7, what='grad scale %d' % s)
    # the five rows did land on one box: fewer boxes carry a size gradient than the image has rows
    assigned = sum(int((g[1].reshape(-1, case.C + 5)[:, case.C + 2] != 0).sum()) for g in first)
    assert 0 < assigned <= G - 4, assigned


def test_loss_and_grad_equals_the_autograd_path(gpu):
    """loss_and_grad (the fused library call) gives the bits torch.autograd gets from
    loss_with_grad (forward, then backward) for the same upstream factor."""
    case = cases.v3(size=96)
    B = 3
    heads = to_dev(case.heads(91, B, obj_mean=-1.0), gpu)
    gt = torch.from_numpy(y.synth.make_ground_truth(92, B, 5, 9, 96, case.C)).to(gpu)
    hd = case.head_obj()
    out, grads = hd.loss_and_grad(heads, gt, 1., 1., 5., 1., 0.5, True, grad_scale=0.5 / B)
    leaves = [h.clone().requires_grad_(True) for h in heads]
    total, per = hd.loss_with_grad(leaves, gt, 1., 1., 5., 1., 0.5, True)
    (0.5 * total.reshape(()) / B).backward()
    assert torch.equal(out['per_image'], per.detach()) and torch.equal(out['sum'], total.detach())
    for g, l in zip(grads, leaves):
        assert torch.equal(g, l.grad)


@pytest.mark.parametrize('version,size,B,G,pad,exact', [
    ('v3', 416, 8, 60, 100, False), ('v3', 96, 3, 0, 3, False), ('v3', 160, 2, 5, 30, True),
    ('v2', 416, 4, 1, 20, False), ('v3', 608, 2, 99, 100, False), ('v3', 160, 40, 7, 12, False),
    ('v3', 96, 150, 3, 5, False)])
def test_fused_loss_and_gradient_equal_the_two_calls(gpu, version, size, B, G, pad, exact):
    """yh_loss_with_grad (gradient written during the forward pass) against yh_loss_forward
    followed by yh_loss_backward: every output bit for bit, with few and many ground-truth
    rows (zeros written inside and after the ignore-mask loop), with the IoU maxima, v2 and v3."""
    case = cases.v3(size=size) if version == 'v3' else cases.v2()
    pix = size if version == 'v3' else 416
    norm, thr = (True, 0.5) if version == 'v3' else (False, 0.6)
    heads = y.synth.torch_heads(gpu, 401, B, case.grids, case.P, case.C, obj_mean=-1.0)
    gt = torch.from_numpy(y.synth.make_ground_truth(402, B, G, pad, pix, case.C)).to(gpu)
    hd = case.head_obj()
    gw = torch.linspace(0.25, 1.5, B, device=gpu)
    two = hd.loss(heads, gt, 1., 0.5, 5., 2., thr, norm, want_parts=True, want_iou_max=exact, want_mask=True)
    want = hd.loss_backward(heads, gt, 1., 0.5, 5., 2., norm, two['noobj_mask'], gw)
    poison = [torch.full_like(w, float('nan')) for w in want]   # freed blocks the allocator hands out next:
    del poison                                                  # the fused pass must overwrite all of them
    one = hd.loss(heads, gt, 1., 0.5, 5., 2., thr, norm, want_parts=True, want_iou_max=exact, grad_per_image=gw)
    def same(a, b):                                  # NaN where the reference divides by a GT count of zero
        return a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=7e30, neginf=-7e30),
                                                  torch.nan_to_num(b, nan=-7.0, posinf=7e30, neginf=-7e30))
    for k in ('per_image', 'sum', 'mean', 'parts') + (('iou_max',) if exact else ()):
        assert same(one[k], two[k]), k
    for s, (g, w) in enumerate(zip(one['grads'], want)):
        assert same(g, w), s
    assert sum(int((g != 0).sum()) for g in one['grads']) > 0


def test_loss_finalize_tickets_survive_their_ring_wrapping_around(gpu):
    """The forward call's last workgroup finds out that it is the last through a ticket cell of a
    library-owned ring (4096 cells, one per call in flight, put back to zero by the workgroup that
    finalizes).  More calls than the ring has cells, alternating two shapes and two streams: every
    call's sum must be what the first call of its shape gave, and a poisoned output must be
    overwritten every time (a call whose last workgroup did not notice leaves it untouched)."""
    case = cases.v3(size=96)
    hd = [case.head_obj(), case.head_obj()]                  # a workspace each: the calls overlap on two streams
    streams = [torch.cuda.Stream(gpu), torch.cuda.Stream(gpu)]
    data, ref, outs = [], [], []
    for k, (B, G, pad) in enumerate(((2, 3, 5), (5, 1, 9))):
        heads = y.synth.torch_heads(gpu, 880 + k, B, case.grids, case.P, case.C, obj_mean=-1.0)
        gt = torch.from_numpy(y.synth.make_ground_truth(890 + k, B, G, pad, 96, case.C)).to(gpu)
        data.append((heads, gt))
        out = hd[k].loss(heads, gt, 1., 1., 5., 1., 0.5, True)
        ref.append(out['sum'].clone())
        outs.append(out)
    torch.cuda.synchronize(gpu)
    for i in range(4400):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            outs[k]['sum'].fill_(float('nan'))
            outs[k] = hd[k].loss(*data[k], 1., 1., 5., 1., 0.5, True, out=outs[k])
        if i % 400 >= 398 or i >= 4390:
            streams[k].synchronize()
            assert torch.equal(outs[k]['sum'], ref[k]), i
    torch.cuda.synchronize(gpu)
    for k in (0, 1):
        assert torch.equal(outs[k]['sum'], ref[k])


def test_loss_workspace_reused_with_new_ground_truth(gpu, c_oracle):
    """The ground-truth table lives in the workspace and is read back through the scalar data
    cache: calls in a row on ONE workspace with different ground truth (a training loop) must each
    see their own table, not a cached line of the call before.  Ignore mask bit for bit and the
    per-image losses against a fresh head object (fresh workspace) and the C oracle."""
    case = cases.v3(size=160)
    B, pad = 6, 24
    heads = y.synth.torch_heads(gpu, 930, B, case.grids, case.P, case.C, obj_mean=-1.0)
    hd = case.head_obj()
    gts = [y.synth.make_ground_truth(940 + rep, B, (3, 23, 1, 17, 8, 23, 2, 11)[rep], pad, 160, case.C) for rep in range(8)]
    dev_gts = [torch.from_numpy(g).to(gpu) for g in gts]
    torch.cuda.synchronize(gpu)
    # back to back, nothing between the launches but the stream's order
    gots = [hd.loss(heads, g, 1., 1., 5., 1., 0.5, True, want_mask=True) for g in dev_gts]
    for rep, (g, got) in enumerate(zip(dev_gts, gots)):
        fresh = case.head_obj().loss(heads, g, 1., 1., 5., 1., 0.5, True, want_mask=True)
        assert torch.equal(got['noobj_mask'], fresh['noobj_mask']), rep
        assert torch.equal(got['per_image'], fresh['per_image']), rep
        if rep in (0, 5):
            per, _, _, _ = c_oracle.loss(case.oracle_cfg(c_oracle), [h.cpu().numpy() for h in heads], gts[rep],
                                         (1., 1., 5., 1.), 0.5, True)
            assert_close(got['per_image'].cpu().numpy(), per, what='per-image loss')


def _same_bits(a, b):
    return a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=7e30, neginf=-7e30),
                                              torch.nan_to_num(b, nan=-7.0, posinf=7e30, neginf=-7e30))


@pytest.mark.parametrize('version,size,B,G,pad,exact', [
    ('v3', 416, 8, 60, 100, False), ('v3', 96, 3, 0, 3, False), ('v3', 160, 2, 5, 30, True),
    ('v2', 416, 4, 1, 20, False), ('v3', 608, 2, 99, 100, True), ('v3', 160, 40, 7, 12, False),
    ('v2', 480, 3, 30, 31, True)])
def test_prepared_loss_equals_the_unprepared_call(gpu, version, size, B, G, pad, exact):
    """yh_loss_prepare builds the ground-truth tables (pair table, cell occupancy, per-image
    counts) ahead of the loss call; the call that finds them must give every output bit of
    the call that builds them itself: forward with mask and IoU maxima, and the fused
    gradient pass.  Two rows of one image share a cell (the later row wins the cell)."""
    case = cases.v3(size=size) if version == 'v3' else cases.v2(size=size)
    norm, thr = (True, 0.5) if version == 'v3' else (False, 0.6)
    heads = y.synth.torch_heads(gpu, 411, B, case.grids, case.P, case.C, obj_mean=-1.0)
    g = y.synth.make_ground_truth(412, B, G, pad, size, case.C)
    if G > 2:
        g[0, 1, :2] = g[0, 0, :2]
    gt = torch.from_numpy(g).to(gpu)
    gw = torch.linspace(0.25, 1.5, B, device=gpu)
    plain = case.head_obj()
    want = plain.loss(heads, gt, 1., 0.5, 5., 2., thr, norm, want_parts=True, want_iou_max=exact, want_mask=True)
    want_g = plain.loss(heads, gt, 1., 0.5, 5., 2., thr, norm, want_parts=True, want_iou_max=exact, grad_per_image=gw)
    hd = case.head_obj()
    hd.loss_prepare(heads, gt)
    got = hd.loss(heads, gt, 1., 0.5, 5., 2., thr, norm, want_parts=True, want_iou_max=exact, want_mask=True)
    hd.loss_prepare([tuple(h.shape) for h in heads], gt)        # shapes alone are enough: the heads do not exist yet
    got_g = hd.loss(heads, gt, 1., 0.5, 5., 2., thr, norm, want_parts=True, want_iou_max=exact, grad_per_image=gw)
    for k in ('per_image', 'sum', 'mean', 'parts', 'noobj_mask') + (('iou_max',) if exact else ()):
        assert _same_bits(got[k], want[k]), k
    for k in ('per_image', 'sum', 'mean', 'parts') + (('iou_max',) if exact else ()):
        assert _same_bits(got_g[k], want_g[k]), k
    for s, (a, b) in enumerate(zip(got_g['grads'], want_g['grads'])):
        assert _same_bits(a, b), s
    if G > 0:
        assert sum(int((a != 0).sum()) for a in got_g['grads']) > 0


def test_a_preparation_serves_one_call_and_only_its_own_ground_truth(gpu):
    """A preparation is consumed by the next loss call on its workspace, and only when that
    call names the same ground-truth tensor, batch and padding: a call with OTHER ground
    truth after a preparation, a second call after one preparation, and a preparation
    followed by a different batch's call all give the unprepared results (a stale table
    would give the earlier batch's mask)."""
    case = cases.v3(size=160)
    B, pad = 6, 24
    heads = y.synth.torch_heads(gpu, 931, B, case.grids, case.P, case.C, obj_mean=-1.0)
    gts = [torch.from_numpy(y.synth.make_ground_truth(950 + r, B, (3, 23, 17)[r], pad, 160, case.C)).to(gpu) for r in range(3)]
    fresh = [case.head_obj().loss(heads, g, 1., 1., 5., 1., 0.5, True, want_mask=True) for g in gts]
    assert not torch.equal(fresh[0]['noobj_mask'], fresh[1]['noobj_mask'])
    hd = case.head_obj()
    hd.loss_prepare(heads, gts[0])
    other = hd.loss(heads, gts[1], 1., 1., 5., 1., 0.5, True, want_mask=True)      # prepared for gts[0]: not used
    hd.loss_prepare(heads, gts[2])
    first = hd.loss(heads, gts[2], 1., 1., 5., 1., 0.5, True, want_mask=True)      # prepared
    again = hd.loss(heads, gts[0], 1., 1., 5., 1., 0.5, True, want_mask=True)      # the preparation is spent
    # a training loop: prepare the next batch's tables right after this batch's loss call
    loop = []
    hd.loss_prepare(heads, gts[0])
    for r in range(3):
        loop.append(hd.loss(heads, gts[r], 1., 1., 5., 1., 0.5, True, want_mask=True))
        if r + 1 < 3:
            hd.loss_prepare(heads, gts[r + 1])
    for name, got, want in (('other', other, fresh[1]), ('first', first, fresh[2]), ('again', again, fresh[0]),
                            ('loop0', loop[0], fresh[0]), ('loop1', loop[1], fresh[1]), ('loop2', loop[2], fresh[2])):
        assert torch.equal(got['noobj_mask'], want['noobj_mask']), name
        assert torch.equal(got['per_image'], want['per_image']), name
    half = [h[:3].contiguous() for h in heads]                  # another batch size: another workspace, no preparation
    hd.loss_prepare(heads, gts[0])
    small = hd.loss(half, gts[0][:3].contiguous(), 1., 1., 5., 1., 0.5, True, want_mask=True)
    assert torch.equal(small['per_image'], fresh[0]['per_image'][:3])


def test_loss_outputs_can_be_reused(gpu):
    """loss(out=previous result) writes the same tensors again and gives the fresh call's values."""
    case = cases.v3(size=96)
    hd = case.head_obj()
    gt = torch.from_numpy(y.synth.make_ground_truth(95, 2, 4, 7, 96, case.C)).to(gpu)
    a = to_dev(case.heads(93, 2), gpu)
    b = to_dev(case.heads(94, 2), gpu)
    first = hd.loss(a, gt, 1., 1., 5., 1., 0.5, True, want_mask=True)
    keep = {k: v.clone() for k, v in hd.loss(b, gt, 1., 1., 5., 1., 0.5, True, want_mask=True).items() if v is not None}
    again = hd.loss(b, gt, 1., 1., 5., 1., 0.5, True, want_mask=True, out=first)
    assert again is first and again['per_image'].data_ptr() == first['per_image'].data_ptr()
    for k in ('per_image', 'sum', 'mean', 'noobj_mask'):
        assert torch.equal(again[k], keep[k])
    other = hd.loss(b, gt, 1., 1., 5., 1., 0.5, True, want_parts=True, out=first)       # different options: fresh tensors
    assert other is not first and other['parts'] is not None


def test_loss_backward_is_the_derivative_of_the_forward(gpu):
    """Directional finite difference of the GPU loss itself (fp32 forward evaluated at
    +-eps along a random direction) against <grad, direction>: 1e-3, the resolution of
    an fp32 difference quotient."""
    case = cases.v3(size=160, C=10)
    B = 2
    heads = to_dev(case.heads(97, B, obj_mean=-1.0), gpu)
    gt = torch.from_numpy(case.gt(98, B, 8, 10)).to(gpu)
    hd = case.head_obj()
    hs = [h.clone().requires_grad_(True) for h in heads]
    total, _ = hd.loss_with_grad(hs, gt, 1., 1., 5., 1., 0.5, True)
    total.sum().backward()
    gen = torch.Generator(device='cpu').manual_seed(3)
    dirs = [torch.randn(h.shape, generator=gen).to(gpu) for h in heads]
    analytic = sum(float((h.grad.double() * d.double()).sum()) for h, d in zip(hs, dirs))
    eps = 1e-2
    def f(sign):
        out = hd.loss([h + sign * eps * d for h, d in zip(heads, dirs)], gt, 1., 1., 5., 1., 0.5, True)
        return float(out['per_image'].double().sum())
    numeric = (f(+1) - f(-1)) / (2 * eps)
    assert abs(numeric - analytic) <= 1e-3 * max(1.0, abs(analytic)), (numeric, analytic)


def test_loss_v2_matches_oracle(gpu, c_oracle):
    case = cases.v2(size=480)
    B, G = 4, 9
    heads = case.heads(91, B)
    gt = case.gt(92, B, G, 60)
    out = case.head_obj().loss(to_dev(heads, gpu), torch.from_numpy(gt).to(gpu), 1., 1., 5., 1., 0.6, False,
                               want_parts=True, want_iou_max=True)
    per, parts, iou, mean = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, (1., 1., 5., 1.), 0.6, False, want_iou=True)
    assert np.array_equal(out['iou_max'].cpu().numpy().view(np.uint32), iou.view(np.uint32))
    assert_close(out['parts'].cpu().numpy(), parts, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, what='per-image loss')
    n0 = O.loss_v2_image(heads[0][0], gt[0], case.priors[0], case.C)
    assert_close(out['per_image'][0].item(), n0, what='loss vs numpy')


def test_loss_scales_and_ragged_gt(gpu, c_oracle):
    """Different GT counts per image and non-default scales."""
    case = cases.v3(size=320, C=20)
    B = 5
    heads = case.heads(101, B)
    gt = case.gt(102, B, 12, 20)
    for b, g in enumerate([12, 1, 5, 9, 3]):
        gt[b, g:] = -1.0
    scales = (2.0, 0.5, 3.0, 1.5)
    out = case.head_obj().loss(to_dev(heads, gpu), torch.from_numpy(gt).to(gpu), *scales, 0.5, True, want_parts=True)
    per, parts, _, _ = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, scales, 0.5, True)
    assert_close(out['parts'].cpu().numpy(), parts, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, what='per-image loss')


def test_loss_full_size_properties(gpu):
    """Config-3 shape at batch 32 (one GPU's share of 256): per-image values do
    not depend on the batch they are computed in, and the sum is their sum."""
    case = cases.v3()
    B = 32
    heads = y.synth.torch_heads(gpu, 7, B, case.grids, case.P, case.C)
    gt = torch.from_numpy(case.gt(8, B, 100, 101)).to(gpu)
    hd = case.head_obj()
    full = hd.loss(heads, gt, 1., 1., 5., 1., 0.5, True)
    half = hd.loss([h[16:] for h in heads], gt[16:], 1., 1., 5., 1., 0.5, True)
    assert torch.equal(full['per_image'][16:], half['per_image'])
    assert_close(full['sum'].item(), full['per_image'].double().sum().item(), rtol=1e-6, what='sum')
    assert torch.isfinite(full['per_image']).all()


def test_class_api_v3(gpu, c_oracle, capsys):
    case = cases.v3(C=20)
    cfg = dict(mode='test', data_shape=[416, 416, 3], num_classes=20, weight_decay=5e-4, keep_prob=0.5,
               data_format='channels_last', batch_size=8, coord_scale=1, noobj_scale=1, obj_scale=5.,
               class_scale=1., num_priors=3, nms_score_threshold=0.5, nms_max_boxes=10,
               nms_iou_threshold=0.5, priors=y.synth.V3_PRIORS)
    heads = case.heads(111, 2, obj_mean=-2.0)
    net = y.YOLOv3(cfg, None)
    pred = net.test_one_image(to_dev(heads, gpu))
    want = c_oracle.detect(case.oracle_cfg(c_oracle), [h[:1] for h in heads], 0.5, 0.5, 10)[0]
    assert pred[0].dtype == np.float32 and pred[1].dtype == np.float32 and pred[2].dtype == np.int32
    assert pred[1].shape == (len(pred[0]), 4)
    assert np.array_equal(pred[2], want[2]) and len(pred[0]) == len(want[0]) > 0
    assert_close(pred[0], want[0], what='scores')
    # channels_first input is transposed like YOLOv3.py:88-91
    cfg_cf = dict(cfg, data_format='channels_first')
    pred_cf = y.YOLOv3(cfg_cf, None).test_one_image([h.permute(0, 3, 1, 2).contiguous() for h in to_dev(heads, gpu)])
    assert all(np.array_equal(a, b) for a, b in zip(pred, pred_cf))

    # train mode: num_train // batch_size iterations, progress line, mean of .5 * total_loss
    Bt = 4
    batches = [(to_dev(case.heads(120 + i, Bt), gpu), torch.from_numpy(case.gt(130 + i, Bt, 6, 60)).to(gpu))
               for i in range(3)]
    provider = dict(data_shape=[416, 416, 3], num_train=3 * Bt + 1, num_val=0,
                    train_generator=(None, iter(batches)), val_generator=None)
    net = y.YOLOv3(dict(cfg, mode='train', batch_size=Bt), provider)
    capsys.readouterr()
    mean_loss = net.train_one_epoch(1e-3)
    text = capsys.readouterr().out
    assert '\r>> iters 0/3 loss ' in text and '\r>> iters 2/3 loss ' in text
    wants = []
    for hs, g in batches:
        _, _, _, m = c_oracle.loss(case.oracle_cfg(c_oracle), [h.cpu().numpy() for h in hs], g.cpu().numpy(),
                                   (1., 1., 5., 1.), 0.5, True)
        wants.append(np.float32(0.5) * m)
    assert_close(mean_loss, np.mean(wants), what='epoch mean loss')


def test_train_one_epoch_trains_through_the_fused_loss(gpu, capsys):
    """A head provider whose outputs carry an autograd graph gets gradients from
    yh_loss_backward and its step(lr) is called (optimizer.minimize, YOLOv3.py:312-318):
    the loss of a fixed batch goes down epoch after epoch."""
    case = cases.v3(size=160, C=5)
    Bt = 4

    class Provider:                                            # the smallest "network": a learnable bias on fixed features
        def __init__(self):
            self.base = to_dev(case.heads(140, Bt, obj_mean=-1.0), gpu)
            self.bias = [torch.zeros_like(b[:1]).requires_grad_(True) for b in self.base]
            self.steps = 0

        def __call__(self, images):
            return [b + w for b, w in zip(self.base, self.bias)]

        def zero_grad(self):
            for w in self.bias:
                w.grad = None

        def step(self, lr):
            with torch.no_grad():
                for w in self.bias:
                    w -= lr * w.grad
            self.steps += 1

    prov = Provider()
    gt = torch.from_numpy(case.gt(141, Bt, 5, 8)).to(gpu)

    def batches():
        while True:
            yield torch.zeros((Bt, 160, 160, 3), device=gpu), gt
    cfg = dict(mode='train', data_shape=[160, 160, 3], num_classes=5, weight_decay=0.0, keep_prob=0.5,
               data_format='channels_last', batch_size=Bt, coord_scale=1, noobj_scale=1, obj_scale=5.,
               class_scale=1., num_priors=3, nms_score_threshold=0.5, nms_max_boxes=10,
               nms_iou_threshold=0.5, priors=y.synth.V3_PRIORS, head_provider=prov)
    data = dict(data_shape=[160, 160, 3], num_train=5 * Bt, num_val=0, train_generator=(None, batches()),
                val_generator=None)
    net = y.YOLOv3(cfg, data)
    losses = [float(net.train_one_epoch(0.05)) for _ in range(4)]
    capsys.readouterr()
    assert prov.steps == 20
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
    assert losses[-1] < 0.99 * losses[0], losses


def test_class_api_v2(gpu, c_oracle, capsys):
    case = cases.v2(size=416)
    cfg = dict(mode='test', is_pretraining=False, data_shape=[416, 416, 3], num_classes=20, weight_decay=1e-4,
               keep_prob=0.5, data_format='channels_last', batch_size=4, coord_scale=1, noobj_scale=1,
               obj_scale=5., class_scale=1., nms_score_threshold=0.5, nms_max_boxes=10, nms_iou_threshold=0.5,
               rescore_confidence=False, priors=y.synth.V2_PRIORS)
    heads = case.heads(141, 1, obj_mean=0.0)
    pred = y.YOLOv2(cfg, None).test_one_image(to_dev(heads, gpu)[0])
    want = c_oracle.detect(case.oracle_cfg(c_oracle), heads, 0.5, 0.5, 10)[0]
    assert np.array_equal(pred[2], want[2]) and len(pred[0]) == len(want[0]) > 0
    assert_close(pred[0], want[0], what='scores')
    assert_close(pred[1], want[1], rtol=RTOL, atol=RTOL * 416, what='bbox')

    Bt = 4
    batches = [(to_dev(case.heads(150 + i, Bt), gpu)[0], torch.from_numpy(case.gt(160 + i, Bt, 5, 60)).to(gpu))
               for i in range(2)]
    provider = dict(data_shape=[416, 416, 3], num_train=2 * Bt, num_val=0,
                    train_generator=(None, iter(batches)), val_generator=None)
    net = y.YOLOv2(dict(cfg, mode='train', batch_size=Bt), provider)
    capsys.readouterr()
    mean_loss = net.train_one_epoch(5e-3)
    assert '\r>> iters 1/2 loss ' in capsys.readouterr().out
    wants = [c_oracle.loss(case.oracle_cfg(c_oracle), [h.cpu().numpy()], g.cpu().numpy(), (1., 1., 5., 1.), 0.6, False)[3]
             for h, g in batches]
    assert_close(mean_loss, np.mean(wants), what='epoch mean loss')


def test_host_threads_call_the_library_at_once(gpu):
    """Three host threads, each with a stream, head objects and inputs of its own, call detect (default
    and dense capacity), the loss with its preparation and the fused gradient at the same time (ctypes
    releases the interpreter lock during a call): the library's host-side tables (prepared losses,
    streaming calls per workspace, LDS limits, tickets) are shared by them; every result is the one the
    thread's inputs give alone."""
    import threading
    case = cases.v3(size=160, C=20)
    B, G, pad = 3, 6, 9
    inputs, want = [], []
    for t in range(3):
        heads = y.synth.torch_heads(gpu, 700 + t, B, case.grids, case.P, case.C, obj_mean=-1.0)
        gt = torch.from_numpy(y.synth.make_ground_truth(710 + t, B, G, pad, 160, case.C)).to(gpu)
        hd = case.head_obj()
        d = hd.detect(heads, 0.3, 0.5, 10)
        dense = hd.detect(heads, 0.05, 0.5, 10, max_candidates=4096)
        l = hd.loss(heads, gt, 1., 1., 5., 1., 0.5, True, grad_per_image=torch.ones(B, device=gpu))
        inputs.append((heads, gt))
        want.append((d.per_image(), dense.per_image(), l['per_image'].clone(), [g.clone() for g in l['grads']]))
    torch.cuda.synchronize(gpu)
    errors = []

    def work(t):
        try:
            heads, gt = inputs[t]
            stream = torch.cuda.Stream(gpu)
            with torch.cuda.stream(stream):
                hd = case.head_obj()
                gw = torch.ones(B, device=gpu)
                for it in range(150):
                    d = hd.detect(heads, 0.3, 0.5, 10)
                    dense = hd.detect(heads, 0.05, 0.5, 10, max_candidates=4096)
                    if it % 2:
                        hd.loss_prepare(heads, gt)
                    l = hd.loss(heads, gt, 1., 1., 5., 1., 0.5, True, grad_per_image=gw)
                    if it % 25 == 24:
                        stream.synchronize()
                        cases.same_detections(d.per_image(), want[t][0])
                        cases.same_detections(dense.per_image(), want[t][1])
                        assert torch.equal(l['per_image'], want[t][2]), it
                        assert all(torch.equal(a, b) for a, b in zip(l['grads'], want[t][3])), it
                stream.synchronize()
        except BaseException as e:                              # reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_abi_rejects_bad_arguments(gpu):
    case = cases.v2()
    hd = case.head_obj()
    heads = to_dev(case.heads(171, 1), gpu)
    with pytest.raises(y._abi.YhError) as e:
        hd.detect(heads, 0.5, 1.5, 10)                     # iou threshold outside [0, 1], as TF rejects
    assert e.value.code == y._abi.YH_ERR_INVALID_ARG
    with pytest.raises(y._abi.YhError) as e:
        hd.detect(heads, 0.5, 0.5, 100000)
    assert e.value.code in (y._abi.YH_ERR_LIMIT, y._abi.YH_ERR_INVALID_ARG)
    with pytest.raises(RuntimeError):
        hd.detect([h.cpu() for h in heads], 0.5, 0.5, 10)  # no CPU path
    # the fused loss + gradient call: gradient tensors must exist and must not be the heads
    import ctypes
    lib = y._abi.load()
    gt = torch.from_numpy(y.synth.make_ground_truth(172, 1, 2, 4, 416, case.C)).to(gpu)
    cfg = hd.config_for(heads)
    nbytes = ctypes.c_size_t(0)
    y._abi.check(lib.yh_loss_workspace_bytes(ctypes.byref(cfg), 1, 4, ctypes.byref(nbytes)))
    ws = torch.zeros(max(nbytes.value, 256), dtype=torch.uint8, device=gpu)
    per, total, gw = torch.empty(1, device=gpu), torch.empty(2, device=gpu), torch.ones(1, device=gpu)
    scales = (ctypes.c_float * 4)(1., 1., 5., 1.)
    hp = y._abi.pointer_array([h.data_ptr() for h in heads])
    st = ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)

    def call(gw_ptr, grads):
        return lib.yh_loss_with_grad(ctypes.byref(cfg), hp, gt.data_ptr(), 1, 4, scales, 0.6, 0, ws.data_ptr(), ws.numel(),
                                     per.data_ptr(), None, None, total.data_ptr(), gw_ptr, grads, st)
    assert call(None, hp) == y._abi.YH_ERR_INVALID_ARG
    assert call(gw.data_ptr(), None) == y._abi.YH_ERR_INVALID_ARG
    assert call(gw.data_ptr(), hp) == y._abi.YH_ERR_INVALID_ARG            # gradient tensors alias the heads
    assert call(gw.data_ptr(), y._abi.pointer_array([0])) == y._abi.YH_ERR_INVALID_ARG
    good = [torch.empty_like(h) for h in heads]
    assert call(gw.data_ptr(), y._abi.pointer_array([g.data_ptr() for g in good])) == y._abi.YH_OK
    torch.cuda.synchronize()
    assert torch.isfinite(per).all() and int((good[0] != 0).sum()) > 0
