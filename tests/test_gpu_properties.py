"""Property tests (hypothesis): random head geometries, thresholds and ground truth, the HIP path
through the C ABI against the C oracle.  Bounded example counts and derandomised draws, so the GPU
run is short and repeats itself.  YH_PROPERTY_EXAMPLES=n in the environment turns the same properties
into a soak: n examples each, drawn afresh (tools/soak.sh)."""
import os

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, assume, event, given, settings, strategies as st

import yolo_tf_amd as y
from tests import cases
from tests.test_gpu_parity import to_dev, assert_close, RTOL

pytestmark = pytest.mark.gpu

if os.environ.get('YH_TEST_LIB'):                   # a soak of another build of the library (a file beside the shipped one)
    from yolo_tf_amd import _abi
    _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), os.environ['YH_TEST_LIB'])

SOAK = int(os.environ.get('YH_PROPERTY_EXAMPLES', '0'))
SETTINGS = dict(deadline=None, derandomize=SOAK == 0, database=None, print_blob=True, suppress_health_check=list(HealthCheck))


MAX_GRID = 7 if SOAK == 0 else int(os.environ.get('YH_PROPERTY_GRID', '26'))     # the soak also draws grids of several workgroups
MAX_PAD = 40 if SOAK == 0 else 130
MAX_B = 3 if SOAK == 0 else 10


def examples(n):
    return SOAK or n


@st.composite
def geometries(draw, max_classes=300):
    S = draw(st.integers(1, 4))
    P = draw(st.sampled_from([1, 2, 3, 5, 9, 16]))
    C = draw(st.sampled_from([1, 2, 3, 20, 59, 64, 65, 80, 128, 129, max_classes]))
    grids = [(draw(st.integers(1, MAX_GRID)), draw(st.integers(1, MAX_GRID))) for _ in range(S)]
    assume(max(h * w for h, w in grids) * P * (C + 5) ** 2 < 2 ** 32)      # the library's documented bound per scale (YH_ERR_LIMIT beyond)
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.Generator(np.random.PCG64(seed))
    priors = [rng.uniform(0.3, 4.0, size=(P, 2)).astype(np.float32) for _ in range(S)]
    mult = tuple(float(draw(st.sampled_from([8.0, 16.0, 32.0]))) for _ in range(S))
    rate = tuple(float(draw(st.sampled_from([8.0, 16.0, 32.0]))) for _ in range(S))
    size = int(32 * max(max(g) for g in grids))
    return cases.Case('prop', grids, priors, mult, rate, C, size, 3), seed


@settings(max_examples=examples(30), **SETTINGS)
@given(geo=geometries(), B=st.integers(1, 3),
       thr=st.sampled_from([0.0, 1e-3, 0.05, 0.3, 0.5, 0.9, 1.0]), iou=st.sampled_from([0.0, 0.3, 0.5, 1.0]),
       mb=st.sampled_from([1, 3, 10, 100]), obj_mean=st.sampled_from([-3.0, 0.0, 3.0]))
def test_detect_stages_against_the_oracle(gpu, c_oracle, geo, B, thr, iou, mb, obj_mean):
    """decode 1e-5 against the oracle; score filter + NMS bit-exact on the GPU's own decoded
    arrays; the fused call (from the head tensors) equal to the staged one, boxes included."""
    case, seed = geo
    heads = case.heads(seed % 9973, B, obj_mean=obj_mean, cold_mean=-2.0)
    dheads = to_dev(heads, gpu)
    hd = case.head_obj()
    boxes, conf = hd.decode(dheads)
    wb, wc = c_oracle.decode(case.oracle_cfg(c_oracle), heads)
    assert_close(boxes.cpu().numpy(), wb, rtol=RTOL, atol=RTOL * case.size * 4, what='boxes')
    assert_close(conf.cpu().numpy(), wc, rtol=RTOL, atol=1e-30, what='conf')
    staged = hd.nms(boxes, conf, thr, iou, mb, want_class_candidates=True).resolved()
    hb, hc = boxes.cpu().numpy(), conf.cpu().numpy()
    want = c_oracle.nms(hb, hc, thr, iou, mb)
    got = staged.per_image()
    cases.same_detections(got, want)
    for g, w in zip(got, want):
        assert np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32))
        assert np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32))
    assert np.array_equal(staged.class_candidates.cpu().numpy(), (hc >= np.float32(thr)).sum(axis=1).astype(np.int32))
    fused = hd.detect(dheads, thr, iou, mb, want_class_candidates=True).resolved()
    cases.same_detections(fused.per_image(), got)
    for f, g in zip(fused.per_image(), got):
        assert np.array_equal(f[0], g[0]) and np.array_equal(f[1], g[1])
    assert torch.equal(fused.class_candidates, staged.class_candidates)


@st.composite
def ground_truths(draw, case, B):
    """[B, pad, 5] with duplicates, zero-area rows, centres outside the image and ragged counts."""
    pad = draw(st.integers(2, MAX_PAD))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.Generator(np.random.PCG64(seed))
    gt = np.full((B, pad, 5), -1.0, np.float32)
    for b in range(B):
        G = int(rng.integers(0 if SOAK else 1, pad))           # at least one pad row; the soak also draws images without any row
        yx = rng.uniform(-0.1 * case.size, 1.1 * case.size, size=(G, 2))
        hw = np.exp(rng.uniform(np.log(2.0), np.log(case.size), size=(G, 2)))
        kind = rng.integers(0, 6, size=G)
        hw[kind == 0] = 0.0                                    # zero-area box
        for g in np.nonzero(kind == 1)[0]:                     # duplicate of an earlier row
            j = int(rng.integers(0, g + 1))
            yx[g], hw[g] = yx[j], hw[j]
        gt[b, :G, 0:2] = np.maximum(yx, 0.0)                   # a negative yc would end the argmin slice early
        gt[b, :G, 2:4] = hw
        gt[b, :G, 4] = rng.integers(0, case.C + 2, size=G)     # labels beyond C: all-zero one-hot row
    return gt


@settings(max_examples=examples(25), **SETTINGS)
@given(data=st.data(), geo=geometries(max_classes=90), B=st.integers(1, MAX_B), version=st.sampled_from([2, 3]),
       thr=st.sampled_from([0.0, 0.3, 0.5, 0.6, 1.0]))
def test_loss_against_the_oracle(gpu, c_oracle, data, geo, B, version, thr):
    """Ignore-mask IoU maxima bit-identical, loss parts and per-image losses 1e-5, with and without
    the division-free mask, v2- and v3-style normalisation."""
    case, seed = geo
    gt = data.draw(ground_truths(case, B))
    heads = case.heads(seed % 7919, B, obj_mean=-1.0)
    norm = version == 3
    scales = (1.0, 0.5, 5.0, 2.0)
    hd = case.head_obj()
    dheads, dgt = to_dev(heads, gpu), torch.from_numpy(gt).to(gpu)
    out = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_iou_max=True)
    per, parts, iou, _ = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, scales, thr, norm, want_iou=True)
    assert np.array_equal(out['iou_max'].cpu().numpy().view(np.uint32), iou.view(np.uint32))
    assert_close(out['parts'].cpu().numpy(), parts, rtol=RTOL, atol=1e-6, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, rtol=RTOL, atol=1e-6, what='per-image loss')
    fast = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True)
    assert np.array_equal(fast['parts'].cpu().numpy().view(np.uint32), out['parts'].cpu().numpy().view(np.uint32))


@settings(max_examples=examples(20), **SETTINGS)
@given(data=st.data(), geo=geometries(max_classes=90), B=st.integers(1, MAX_B + 1), version=st.sampled_from([2, 3]),
       thr=st.sampled_from([0.0, 0.3, 0.5, 0.6, 1.0]), exact=st.booleans())
def test_loss_gradient_paths_agree(gpu, data, geo, B, version, thr, exact):
    """The three ways to the gradient give the same bits on random geometries and ground truth with
    duplicated rows (several rows assigned one box): forward then backward, the fused pass, and the
    fused pass on a prepared workspace; the fused pass twice gives the same bits twice; the
    forward outputs of all of them are those of the plain forward call."""
    case, seed = geo
    gt = data.draw(ground_truths(case, B))
    heads = case.heads(seed % 7907, B, obj_mean=-1.0)
    norm = version == 3
    scales = (1.0, 0.5, 5.0, 2.0)
    dheads, dgt = to_dev(heads, gpu), torch.from_numpy(gt).to(gpu)
    gw = torch.linspace(0.25, 1.5, B, device=gpu)
    hd = case.head_obj()
    fwd = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_iou_max=exact, want_mask=True)
    two = hd.loss_backward(dheads, dgt, *scales, norm, fwd['noobj_mask'], gw)
    fused = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_iou_max=exact, grad_per_image=gw)
    again = [g.clone() for g in hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_iou_max=exact,
                                        grad_per_image=gw)['grads']]
    other = case.head_obj()
    other.loss_prepare([tuple(h.shape) for h in dheads], dgt)
    prepared = other.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_iou_max=exact, grad_per_image=gw)
    other.loss_prepare(dheads, dgt)
    prepared_fwd = other.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_iou_max=exact, want_mask=True)

    def same(a, b):
        return a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=7e30, neginf=-7e30),
                                                  torch.nan_to_num(b, nan=-7.0, posinf=7e30, neginf=-7e30))
    for k in ('per_image', 'sum', 'parts') + (('iou_max',) if exact else ()):
        assert same(fused[k], fwd[k]) and same(prepared[k], fwd[k]) and same(prepared_fwd[k], fwd[k]), k
    assert torch.equal(prepared_fwd['noobj_mask'], fwd['noobj_mask'])
    for s in range(len(two)):
        assert same(fused['grads'][s], two[s]), ('fused against two calls', s)
        assert same(again[s], two[s]), ('second fused call', s)
        assert same(prepared['grads'][s], two[s]), ('prepared', s)


@st.composite
def candidate_sets(draw):
    """Decoded boxes [B, N, 4] and confidences [B, N, C] whose class segments range from empty to
    several thousand candidates, with score distributions that make a radix selection's groups
    degenerate and box layouts from sparse to heavily overlapping."""
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.Generator(np.random.PCG64(seed))
    B = draw(st.integers(1, 3))
    C = draw(st.sampled_from([1, 2, 3, 5, 17, 80, 300, 1100]))
    N = draw(st.sampled_from([1, 7, 64, 65, 300, 1500] + ([2049, 4000, 6000] if SOAK else []))) + draw(st.integers(0, 50))
    N = max(1, min(N, 1500000 // (B * C)))
    layout = draw(st.sampled_from(['spread', 'clustered', 'nested', 'degenerate']))
    if layout == 'clustered':
        k = int(rng.integers(1, 12))
        centre = rng.uniform(60, 340, size=(B, k, 2))
        size = rng.uniform(30, 120, size=(B, k, 2))
        which = rng.integers(0, k, size=(B, N))
        c = np.take_along_axis(centre, which[..., None], axis=1) + rng.normal(0, 6, size=(B, N, 2))
        sz = np.take_along_axis(size, which[..., None], axis=1) * rng.uniform(0.8, 1.25, size=(B, N, 2))
    elif layout == 'nested':
        c = np.full((B, N, 2), 200.0) + rng.normal(0, 1, size=(B, N, 2))
        sz = rng.uniform(1, 400, size=(B, N, 2))
    else:
        c = rng.uniform(0, float(draw(st.sampled_from([300, 900, 3000]))), size=(B, N, 2))
        sz = rng.uniform(10, 160, size=(B, N, 2))
    boxes = np.concatenate([c - sz / 2, c + sz / 2], axis=-1).astype(np.float32)
    if layout == 'degenerate':
        kind = rng.integers(0, 5, size=(B, N))
        boxes[kind == 0, 2:] = boxes[kind == 0, :2]                      # zero area
        boxes[kind == 1] = boxes[kind == 1][:, [2, 3, 0, 1]]             # corners the other way round
        boxes[kind == 2] = boxes[0, 0]                                   # copies of one box
    u = rng.uniform(size=(B, N, C))
    dist = draw(st.sampled_from(['uniform', 'ties', 'one_value_band', 'log_uniform', 'two_bands', 'half_zero_bits']))
    if dist == 'ties':
        conf = rng.choice(np.array([0.3, 0.55, 0.5500001, 0.8, 0.95], np.float32), size=(B, N, C))
    elif dist == 'one_value_band':
        conf = (0.75 + 1e-5 * u).astype(np.float32)
    elif dist == 'log_uniform':
        conf = np.exp(np.log(2e-3) + u * (np.log(1.0) - np.log(2e-3))).astype(np.float32)
    elif dist == 'two_bands':
        conf = np.where(u < 0.1, 0.9 + 0.05 * u, 0.01 + 0.001 * u).astype(np.float32)
    elif dist == 'half_zero_bits':
        conf = (np.floor(u * 64) / 64 + 1 / 128).astype(np.float32)
    else:
        conf = u.astype(np.float32)
    conf = conf.astype(np.float32)
    conf[rng.uniform(size=conf.shape) < draw(st.sampled_from([0.0, 0.5, 0.97, 0.999]))] = 0.0
    return boxes, conf


@settings(max_examples=examples(12), **SETTINGS)
@given(cs=candidate_sets(), thr=st.sampled_from([0.001, 0.05, 0.3, 0.6]), iou=st.sampled_from([0.0, 0.3, 0.45, 0.5, 0.7, 1.0]),
       mb=st.sampled_from([1, 5, 20, 100, 200, 512]), tight=st.booleans())
def test_nms_stage_on_random_candidate_sets(gpu, c_oracle, cs, thr, iou, mb, tight):
    """Score filter + per-class NMS on given boxes and confidences against the oracle: kept rows,
    classes, scores and boxes bit for bit and the per-class candidate counts, whatever launch
    shape the candidate capacity selects (the default one, or room for every pair)."""
    boxes, conf = cs
    B, N, C = conf.shape
    longest = int((conf >= np.float32(thr)).sum(axis=1).max())
    event('longest class segment: %s' % ('0' if longest == 0 else '<= 64' if longest <= 64 else '<= 2048' if longest <= 2048 else '> 2048'))
    event('candidates of an image: %s' % ('<= 2048' if (conf >= np.float32(thr)).sum(axis=(1, 2)).max() <= 2048 else '> 2048'))
    hd = y.DetectionHead(C, [np.ones((1, 2), np.float32)], [1.0], [1.0])
    res = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), thr, iou, mb,
                 max_candidates=None if tight else N * C, want_class_candidates=True).resolved()
    want = c_oracle.nms(boxes, conf, thr, iou, mb)
    got = res.per_image()
    cases.same_detections(got, want)
    for g, w in zip(got, want):
        assert np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32))
        assert np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32))
        assert np.array_equal(g[3], w[3])
    got_cc, want_cc = res.class_candidates.cpu().numpy(), (conf >= np.float32(thr)).sum(axis=1).astype(np.int32)
    if os.environ.get('YH_PROPERTY_DUMP') and not np.array_equal(got_cc, want_cc):      # what a soak keeps of a failure
        again = hd.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(conf).to(gpu), thr, iou, mb,
                       max_candidates=None if tight else N * C, want_class_candidates=True).resolved()
        np.savez(os.path.join(os.environ['YH_PROPERTY_DUMP'], 'nms_cc_%d_%d_%d.npz' % (B, N, C)), boxes=boxes, conf=conf,
                 thr=thr, iou=iou, mb=mb, tight=tight, got=got_cc, want=want_cc, status=res.status.cpu().numpy(),
                 count=res.count.cpu().numpy(), again=again.class_candidates.cpu().numpy())
    assert np.array_equal(got_cc, want_cc), ('class candidates', got_cc[got_cc != want_cc][:8], want_cc[got_cc != want_cc][:8],
                                             np.argwhere(got_cc != want_cc)[:8].tolist(), B, N, C)


TUNINGS = [dict(), dict(row_steps=1), dict(row_steps=2), dict(row_steps=3), dict(row_steps=5), dict(row_steps=50),
           dict(detect_kernel=2, stream_tile_kib=1),
           dict(detect_kernel=2, stream_tile_kib=8, stream_buffers=3, stream_units=8192),
           dict(detect_kernel=2, stream_tile_kib=8, stream_buffers=2, stream_units=64),
           dict(detect_kernel=2, stream_tile_kib=4, stream_buffers=2, stream_units=300),
           dict(detect_kernel=2, stream_tile_kib=4, stream_buffers=3, stream_units=7),
           dict(detect_kernel=2, stream_tile_kib=4, stream_buffers=4, stream_units=100000)]


@settings(max_examples=examples(10), **SETTINGS)
@given(geo=geometries(), B=st.integers(1, 3), thr=st.sampled_from([1e-3, 0.05, 0.3, 0.5]), iou=st.sampled_from([0.3, 0.5]),
       mb=st.sampled_from([3, 10, 100]), obj_mean=st.sampled_from([-3.0, 0.0, 3.0]), hot_mean=st.sampled_from([-2.0, 2.0]),
       cap_kind=st.sampled_from(['default', 'one launch', 'every pair']))
def test_streaming_kernels_agree_on_random_heads(gpu, geo, B, thr, iou, mb, obj_mean, hot_mean, cap_kind):
    """The objectness-first kernel at several unit lengths, the workgroup-tile kernel and the
    wave-private kernel in several shapes give the same detections and class counts from the
    same head tensors, bit for bit, whatever the geometry: tensors whose float count is not a
    multiple of four, rows that straddle tiles, few and all rows surviving."""
    from yolo_tf_amd import _abi
    case, seed = geo
    dheads = to_dev(case.heads(seed % 9967, B, obj_mean=obj_mean, hot_mean=hot_mean, cold_mean=-2.0), gpu)
    cap = {'default': None, 'one launch': min(case.N * case.C, 2048), 'every pair': case.N * case.C}[cap_kind]
    try:
        ref = None
        for t in TUNINGS:
            _abi.set_tuning(**t)
            got = case.head_obj().detect(dheads, thr, iou, mb, max_candidates=cap, want_class_candidates=True).resolved()
            if ref is None:
                ref = got
                continue
            cases.same_detections(got.per_image(), ref.per_image())
            for a, b in zip(got.per_image(), ref.per_image()):
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), t
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), t
            assert torch.equal(got.class_candidates, ref.class_candidates), t
    finally:
        _abi.set_tuning()


@settings(max_examples=examples(10), **SETTINGS)
@given(geo=geometries(), B=st.integers(1, 3), batches=st.integers(2, 7), lanes=st.integers(1, 3),
       thr=st.sampled_from([0.05, 0.3, 0.5]), mb=st.sampled_from([3, 10, 100]), obj_mean=st.sampled_from([-3.0, 0.0, 3.0]),
       cap_kind=st.sampled_from(['default', 'one launch', 'every pair']), kernel=st.sampled_from([0, 1, 2]))
def test_lanes_match_detect_on_random_heads(gpu, geo, B, batches, lanes, thr, mb, obj_mean, cap_kind, kernel):
    """DetectLanes (pipelines on streams of their own, a batch's select stage inside or right before
    the next batch's launch of its lane) gives detect()'s result for every batch of a random
    sequence, in submission order, bit for bit; a batch whose candidates overflow the capacity is
    flagged as detect() flags it."""
    from yolo_tf_amd import _abi
    case, seed = geo
    cap = {'default': None, 'one launch': min(case.N * case.C, 2048), 'every pair': case.N * case.C}[cap_kind]
    try:
        _abi.set_tuning(detect_kernel=kernel)
        sets = [to_dev(case.heads((seed + 31 * i) % 9949, B, obj_mean=obj_mean, cold_mean=-2.0), gpu) for i in range(batches)]
        hd = case.head_obj()
        want = [hd.detect(h, thr, 0.5, mb, max_candidates=cap, want_class_candidates=True) for h in sets]
        want = [(w.overflowed(), w.per_image() if not w.overflowed() else None, w.class_candidates.clone()) for w in want]
        torch.cuda.synchronize()
        pipe = y.DetectLanes(case.head_obj(), thr, 0.5, mb, lanes=lanes, max_candidates=cap, want_class_candidates=True)
        def read(d):                                  # at once: a lane's output sets alternate, the second call after this one reuses them
            ov = d.overflowed()
            rows = [(s[:k], b[:k], c[:k], r[:k]) for s, b, c, r, k in zip(d.scores.cpu().numpy(), d.boxes.cpu().numpy(),
                    d.class_id.cpu().numpy(), d.index.cpu().numpy(), d.count.cpu().numpy())]
            return ov, rows, d.class_candidates.clone()
        got = []
        for h in sets:
            d = pipe.submit(h, wait_current=True)
            if d is not None:
                got.append(read(d))
        for d in pipe.flush():
            got.append(read(d))
        assert len(got) == len(want) and pipe.flush() == []
        for i, ((gov, gi, gc), (ov, w, wc)) in enumerate(zip(got, want)):
            assert gov == ov, i
            if ov:
                continue
            cases.same_detections(gi, w)
            for a, b in zip(gi, w):
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), i
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), i
            assert torch.equal(gc, wc), i
    finally:
        _abi.set_tuning()


@settings(max_examples=examples(10), **SETTINGS)
@given(data=st.data(), geo=geometries(max_classes=90), B=st.integers(1, MAX_B), version=st.sampled_from([2, 3]),
       thr=st.sampled_from([0.3, 0.5, 0.6]), steps=st.integers(2, 6))
def test_loss_call_sequences_on_one_workspace(gpu, data, geo, B, version, thr, steps):
    """A training loop's calls on ONE head object (one workspace, whose ground-truth tables are read back
    through the scalar data cache): every step has new ground truth, is prepared ahead or not, asks for
    the gradient or not; every step's outputs are those of a fresh head object given the same inputs,
    bit for bit (no table, count or prepared state of an earlier step is seen by a later one)."""
    case, seed = geo
    norm = version == 3
    scales = (1.0, 0.5, 5.0, 2.0)
    pad = data.draw(st.integers(2, MAX_PAD))
    dheads = to_dev(case.heads(seed % 7901, B, obj_mean=-1.0), gpu)
    gw = torch.linspace(0.25, 1.5, B, device=gpu)
    hd = case.head_obj()
    plan = []
    for i in range(steps):
        rng = np.random.Generator(np.random.PCG64(seed + 17 * i))
        gt = np.full((B, pad, 5), -1.0, np.float32)
        for b in range(B):
            G = int(rng.integers(1, pad))
            gt[b, :G, 0:2] = rng.uniform(0.0, case.size, size=(G, 2))
            gt[b, :G, 2:4] = np.exp(rng.uniform(np.log(2.0), np.log(case.size), size=(G, 2)))
            gt[b, :G, 4] = rng.integers(0, case.C, size=G)
        plan.append((torch.from_numpy(gt).to(gpu), data.draw(st.booleans()), data.draw(st.booleans()), data.draw(st.booleans())))
    got = []
    for dgt, prepare, grad, stale in plan:                      # back to back, nothing but the stream's order between the calls
        if prepare:
            if stale:                                           # a preparation that is never used: other ground truth follows
                hd.loss_prepare(dheads, plan[0][0])
            hd.loss_prepare(dheads, dgt)
        out = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_mask=not grad, grad_per_image=gw if grad else None)
        got.append({k: ([g.clone() for g in v] if k == 'grads' else v.clone()) for k, v in out.items() if v is not None})

    def same(a, b):
        return a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=7e30, neginf=-7e30),
                                                  torch.nan_to_num(b, nan=-7.0, posinf=7e30, neginf=-7e30))
    for i, ((dgt, prepare, grad, stale), g) in enumerate(zip(plan, got)):
        w = case.head_obj().loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_mask=not grad,
                                 grad_per_image=gw if grad else None)
        for k in ('per_image', 'sum', 'parts') + (('grads',) if grad else ('noobj_mask',)):
            if k == 'grads':
                assert all(same(a, b) for a, b in zip(g[k], w[k])), (i, k, prepare, stale)
            else:
                assert same(g[k], w[k]), (i, k, prepare, stale)


@settings(max_examples=examples(8), **SETTINGS)
@given(geos=st.lists(geometries(), min_size=2, max_size=5), data=st.data())
def test_static_scratch_on_random_call_sequences(gpu, geos, data):
    """yh_detect_static (the library's own scratch array instead of a workspace) called for a sequence of
    unrelated shapes, capacities and thresholds in a row: each call finds the leftovers of the one before
    where its counters lie, and gives yh_detect's outputs on a fresh workspace, bit for bit."""
    from yolo_tf_amd import _abi
    for case, seed in geos:
        B = data.draw(st.integers(1, 3))
        thr = data.draw(st.sampled_from([1e-3, 0.05, 0.3, 0.5]))
        mb = data.draw(st.sampled_from([3, 10, 100]))
        cap = data.draw(st.sampled_from([8, 512, 2048, 6000]))
        cap = max(1, min(cap, case.N * case.C))
        dheads = to_dev(case.heads(seed % 9923, B, obj_mean=data.draw(st.sampled_from([-3.0, 0.0, 3.0])), cold_mean=-2.0), gpu)
        hd = case.head_obj()
        want = hd.detect(dheads, thr, 0.5, mb, max_candidates=cap, want_class_candidates=True)
        outs = hd._alloc_outputs(B, mb, gpu, True)
        try:
            got = hd.detect_static(dheads, thr, 0.5, mb, cap, outs)
        except _abi.YhError as e:                                # a shape beyond the array: said so, nothing ran
            assert e.code == _abi.YH_ERR_LIMIT
            continue
        assert torch.equal(got.status, want.status)
        assert torch.equal(got.class_candidates, want.class_candidates) or want.overflowed()
        if want.overflowed():
            continue
        count = want.count.cpu().numpy()
        assert np.array_equal(got.count.cpu().numpy(), count)
        for b, k in enumerate(count):
            for a, w in ((got.scores, want.scores), (got.boxes, want.boxes), (got.class_id, want.class_id), (got.index, want.index)):
                assert np.array_equal(a[b, :k].cpu().numpy().view(np.uint32), w[b, :k].cpu().numpy().view(np.uint32)), (case.grids, b)


@settings(max_examples=examples(8), **SETTINGS)
@given(version=st.sampled_from([2, 3]), cells=st.integers(1, 8), C=st.sampled_from([1, 2, 20, 80]),
       fmt=st.sampled_from(['channels_last', 'channels_first']), Bt=st.integers(1, 5), iters=st.integers(1, 3),
       thr=st.sampled_from([0.05, 0.3, 0.5]), iou=st.sampled_from([0.3, 0.5]), mb=st.sampled_from([1, 10, 50]),
       w=st.tuples(st.sampled_from([1., 2.]), st.sampled_from([1., .5]), st.sampled_from([5., 1.]), st.sampled_from([1., 3.])),
       seed=st.integers(0, 10 ** 6))
def test_class_api_on_random_configs(gpu, c_oracle, version, cells, C, fmt, Bt, iters, thr, iou, mb, w, seed):
    """The YOLOv3 / YOLOv2 classes over random configuration dicts (image size, class count, data format,
    thresholds, loss scales, batch size): test_one_image gives, for image 0, the bits of the head path set up
    independently for the same geometry; train_one_epoch's mean over the iterations is the oracle's (.5 * total_loss for
    v3, total_loss for v2)."""
    import contextlib, io
    size = 32 * cells
    case = cases.v3(size=size, C=C) if version == 3 else cases.v2(size=size, C=C)
    cfg = dict(mode='test', data_shape=[size, size, 3], num_classes=C, weight_decay=5e-4, keep_prob=0.5,
               data_format=fmt, batch_size=Bt, coord_scale=w[0], noobj_scale=w[1], obj_scale=w[2], class_scale=w[3],
               nms_score_threshold=thr, nms_max_boxes=mb, nms_iou_threshold=iou,
               priors=y.synth.V3_PRIORS if version == 3 else y.synth.V2_PRIORS)
    if version == 3:
        cfg['num_priors'] = 3
    else:
        cfg.update(is_pretraining=False, rescore_confidence=False)
    cls = y.YOLOv3 if version == 3 else y.YOLOv2

    def given_as(hs):                                           # the layout the configuration announces
        hs = [h.permute(0, 3, 1, 2).contiguous() if fmt == 'channels_first' else h for h in hs]
        return hs if version == 3 else hs[0]

    heads = case.heads(seed % 9941, 2, obj_mean=0.0, cold_mean=-2.0)
    pred = cls(cfg, None).test_one_image(given_as(to_dev(heads, gpu)))
    # image 0 through a head object built from the case's own priors and multipliers (the class derives its from
    # the configuration): the same bits; that path is held to the oracle by the properties above
    want = case.head_obj().detect(to_dev([h[:1] for h in heads], gpu), thr, iou, mb).per_image()[0]
    event('detections of image 0: %s' % ('none' if len(want[0]) == 0 else 'some'))
    assert pred[0].dtype == np.float32 and pred[1].dtype == np.float32 and pred[2].dtype == np.int32
    assert np.array_equal(pred[2], want[2])
    assert np.array_equal(pred[0].view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(pred[1].view(np.uint32), want[1].view(np.uint32))
    G = 1 + seed % 7
    batches = [(given_as(to_dev(case.heads((seed + 7 * i) % 9931, Bt, obj_mean=-1.0), gpu)),
                torch.from_numpy(case.gt(seed + 13 * i, Bt, G, G + 3)).to(gpu)) for i in range(iters)]
    provider = dict(data_shape=[size, size, 3], num_train=iters * Bt, num_val=0,
                    train_generator=(None, iter(batches)), val_generator=None)
    net = cls(dict(cfg, mode='train', batch_size=Bt), provider)
    with contextlib.redirect_stdout(io.StringIO()):
        mean_loss = net.train_one_epoch(1e-3)
    wants = []
    for i in range(iters):
        hs = [h for h in case.heads((seed + 7 * i) % 9931, Bt, obj_mean=-1.0)]
        m = c_oracle.loss(case.oracle_cfg(c_oracle), hs, case.gt(seed + 13 * i, Bt, G, G + 3), w,
                          0.5 if version == 3 else 0.6, version == 3)[3]
        wants.append(np.float32(0.5) * m if version == 3 else m)
    assert_close(mean_loss, np.mean(wants), rtol=2 * RTOL, what='epoch mean loss')



@settings(max_examples=examples(8), **SETTINGS)
@given(geo=geometries(), B=st.integers(1, 3), batches=st.integers(2, 6), thr=st.sampled_from([1e-3, 0.05, 0.3, 0.5]),
       mb=st.sampled_from([3, 10, 100]), obj_mean=st.sampled_from([-3.0, 0.0, 3.0]),
       cap_kind=st.sampled_from(['default', 'one launch', 'every pair']), kernel=st.sampled_from([0, 1, 2]))
def test_candidates_and_select_on_two_streams(gpu, geo, B, batches, thr, mb, obj_mean, cap_kind, kernel):
    """yh_detect as its two calls on two streams, as a serving loop overlaps them: yh_detect_candidates of batch
    i on one stream, yh_detect_select of batch i on the other behind an event, two workspaces taking turns (a
    workspace's next streaming call waits for its select).  Every batch's outputs are yh_detect's."""
    import ctypes
    from yolo_tf_amd import _abi
    case, seed = geo
    cap = {'default': min(512, case.N * case.C), 'one launch': min(case.N * case.C, 2048), 'every pair': case.N * case.C}[cap_kind]
    lib = _abi.load()
    try:
        _abi.set_tuning(detect_kernel=kernel)
        sets = [to_dev(case.heads((seed + 37 * i) % 9929, B, obj_mean=obj_mean, cold_mean=-2.0), gpu) for i in range(batches)]
        hd = case.head_obj()
        want = [hd.detect(h, thr, 0.5, mb, max_candidates=cap, want_class_candidates=True) for h in sets]
        cfg = hd.config_for(sets[0])
        nbytes = ctypes.c_size_t(0)
        _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
        s_cand, s_sel = torch.cuda.Stream(gpu), torch.cuda.Stream(gpu)
        with torch.cuda.stream(s_cand):
            wss = [torch.zeros(max(256, nbytes.value), dtype=torch.uint8, device=gpu) for _ in range(2)]
        outs = [hd._alloc_outputs(B, mb, gpu, True) for _ in range(batches)]
        torch.cuda.synchronize(gpu)
        ev_c = [torch.cuda.Event() for _ in range(2)]
        ev_s = [torch.cuda.Event() for _ in range(2)]
        for i, h in enumerate(sets):
            k = i & 1
            s_cand.wait_event(ev_s[k])                         # the workspace's last select has read it (no-op at first)
            _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), _abi.pointer_array([t.data_ptr() for t in h]), B, thr, mb, cap,
                                                wss[k].data_ptr(), wss[k].numel(), ctypes.c_void_p(s_cand.cuda_stream)))
            ev_c[k].record(s_cand)
            s_sel.wait_event(ev_c[k])
            _abi.check(lib.yh_detect_select(ctypes.byref(cfg), B, 0.5, mb, cap, wss[k].data_ptr(), wss[k].numel(),
                                            *[t.data_ptr() for t in outs[i]], ctypes.c_void_p(s_sel.cuda_stream)))
            ev_s[k].record(s_sel)
        torch.cuda.synchronize(gpu)
        for i, w in enumerate(want):
            g = y.Detections(*outs[i], None)
            assert torch.equal(g.status, w.status), i
            if w.overflowed():
                continue
            assert torch.equal(g.class_candidates, w.class_candidates), i
            cases.same_detections(g.per_image(), w.per_image())
            for a, b in zip(g.per_image(), w.per_image()):
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), i
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), i
    finally:
        _abi.set_tuning()


@settings(max_examples=examples(10), **SETTINGS)
@given(geo=geometries(max_classes=90), B=st.integers(1, 2), thr=st.sampled_from([1e-3, 0.3, 0.5]), mb=st.sampled_from([3, 10]),
       share=st.sampled_from([0.001, 0.02, 0.3]), seed2=st.integers(0, 10 ** 6))
def test_detect_with_special_values_in_the_head(gpu, c_oracle, geo, B, thr, mb, share, seed2):
    """Head tensors with infinities, NaN, huge and denormal logits scattered over every channel: the decode has
    the oracle's NaN and infinity pattern and its finite values to 1e-5; score filter and NMS on the GPU's own
    decoded arrays are the oracle's bit for bit (NaN confidences are no candidates, NaN and infinite boxes
    suppress nothing); the fused call and every streaming kernel give the staged result."""
    from yolo_tf_amd import _abi
    case, seed = geo
    heads = case.heads(seed % 9913, B, obj_mean=0.0, cold_mean=-2.0)
    rng = np.random.Generator(np.random.PCG64(seed2))
    special = np.array([np.inf, -np.inf, np.nan, 1e30, -1e30, 1e-40, -1e-40, 88.0, -88.0, 104.0, -104.0, 0.0], np.float32)
    for h in heads:
        m = rng.uniform(size=h.shape) < share
        h[m] = rng.choice(special, size=int(m.sum()))
    dheads = to_dev(heads, gpu)
    hd = case.head_obj()
    boxes, conf = hd.decode(dheads)
    wb, wc = c_oracle.decode(case.oracle_cfg(c_oracle), heads)
    assert_close(boxes.cpu().numpy(), wb, rtol=RTOL, atol=RTOL * case.size * 4, what='boxes')
    assert_close(conf.cpu().numpy(), wc, rtol=RTOL, atol=1e-30, what='conf')
    hb, hc = boxes.cpu().numpy(), conf.cpu().numpy()
    staged = hd.nms(boxes, conf, thr, 0.5, mb, want_class_candidates=True).resolved()
    got = staged.per_image()
    cases.same_detections(got, c_oracle.nms(hb, hc, thr, 0.5, mb))
    assert np.array_equal(staged.class_candidates.cpu().numpy(), (hc >= np.float32(thr)).sum(axis=1).astype(np.int32))
    try:
        for t in (dict(), dict(row_steps=2), dict(detect_kernel=2, stream_tile_kib=1),
                  dict(detect_kernel=2, stream_tile_kib=4, stream_buffers=2, stream_units=300)):
            _abi.set_tuning(**t)
            for cap in (None, case.N * case.C):
                fused = case.head_obj().detect(dheads, thr, 0.5, mb, max_candidates=cap, want_class_candidates=True).resolved()
                cases.same_detections(fused.per_image(), got)
                for f, g in zip(fused.per_image(), got):
                    assert np.array_equal(f[0].view(np.uint32), g[0].view(np.uint32)), (t, cap)
                    assert np.array_equal(f[1].view(np.uint32), g[1].view(np.uint32)), (t, cap)
                assert torch.equal(fused.class_candidates, staged.class_candidates), (t, cap)
    finally:
        _abi.set_tuning()


@settings(max_examples=examples(10), **SETTINGS)
@given(data=st.data(), geo=geometries(max_classes=90), B=st.integers(1, 4), version=st.sampled_from([2, 3]),
       share=st.sampled_from([0.0005, 0.01, 0.2]), seed2=st.integers(0, 10 ** 6))
def test_loss_with_special_values_in_the_head(gpu, c_oracle, data, geo, B, version, share, seed2):
    """The loss of head tensors with infinities, NaN, huge and denormal logits: per-image losses and parts have
    the oracle's NaN and infinity pattern and its finite values to 1e-5 (an image without such a logit in a term
    keeps a finite term); the fused gradient pass, the two calls and the prepared call agree bit for bit."""
    case, seed = geo
    gt = data.draw(ground_truths(case, B))
    heads = case.heads(seed % 7877, B, obj_mean=-1.0)
    rng = np.random.Generator(np.random.PCG64(seed2))
    special = np.array([np.inf, -np.inf, np.nan, 1e30, -1e30, 1e-40, -1e-40, 88.0, -88.0, 104.0, -104.0, 0.0], np.float32)
    for h in heads:
        m = rng.uniform(size=h.shape) < share
        h[m] = rng.choice(special, size=int(m.sum()))
    norm, thr = version == 3, 0.5 if version == 3 else 0.6
    scales = (1.0, 0.5, 5.0, 2.0)
    dheads, dgt = to_dev(heads, gpu), torch.from_numpy(gt).to(gpu)
    hd = case.head_obj()
    out = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, want_mask=True)
    per, parts, _, _ = c_oracle.loss(case.oracle_cfg(c_oracle), heads, gt, scales, thr, norm)
    assert_close(out['parts'].cpu().numpy(), parts, rtol=RTOL, atol=1e-6, what='loss parts')
    assert_close(out['per_image'].cpu().numpy(), per, rtol=RTOL, atol=1e-6, what='per-image loss')
    gw = torch.linspace(0.25, 1.5, B, device=gpu)
    two = hd.loss_backward(dheads, dgt, *scales, norm, out['noobj_mask'], gw)
    fused = hd.loss(dheads, dgt, *scales, thr, norm, want_parts=True, grad_per_image=gw)
    other = case.head_obj()
    other.loss_prepare(dheads, dgt)
    prepared = other.loss(dheads, dgt, *scales, thr, norm, want_parts=True, grad_per_image=gw)

    def same(a, b):
        return a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=7e30, neginf=-7e30),
                                                  torch.nan_to_num(b, nan=-7.0, posinf=7e30, neginf=-7e30))
    for k in ('per_image', 'parts'):
        assert same(fused[k], out[k]) and same(prepared[k], out[k]), k
    for s in range(len(two)):
        assert same(fused['grads'][s], two[s]) and same(prepared['grads'][s], two[s]), s


@settings(max_examples=min(examples(5), 300), **SETTINGS)
@given(n=st.integers(2, 4), cin=st.integers(1, 9), cout=st.integers(1, 9), k=st.sampled_from([1, 3]), stride=st.sampled_from([1, 2]),
       h=st.integers(1, 13), w=st.integers(1, 13), act=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_conv_layer_on_random_shapes(gpu, n, cin, cout, k, stride, h, w, act, seed):
    """backbone.ConvBN on MIOpen (channels_last) against the independent NumPy float64 restatement of the
    reference's _conv_layer (TensorFlow 'same' padding for odd and even sizes at both strides, bias, batch-statistics
    normalisation with epsilon 1e-3, leaky ReLU 0.1) on random shapes: 1e-4."""
    import importlib.util, os
    from yolo_tf_amd import backbone as bb
    spec = importlib.util.spec_from_file_location(
        'make_conv_golden', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_conv_golden.py'))
    gold = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gold)
    rng = np.random.Generator(np.random.PCG64(seed))
    params = gold.Params(rng)
    params.conv_bn('l', cin, cout, k)
    x = rng.normal(0, 1, size=(n, cin, h, w))
    want = params.layer('l', x, stride=stride, activation=act)
    layer = bb.ConvBN(cin, cout, k, stride, activation=act)
    state = layer.state_dict()
    for name in ('conv.weight', 'conv.bias', 'bn.weight', 'bn.bias'):
        state[name] = torch.from_numpy(params.store['l.' + name]).to(state[name].dtype)
    layer.load_state_dict(state)
    layer = layer.train().to(gpu)
    got = layer(torch.from_numpy(x).float().to(gpu).contiguous(memory_format=torch.channels_last))
    assert tuple(got.shape) == want.shape
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), want, rtol=1e-4, atol=1e-4)
