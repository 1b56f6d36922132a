"""yh_detect_static: yh_detect with the library's static device scratch instead of a caller
workspace (for callers that cannot allocate).  Called through the C ABI; every output must be
yh_detect's, bit for bit, on calls in a row whose shapes differ (the array keeps the last call's
contents, so the second, smaller call finds the first one's data where its counters lie)."""
import ctypes

import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from yolo_tf_amd import _abi
from tests import cases

pytestmark = pytest.mark.gpu


def _outputs(B, C, mb, dev):
    cap = C * mb
    return [torch.full((B,), -7, dtype=torch.int32, device=dev),
            torch.full((B, cap), -7.0, dtype=torch.float32, device=dev),
            torch.full((B, cap, 4), -7.0, dtype=torch.float32, device=dev),
            torch.full((B, cap), -7, dtype=torch.int32, device=dev),
            torch.full((B, cap), -7, dtype=torch.int32, device=dev),
            torch.full((B, C), -7, dtype=torch.int32, device=dev),
            torch.full((B,), -7, dtype=torch.int32, device=dev)]


def _both(case, heads, thr, iou, mb, cap, dev):
    """(yh_detect's outputs with a fresh zero-filled workspace, yh_detect_static's), raw C ABI."""
    lib = _abi.load()
    cfg = y.head.make_config(case.grids, case.priors, case.box_mult, case.gt_rate, case.C)
    B = int(heads[0].shape[0])
    hp = _abi.pointer_array([h.data_ptr() for h in heads])
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nbytes = ctypes.c_size_t(0)
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
    ws = torch.zeros(max(256, nbytes.value), dtype=torch.uint8, device=dev)
    want = _outputs(B, case.C, mb, dev)
    _abi.check(lib.yh_detect(ctypes.byref(cfg), hp, B, thr, iou, mb, cap, ws.data_ptr(), ws.numel(),
                             *[t.data_ptr() for t in want], st))
    got = _outputs(B, case.C, mb, dev)
    _abi.check(lib.yh_detect_static(ctypes.byref(cfg), hp, B, thr, iou, mb, cap, *[t.data_ptr() for t in got], st))
    torch.cuda.synchronize(dev)
    return want, got


def _assert_same(want, got):
    count = want[0].cpu().numpy()
    assert np.array_equal(got[0].cpu().numpy(), count), 'out_count differs'
    assert torch.equal(got[5], want[5]), 'out_class_candidates differs'
    assert torch.equal(got[6], want[6]), 'status differs'
    for b, k in enumerate(count):
        for i in (1, 2, 3, 4):
            a, w = got[i][b, :k].cpu().numpy(), want[i][b, :k].cpu().numpy()
            assert np.array_equal(a.view(np.uint32), w.view(np.uint32)), 'output %d of image %d differs' % (i, b)
    return int(count.sum())


def test_static_scratch_matches_workspace_calls_in_a_row(gpu):
    # The shapes are chosen so that the counters of each smaller call lie inside the region the
    # big call fills completely (its work units' key slots, 320..545 KiB into the array; an
    # unused slot holds an all-ones key): about 351 KiB for the second call, 446 KiB for the third.
    big = cases.v3()                                           # 10647 boxes per image, 80 classes
    heads_big = y.synth.torch_heads(gpu, 901, 20, big.grids, big.P, big.C)
    assert _assert_same(*_both(big, heads_big, 0.5, 0.5, 10, 512, gpu)) > 0
    # a smaller input right after
    small = cases.v3(size=160, C=20)
    heads_small5 = y.synth.torch_heads(gpu, 902, 5, small.grids, small.P, small.C, obj_mean=-1.0)
    assert _assert_same(*_both(small, heads_small5, 0.4, 0.5, 10, 2048, gpu)) > 0
    # the multi-launch select stage (capacity beyond 2048: class histogram and cursors in the scratch),
    # after the big call again, then once more on its own leftovers
    assert _assert_same(*_both(big, heads_big, 0.5, 0.5, 10, 512, gpu)) > 0
    heads_small = [h[:2].contiguous() for h in heads_small5]
    for _ in range(2):
        assert _assert_same(*_both(small, heads_small, 0.05, 0.4, 20, 6000, gpu)) > 0
    # and the first shape again
    assert _assert_same(*_both(big, heads_big, 0.5, 0.5, 10, 512, gpu)) > 0
    # an overflowing capacity flags the same images
    want, got = _both(small, heads_small, 0.05, 0.4, 20, 8, gpu)
    assert int((want[6] & _abi.YH_STATUS_CANDIDATE_OVERFLOW).max()) == 1 and torch.equal(want[6], got[6])


def test_static_scratch_rejects_what_does_not_fit(gpu):
    lib = _abi.load()
    case = cases.v3(size=608)
    cfg = y.head.make_config(case.grids, case.priors, case.box_mult, case.gt_rate, case.C)
    B, mb, cap = 256, 10, 512
    nbytes = ctypes.c_size_t(0)
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, mb, ctypes.byref(nbytes)))
    assert nbytes.value > _abi.YH_STATIC_SCRATCH_BYTES
    fake = _abi.pointer_array([256, 256, 256])                 # never dereferenced: the size check comes first
    outs = _outputs(1, 1, 1, gpu)
    rc = lib.yh_detect_static(ctypes.byref(cfg), fake, B, 0.5, 0.5, mb, cap, *[t.data_ptr() for t in outs],
                              ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream))
    assert rc == _abi.YH_ERR_LIMIT and b'static scratch' in lib.yh_last_error()
    # an empty batch is fine and touches nothing
    assert lib.yh_detect_static(ctypes.byref(cfg), None, 0, 0.5, 0.5, mb, cap, *[None] * 7, None) == _abi.YH_OK


def test_python_wrapper_of_the_static_entry_point(gpu):
    """DetectionHead.detect_static against DetectionHead.detect on the same batch."""
    case = cases.v3(416)
    heads = [torch.from_numpy(h).to(gpu) for h in y.synth.make_heads(31, 3, case.grids, case.P, case.C)]
    hd = y.DetectionHead(case.C, case.priors, case.box_mult, case.gt_rate)
    want = hd.detect(heads, 0.5, 0.5, 10, max_candidates=512, want_class_candidates=True)
    outs = hd._alloc_outputs(3, 10, gpu, True)
    got = hd.detect_static(heads, 0.5, 0.5, 10, 512, outs)
    assert not got.overflowed()
    for (gs, gb, gc, gi), (ws, wb, wc, wi) in zip(got.per_image(), want.per_image()):
        assert np.array_equal(gi, wi) and np.array_equal(gc, wc)
        assert np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
        assert np.array_equal(gb.view(np.uint32), wb.view(np.uint32))
