"""Seeded random sweeps over geometry, value distributions and thresholds, larger than the hypothesis cases:
the objectness-first kernel against the staged path (full decode, then yh_nms), and the workgroup-tile kernel
(one workgroup per run of tiles, or a fixed number walking the runs) and yh_nms against the C oracle's NMS on
the GPU's decoded arrays.  Bit-exact: kept rows, classes, scores, boxes, candidate counts."""
import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from tests import cases

pytestmark = pytest.mark.gpu


def _geometry(rng, classes, top):
    S, P, C = int(rng.integers(1, 4)), int(rng.choice([1, 3, 5, 9])), int(rng.choice(classes))
    grids = [(int(rng.integers(1, top)), int(rng.integers(1, top))) for _ in range(S)]
    priors = [rng.uniform(0.3, 4.0, size=(P, 2)).astype(np.float32) for _ in range(S)]
    mult = tuple(float(rng.choice([8., 16., 32.])) for _ in range(S))
    return S, P, C, grids, priors, mult


def test_rows_kernel_random_sweep(gpu, tuning):
    rng = np.random.Generator(np.random.PCG64(77))
    compared = 0
    for it in range(60):
        S, P, C, grids, priors, mult = _geometry(rng, [1, 7, 20, 80, 96, 97, 113, 200], 60)
        B = int(rng.integers(1, 5))
        dist = dict(obj_mean=float(rng.choice([-6., -4., -3., -1., 0., 2., 6.])), obj_std=float(rng.choice([1., 2., 3.])),
                    hot_mean=float(rng.choice([-1., 1., 3.])), cold_mean=float(rng.choice([-8., -5., -3.])),
                    cold_std=float(rng.choice([0.5, 1.5])))
        thr, mb = float(rng.choice([0.1, 0.3, 0.5, 0.8])), int(rng.choice([1, 5, 20, 100]))
        heads = y.synth.make_heads(int(rng.integers(1 << 30)), B, grids, P, C, **dist)
        dheads = [torch.from_numpy(h).to(gpu) for h in heads]
        hd = y.DetectionHead(C, priors, mult, mult)
        boxes, conf = hd.decode(dheads)
        cc = (conf.cpu().numpy() >= np.float32(thr)).sum(axis=1).astype(np.int32)
        most = int(cc.sum(axis=1).max())
        staged = hd.nms(boxes, conf, thr, 0.5, mb, max_candidates=max(2048, most + 1), want_class_candidates=True)
        assert not staged.overflowed()
        # the plan's own steps per unit and a random number of them; steps read whole from the default number of
        # surviving rows, from a single one (nearly every step then, entered through the objectness column, through
        # the probe of the unit's first step, or blind behind another whole step) and never
        for steps, dense_min, probe in ((0, 0, 0), (int(rng.integers(1, 17)), int(rng.choice([1, 8, 30])), int(rng.choice([1, 2]))),
                                        (int(rng.integers(1, 17)), 65, 0), (0, 1, 1)):
            tuning(row_steps=steps, rows_dense_min=dense_min, rows_probe=probe)
            got = y.DetectionHead(C, priors, mult, mult).detect(dheads, thr, 0.5, mb, max_candidates=2048,
                                                                want_class_candidates=True)
            if most > 2048:
                assert got.overflowed()
                continue
            assert not got.overflowed(), (it, most)
            cases.same_detections(got.per_image(), staged.per_image())
            for a, b in zip(got.per_image(), staged.per_image()):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert np.array_equal(got.class_candidates.cpu().numpy(), cc), it
            compared += 1
    assert compared >= 120


def test_tile_kernel_random_sweep(gpu, tuning, c_oracle):
    rng = np.random.Generator(np.random.PCG64(99))
    for it in range(40):
        S, P, C, grids, priors, mult = _geometry(rng, [1, 2, 7, 20, 80, 97, 200], 30)
        B = int(rng.integers(1, 4))
        dist = dict(obj_mean=float(rng.choice([-3., 0., 2.])), hot_mean=float(rng.choice([-1., 1., 3.])),
                    cold_mean=float(rng.choice([-12., -6., -3.])), cold_std=float(rng.choice([0.5, 1.5])))
        thr, mb = float(rng.choice([0.0, 0.001, 0.05, 0.3, 0.7])), int(rng.choice([1, 10, 100, 300]))
        heads = y.synth.make_heads(int(rng.integers(1 << 30)), B, grids, P, C, **dist)
        dheads = [torch.from_numpy(h).to(gpu) for h in heads]
        boxes, conf = y.DetectionHead(C, priors, mult, mult).decode(dheads)
        hb, hc = boxes.cpu().numpy(), conf.cpu().numpy()
        cc = (hc >= np.float32(thr)).sum(axis=1).astype(np.int32)
        cap = max(4096, int(cc.sum(axis=1).max()) + 1)       # beyond the single-launch select stage: the tile kernel
        want = c_oracle.nms(hb, hc, thr, 0.5, mb)
        for resident in (0, 1):
            tuning(stream_resident=resident, stream_blocks=int(rng.choice([0, 7, 100000])))
            got = y.DetectionHead(C, priors, mult, mult).detect(dheads, thr, 0.5, mb, max_candidates=cap,
                                                                want_class_candidates=True)
            assert not got.overflowed(), (it, cap)
            cases.same_detections(got.per_image(), want)
            assert np.array_equal(got.class_candidates.cpu().numpy(), cc), it
            staged = y.DetectionHead(C, priors, mult, mult).nms(boxes, conf, thr, 0.5, mb, max_candidates=cap)
            cases.same_detections(staged.per_image(), want)
