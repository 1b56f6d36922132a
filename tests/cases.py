"""Shared test cases: head geometries of the benchmark configurations at sizes
the CPU oracle finishes in seconds."""
import numpy as np

import yolo_tf_amd as y
from oracle import yolo_head_oracle as O

syn = y.synth


class Case:
    def __init__(self, name, grids, priors, box_mult, gt_rate, C, size, version):
        self.name, self.grids, self.priors = name, grids, [np.asarray(p, np.float32) for p in priors]
        self.box_mult, self.gt_rate, self.C, self.size, self.version = box_mult, gt_rate, C, size, version
        self.P = self.priors[0].shape[0]
        self.N = sum(h * w * self.P for h, w in grids)

    def heads(self, seed, B, **dist):
        return syn.make_heads(seed, B, self.grids, self.P, self.C, **dist)

    def gt(self, seed, B, G, pad):
        return syn.make_ground_truth(seed, B, G, pad, self.size, self.C)

    def head_obj(self):
        return y.DetectionHead(self.C, self.priors, self.box_mult, self.gt_rate)

    def oracle_cfg(self, co):
        return co.make_config(self.grids, self.priors, self.box_mult, self.gt_rate, self.C)


def v3(size=416, C=80):
    return Case('v3-%d' % size, syn.v3_grids(size), O.v3_grid_priors(syn.V3_PRIORS),
                O.V3_BOX_MULT, O.V3_GT_RATE, C, size, 3)


def v2(size=416, C=20):
    return Case('v2-%d' % size, [(size // 32, size // 32)], [np.asarray(syn.V2_PRIORS, np.float32)],
                O.V2_BOX_MULT, O.V2_GT_RATE, C, size, 2)


def stress(grid=20, P=10, C=80):
    """Single scale, P priors per cell (num_priors is config in the reference)."""
    rng = np.random.Generator(np.random.PCG64(99))
    priors = rng.uniform(0.5, 4.0, size=(P, 2)).astype(np.float32)
    return Case('stress-%d' % grid, [(grid, grid)], [priors], (32.0,), (32.0,), C, grid * 32, 2)


def same_detections(got, want):
    """Bit-exact: box rows, classes and scores' bit patterns of two per-image lists."""
    assert len(got) == len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g[3], w[3]), 'image %d: kept box rows differ' % b
        assert np.array_equal(g[2], w[2]), 'image %d: class ids differ' % b
