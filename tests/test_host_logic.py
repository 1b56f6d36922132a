"""CPU: host-side logic of the class API and the data-parallel helpers."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from oracle import yolo_head_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

V3_CFG = dict(mode='test', data_shape=[416, 416, 3], num_classes=80, weight_decay=5e-4, keep_prob=0.5,
              data_format='channels_last', batch_size=12, coord_scale=1, noobj_scale=1, obj_scale=5.,
              class_scale=1., num_priors=3, nms_score_threshold=0.5, nms_max_boxes=10,
              nms_iou_threshold=0.5, priors=y.synth.V3_PRIORS)
V2_CFG = dict(mode='test', is_pretraining=False, data_shape=[416, 416, 3], num_classes=20, weight_decay=1e-4,
              keep_prob=0.5, data_format='channels_last', batch_size=32, coord_scale=1, noobj_scale=1,
              obj_scale=5., class_scale=1., nms_score_threshold=0.5, nms_max_boxes=10, nms_iou_threshold=0.5,
              rescore_confidence=False, priors=y.synth.V2_PRIORS)


def test_v3_constructor_mirrors_reference():
    net = y.YOLOv3(V3_CFG, None)
    assert net.batch_size == 1                      # forced in test mode
    assert net.final_units == 255 and net.stride == [8., 16., 32.]
    want = O.v3_grid_priors(y.synth.V3_PRIORS)
    for a, b in zip(net.priors, want):
        assert a.shape == (1, 1, 3, 2) and np.array_equal(a.reshape(-1, 2), b)
    assert net.head.box_mult == [32., 32., 16.] and net.head.gt_rate == [32., 16., 8.]
    for bad in (dict(V3_CFG, mode='eval'), dict(V3_CFG, data_format='nchw'), dict(V3_CFG, data_shape=[416, 416])):
        with pytest.raises(AssertionError):
            y.YOLOv3(bad, None)
    with pytest.raises(AssertionError):
        net.save_weight('newest', '/tmp/x')


def test_v2_constructor_mirrors_reference():
    net = y.YOLOv2(V2_CFG, None)
    assert net.batch_size == 1 and net.num_priors == 5 and net.final_units == 125
    assert net.priors.shape == (1, 1, 5, 2) and net.head.box_mult == [32.0]
    provider = dict(data_shape=[416, 416, 3], num_train=10, num_val=0, train_generator=(None, iter([])), val_generator=None)
    assert y.YOLOv2(dict(V2_CFG, mode='train'), provider).batch_size == 32


def test_get_priors_and_normalized_gn_match_the_restatement():
    net3, net2 = y.YOLOv3(V3_CFG, None), y.YOLOv2(V2_CFG, None)
    for net, pr in ((net3, net3.priors[1]), (net2, net2.priors)):
        got = net._get_priors((1, 5, 7, 0), pr)
        want = O.get_priors(5, 7, pr.reshape(-1, 2))
        for a, b in zip(got, want):
            assert a.dtype == np.float32 and np.array_equal(a, b)
    gt = y.synth.make_ground_truth(5, 2, 4, 9, 416, 20)
    for net in (net3, net2):
        net.ground_truth = torch.from_numpy(gt)
        got = net._get_normlized_gn(16.0, 1)
        want = O.normalized_gn(gt[1], 16.0)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)


def test_shard_range_partitions_the_batch():
    for total in (0, 1, 7, 64, 256, 513):
        for world in (1, 2, 3, 4, 8):
            spans = [y.parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_global_mean_loss_single_process():
    v = y.parallel.global_mean_loss(torch.tensor([12.0]), 4)
    assert v.dtype == torch.float32 and v.dim() == 0 and float(v) == 3.0


def test_synth_shapes_and_padding():
    heads = y.synth.make_heads(1, 2, y.synth.v3_grids(96), 3, 4)
    assert [h.shape for h in heads] == [(2, 3, 3, 27), (2, 6, 6, 27), (2, 12, 12, 27)]
    assert all(h.dtype == np.float32 for h in heads)
    assert np.array_equal(heads[0], y.synth.make_heads(1, 2, y.synth.v3_grids(96), 3, 4)[0])
    gt = y.synth.make_ground_truth(2, 3, 5, 8, 96, 4)
    assert gt.shape == (3, 8, 5) and np.all(gt[:, 5:] == -1) and np.all(gt[:, :5, 0] >= 1)
    with pytest.raises(ValueError):
        y.synth.make_ground_truth(2, 3, 8, 8, 96, 4)


def test_two_rank_gloo_loss_mean_matches_full_batch():
    """world_size 2 on CPU (gloo): image sharding + the scalar all-reduce give
    the full-batch reduce_mean.  Per-rank per-image losses come from the CPU
    oracle here (the HIP kernels need a GPU); the code under test is the
    sharding and the all-reduce."""
    script = os.path.join(ROOT, 'tests', 'gloo_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29731', script],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('GLOO_OK') == 2, r.stdout + r.stderr


def test_two_rank_gloo_optimizer_step():
    """world_size 2 on CPU (gloo): the convolution stacks' step(lr) all-reduces the
    gradients in one flat bucket, so both ranks take the step of the summed gradient."""
    script = os.path.join(ROOT, 'tests', 'gloo_step_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29733', script],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('GLOO_STEP_OK') == 2, r.stdout + r.stderr


def test_ground_truth_wire_format_round_trip():
    """utils/image_augmentor.py:205-244: centre filter, clamp, (yc, xc, h, w, class), -1 padding."""
    boxes = [[10, 20, 50, 80], [-30, -30, 10, 10], [400, 400, 430, 440], [390, 10, 430, 60]]
    cls = [3, 5, 7, 9]
    blk = y.wire.encode_ground_truth(boxes, cls, (416, 416), 6)
    assert blk.shape == (6, 5) and blk.dtype == np.float32
    # box 1: centre (-10, -10) outside; box 2: centre (415, 420): x centre beyond 415 -> dropped
    assert np.array_equal(blk[0], np.float32([30, 50, 40, 60, 3]))
    # box 3 is clamped to 415 before centre and size are taken
    assert np.array_equal(blk[1], np.float32([(390 + 415) / 2., 35, 25, 50, 9]))
    assert np.all(blk[2:] == -1)
    b2, c2 = y.wire.decode_ground_truth(blk)
    assert np.array_equal(c2, [3, 9]) and np.allclose(b2[0], [10, 20, 50, 80])
    from oracle import yolo_head_oracle as O
    yx, hw, lab = O.normalized_gn(blk, 32.0)
    assert yx.shape == (2, 2) and np.array_equal(lab, [3, 9])
    with pytest.raises(ValueError):
        y.wire.encode_ground_truth(boxes[:1] * 6, [1] * 6, (416, 416), 6)      # no pad row left
    batch = y.wire.encode_ground_truth_batch([boxes, boxes[:1]], [cls, cls[:1]], (416, 416), 6)
    assert batch.shape == (2, 6, 5) and np.all(batch[1, 1:] == -1)
    assert y.wire.encode_ground_truth([], [], (416, 416), 4).tolist() == [[-1.0] * 5] * 4


def test_detection_records_and_voc_names():
    assert len(y.wire.VOC_CLASSNAMES) == 20 and y.wire.classname_to_ids['tvmonitor'] == 19
    rec = y.wire.detections_to_records([0.9], [[1, 2, 3, 4]], [14])
    assert rec == [dict(name='person', class_id=14, score=0.9 if False else rec[0]['score'], y1=1.0, x1=2.0, y2=3.0, x2=4.0)]
    assert abs(rec[0]['score'] - 0.9) < 1e-7
