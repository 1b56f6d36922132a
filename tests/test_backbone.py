"""The convolution stacks that feed the path (backbone.py): structure, TensorFlow 'same'
padding, the NHWC hand-over, optimiser and checkpoint stand-ins.  Parity of the stacks with
the reference is unpinned (TensorFlow is absent and the reference holds no activations);
these tests pin the structure the reference's source states."""
import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from yolo_tf_amd import backbone as bb


def test_same_padding_matches_tensorflow_rule():
    # out = ceil(in / stride); the odd pixel goes after
    assert bb.same_padding(416, 3, 1) == (1, 1)
    assert bb.same_padding(416, 3, 2) == (0, 1)
    assert bb.same_padding(415, 3, 2) == (1, 1)
    assert bb.same_padding(13, 1, 1) == (0, 0)
    assert bb.same_padding(7, 2, 2) == (0, 1)
    assert bb.same_padding(8, 2, 2) == (0, 0)


def test_stride2_conv_pads_bottom_right_only():
    torch.manual_seed(0)
    layer = bb.ConvBN(1, 1, 3, 2, activation=False)
    x = torch.arange(16, dtype=torch.float32).reshape(1, 1, 4, 4)
    w = layer.conv.weight.detach()[0, 0]
    want = torch.zeros(2, 2)
    xp = torch.zeros(5, 5)
    xp[:4, :4] = x[0, 0]
    for i in range(2):
        for j in range(2):
            want[i, j] = (xp[2 * i:2 * i + 3, 2 * j:2 * j + 3] * w).sum()
    got = layer.conv(bb._pad_same(x, 3, 2))[0, 0] - layer.conv.bias.detach()
    assert torch.allclose(got, want, atol=1e-5)


def _conv_params(cin, cout, k):
    return cin * cout * k * k + cout + 2 * cout               # kernel + bias + BN gamma, beta


def _darknet53_param_count(final_units):
    n = _conv_params(3, 32, 3)
    cin = 32
    for filters, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        n += _conv_params(cin, filters, 3)
        n += blocks * (_conv_params(filters, filters // 2, 1) + _conv_params(filters // 2, filters, 3))
        cin = filters
    for bottom, filters, pyramid in ((1024, 1024, None), (512, 256, 512), (256, 128, 128)):
        c = bottom
        if pyramid is not None:
            n += _conv_params(pyramid, filters, 1)
            c += filters
        half = filters // 2
        n += _conv_params(c, half, 1) + _conv_params(half, filters, 3)
        n += 2 * (_conv_params(filters, half, 1) + _conv_params(half, filters, 3))
        n += _conv_params(filters, final_units, 1)
    return n


def test_darknet53_structure_and_nhwc_heads():
    torch.manual_seed(1)
    net = bb.Darknet53Heads(num_classes=4, num_priors=3)
    assert sum(p.numel() for p in net.parameters()) == _darknet53_param_count(27)
    images = np.random.default_rng(0).uniform(0, 255, size=(2, 64, 64, 3)).astype(np.float32)
    with torch.no_grad():
        heads = net(images)
    assert [tuple(h.shape) for h in heads] == [(2, 2, 2, 27), (2, 4, 4, 27), (2, 8, 8, 27)]
    assert all(h.is_contiguous() and h.dtype == torch.float32 for h in heads)
    assert all(torch.isfinite(h).all() for h in heads)
    # channels_first hands over NCHW of the same values (the class transposes, YOLOv3.py:88-91)
    net_cf = bb.Darknet53Heads(num_classes=4, num_priors=3, data_format='channels_first')
    net_cf.load_state_dict(net.state_dict())
    with torch.no_grad():
        heads_cf = net_cf(np.ascontiguousarray(images.transpose(0, 3, 1, 2)))
    for a, b in zip(heads, heads_cf):
        assert tuple(b.shape) == (a.shape[0], a.shape[3], a.shape[1], a.shape[2])
        assert torch.allclose(a, b.permute(0, 2, 3, 1), atol=1e-4, rtol=1e-4)


def test_darknet19_structure():
    torch.manual_seed(2)
    net = bb.Darknet19Head(num_classes=20, num_priors=5)
    convs = [spec for spec in net.FEATURES if spec != 'M']
    assert len(convs) == 18 and len(net.head) == 5
    assert net.predictions.conv.in_channels == 1024 + 512 and net.predictions.conv.out_channels == 125
    assert not net.predictions.activation
    images = torch.rand(1, 96, 64, 3) * 255
    with torch.no_grad():
        (pred,) = net(images)
    assert tuple(pred.shape) == (1, 3, 2, 125) and pred.is_contiguous()


def test_batch_norm_uses_batch_statistics_with_tf_defaults():
    layer = bb.ConvBN(3, 8, 1)
    assert layer.bn.eps == 1e-3 and abs(layer.bn.momentum - 0.01) < 1e-12
    net = bb.Darknet19Head(2, 1)
    assert net.training                                       # YOLOv2.py:49, 351: training=True is baked in


def test_momentum_step_and_weight_decay():
    torch.manual_seed(3)
    net = bb.Darknet19Head(1, 1, weight_decay=0.5)
    p = net.predictions.conv.bias
    p.data.fill_(2.0)
    for q in net.parameters():
        q.grad = torch.zeros_like(q)
    p.grad = torch.ones_like(p)
    net.step(0.1)                                             # accum = 1 + 0.5 * 2 = 2; p = 2 - 0.2
    assert torch.allclose(p, torch.full_like(p, 1.8))
    p.grad = torch.zeros_like(p)
    net.step(0.1)                                             # accum = 0.9 * 2 + 0.5 * 1.8 = 2.7; p = 1.8 - 0.27
    assert torch.allclose(p, torch.full_like(p, 1.53), atol=1e-6)
    want = sum(float((q.detach().double() ** 2).sum()) for q in net.parameters()) / 2
    assert abs(net.l2_loss_sum() - want) <= 1e-9 * want


def test_checkpoint_round_trip_and_pretraining_subset(tmp_path):
    torch.manual_seed(4)
    a = bb.Darknet53Heads(2, 3)
    b = bb.Darknet53Heads(2, 3)
    path = str(tmp_path / 'ckpt' / 'model-7')
    a.save(path)
    b.load_pretraining(path)
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa if k.startswith('backone.'))
    assert any(not torch.equal(sa[k], sb[k]) for k in sa if k.startswith('head.') and k.endswith('conv.weight'))
    b.load(path)
    assert all(torch.equal(sa[k], b.state_dict()[k]) for k in sa)


@pytest.mark.gpu
def test_yolov3_end_to_end_from_images():
    torch.manual_seed(5)
    priors = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]
    cfg = dict(mode='test', data_shape=[128, 128, 3], num_classes=6, weight_decay=1e-4, keep_prob=0.5,
               data_format='channels_last', batch_size=1, coord_scale=1, noobj_scale=1, obj_scale=5., class_scale=1.,
               num_priors=3, nms_score_threshold=0.3, nms_max_boxes=10, nms_iou_threshold=0.5, priors=priors)
    net = y.YOLOv3(cfg, None)
    image = np.random.default_rng(1).uniform(0, 255, size=(1, 128, 128, 3)).astype(np.float32)
    scores, bbox, class_id = net.test_one_image(image)
    assert isinstance(net.head_provider, bb.Darknet53Heads)
    assert scores.dtype == np.float32 and bbox.dtype == np.float32 and class_id.dtype == np.int32
    assert bbox.shape == (scores.shape[0], 4) and class_id.shape == scores.shape
    # the heads of a second pass, recorded, through the oracle (two passes of the convolution
    # stack need not agree to the last bit, so the comparison is on the very tensors the path read)
    from oracle import yolo_head_oracle as orc
    stack, seen = net.head_provider, {}

    def recording(images):
        with torch.no_grad():
            seen['heads'] = stack(images)
        return seen['heads']
    net.head_provider = recording
    scores, bbox, class_id = net.test_one_image(image)
    heads = [h[0].cpu().numpy() for h in seen['heads']]
    assert [h.shape for h in heads] == [(4, 4, 33), (8, 8, 33), (16, 16, 33)]
    want = orc.detect(heads, orc.v3_grid_priors(priors), orc.V3_BOX_MULT, 6, 0.3, 10, 0.5)
    assert len(scores) > 0 and class_id.tolist() == want[2].tolist()
    np.testing.assert_allclose(scores, want[0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bbox, want[1], rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_yolov2_trains_end_to_end_from_images():
    torch.manual_seed(6)
    rng = np.random.default_rng(2)
    B, size, C = 2, 96, 3
    images = torch.from_numpy(rng.uniform(0, 255, size=(B, size, size, 3)).astype(np.float32)).cuda()
    gt = torch.from_numpy(y.synth.make_ground_truth(3, B, 2, 4, size, C)).cuda()

    def batches():
        while True:
            yield images, gt
    cfg = dict(mode='train', data_shape=[size, size, 3], num_classes=C, weight_decay=1e-4, keep_prob=0.5,
               data_format='channels_last', batch_size=B, coord_scale=1, noobj_scale=1, obj_scale=5., class_scale=1.,
               nms_score_threshold=0.5, nms_max_boxes=10, nms_iou_threshold=0.5, rescore_confidence=False,
               priors=[[1., 1.], [2., 1.5]])
    prov = dict(num_train=3 * B, num_val=0, train_generator=(None, batches()), val_generator=None)
    net = y.YOLOv2(cfg, prov)
    first = float(net.train_one_epoch(1e-3))
    for _ in range(3):
        last = float(net.train_one_epoch(1e-3))
    assert isinstance(net.head_provider, bb.Darknet19Head)
    assert np.isfinite(first) and np.isfinite(last) and last < first


# ---- the convolution building blocks against an independent NumPy float64 restatement -----------------
# tests/golden/conv_chain.npz comes from tests/golden/make_conv_golden.py, which restates _conv_layer, _bn,
# _darknet_block and _yolo3_header (YOLOv3.py:396-417, 487-517) from the source text with NumPy only.  On
# the GPU the modules run on MIOpen in channels_last memory, which is what feeds the HIP kernels.
def _load_conv_golden():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'conv_chain.npz')
    return np.load(path)


def _fill(module, gold, prefix):
    state = module.state_dict()
    for name in state:
        if name.endswith(('conv.weight', 'conv.bias', 'bn.weight', 'bn.bias')):
            state[name] = torch.from_numpy(gold[prefix + name]).to(state[name].dtype)
    module.load_state_dict(state)
    return module.train()                                     # batch statistics, as the reference's graph has them


def _conv_blocks_match(dev, tol):
    gold = _load_conv_golden()
    t = lambda a: torch.from_numpy(gold[a]).float().to(dev).contiguous(memory_format=torch.channels_last)
    close = lambda got, name: np.testing.assert_allclose(got.detach().float().cpu().numpy(), gold[name], rtol=tol, atol=tol,
                                                         err_msg=name)
    x = t('layer.x')
    for name, cout, k, stride, act in (('c3', 7, 3, 1, True), ('c3s2', 6, 3, 2, True), ('c1', 4, 1, 1, False)):
        layer = _fill(bb.ConvBN(5, cout, k, stride, activation=act), gold, 'layer.%s.' % name).to(dev)
        close(layer(x), 'layer.%s.y' % name)
    block = _fill(bb.DarknetBlock(4, 8, 2), gold, 'block.').to(dev)
    close(block(t('block.x')), 'block.y')
    header = _fill(bb.Yolo3Header(6, 8, 11, pyramid_channels=5), gold, 'header.').to(dev)
    pred, conv5 = header(t('header.bottom'), t('header.pyramid'))
    close(conv5, 'header.conv5')
    close(pred, 'header.pred')
    assert pred.shape == (2, 11, 7, 6)


def test_conv_blocks_match_the_numpy_restatement_cpu():
    _conv_blocks_match(torch.device('cpu'), 1e-4)


@pytest.mark.gpu
def test_conv_blocks_match_the_numpy_restatement_on_miopen(gpu):
    """The same on the MI355X (MIOpen convolutions, channels_last): 1e-4, fp32 against float64."""
    _conv_blocks_match(gpu, 1e-4)
