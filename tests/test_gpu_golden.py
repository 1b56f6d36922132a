"""GPU: the HIP path against the committed golden fixtures (tests/golden/*.npz,
generated from the NumPy oracle; margin-safe, so kept rows must match exactly)."""
import os

import numpy as np
import pytest
import torch

import yolo_tf_amd as y

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return g, [g['head%d' % s] for s in range(g['grids'].shape[0])]


@pytest.mark.parametrize('name', ['detect_v2_416_voc', 'detect_v3_160'])
def test_detect_golden(gpu, name):
    g, heads = load(name)
    S = len(heads)
    hd = y.DetectionHead(int(g['num_classes']), list(g['priors']), g['box_mult'], [1.0] * S)
    dheads = [torch.from_numpy(h).to(gpu) for h in heads]
    boxes, conf = hd.decode(dheads)
    np.testing.assert_allclose(boxes.cpu().numpy(), g['bbox'], rtol=1e-5, atol=1e-5 * 416)
    np.testing.assert_allclose(conf.cpu().numpy(), g['conf'], rtol=1e-5, atol=1e-30)
    got = hd.detect(dheads, float(g['score_threshold']), float(g['iou_threshold']), int(g['max_boxes'])).per_image()
    off = 0
    for b, (s, bx, cl, ix) in enumerate(got):
        k = int(g['det_count'][b])
        assert np.array_equal(ix, g['det_index'][off:off + k]), 'image %d: kept rows' % b
        assert np.array_equal(cl, g['det_class'][off:off + k])
        np.testing.assert_allclose(s, g['det_scores'][off:off + k], rtol=1e-5)
        np.testing.assert_allclose(bx, g['det_bbox'][off:off + k], rtol=1e-5, atol=1e-5 * 416)
        off += k


@pytest.mark.parametrize('name', ['loss_v3_160', 'loss_v2_416_voc'])
def test_loss_golden(gpu, name):
    g, heads = load(name)
    S = len(heads)
    hd = y.DetectionHead(int(g['num_classes']), list(g['priors']), [1.0] * S, g['gt_rate'])
    sc = g['loss_scales']
    out = hd.loss([torch.from_numpy(h).to(gpu) for h in heads], torch.from_numpy(g['ground_truth']).to(gpu),
                  sc[0], sc[1], sc[2], sc[3], float(g['ignore_threshold']), int(g['version']) == 3,
                  want_parts=True, want_iou_max=True)
    assert np.array_equal(out['iou_max'].cpu().numpy().view(np.uint32), g['iou_max'].view(np.uint32))
    np.testing.assert_allclose(out['parts'].cpu().numpy(), g['loss_parts'], rtol=1e-5)
    np.testing.assert_allclose(out['per_image'].cpu().numpy(), g['loss_per_image'], rtol=1e-5)
    np.testing.assert_allclose(out['sum'].item() / len(g['loss_per_image']), g['loss_mean'], rtol=1e-5)
