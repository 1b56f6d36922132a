"""Generates tests/golden/conv_chain.npz: an independent NumPy float64 restatement of the reference's
convolution building blocks, for the tests that hold backbone.ConvBN / DarknetBlock / Yolo3Header to it.

Restated from the reference's source text (TensorFlow is absent, so these vectors pin the torch modules to
this reading of the source, not to TensorFlow's arithmetic):
  _conv_layer   YOLOv3.py:496-509: tf.layers.conv2d(padding='same', use_bias) -> _bn -> leaky_relu(0.1);
  _bn           YOLOv3.py:511-517: batch_normalization(training=True, epsilon 1e-3): batch mean and biased
                variance over (N, H, W), gamma and beta per channel;
  _darknet_block YOLOv3.py:487-494: 3x3 stride-2 convolution, then residual units of 1x1 (filters / 2) and
                3x3 (filters), added to their input;
  _yolo3_header YOLOv3.py:396-417: with a pyramid input, its 1x1 convolution (batch-normalised, no
                activation), nearest-neighbour resize to the bottom's grid, concatenation [bottom, resized];
                then 1x1 / 3x3 x 3, and the 1x1 prediction convolution on top of the sixth.
TensorFlow's 'same' padding: out = ceil(in / stride), total = max((out - 1) * stride + kernel - in, 0),
total // 2 rows before, the rest after.  Nearest resize (align_corners False): source = floor(i * in / out).

Run from the repo root:  python tests/golden/make_conv_golden.py      (NumPy only; imports nothing of the package)"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EPS, SLOPE = 1e-3, 0.1


def same_pad(size, kernel, stride):
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    return total // 2, total - total // 2, out


def conv2d_same(x, w, b, stride):
    """x [N, Cin, H, W], w [Cout, Cin, kh, kw] (cross-correlation, as tf.nn.conv2d), b [Cout]."""
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    top, bottom, oh = same_pad(h, kh, stride)
    left, right, ow = same_pad(wd, kw, stride)
    xp = np.zeros((n, cin, h + top + bottom, wd + left + right), np.float64)
    xp[:, :, top:top + h, left:left + wd] = x
    out = np.zeros((n, cout, oh, ow), np.float64)
    for i in range(oh):
        for j in range(ow):
            patch = xp[:, :, i * stride:i * stride + kh, j * stride:j * stride + kw]      # [N, Cin, kh, kw]
            out[:, :, i, j] = np.tensordot(patch, w, axes=([1, 2, 3], [1, 2, 3]))
    return out + b.reshape(1, -1, 1, 1)


def batch_norm(x, gamma, beta):
    mean = x.mean(axis=(0, 2, 3), keepdims=True)
    var = ((x - mean) ** 2).mean(axis=(0, 2, 3), keepdims=True)
    return gamma.reshape(1, -1, 1, 1) * (x - mean) / np.sqrt(var + EPS) + beta.reshape(1, -1, 1, 1)


def leaky(x):
    return np.where(x > 0, x, SLOPE * x)


def resize_nearest(x, oh, ow):
    ih, iw = x.shape[2], x.shape[3]
    rows = np.floor(np.arange(oh) * (ih / oh)).astype(int)
    cols = np.floor(np.arange(ow) * (iw / ow)).astype(int)
    return x[:, :, rows][:, :, :, cols]


class Params:
    """Weights drawn once, kept under the torch modules' state_dict names."""

    def __init__(self, rng):
        self.rng, self.store = rng, {}

    def conv_bn(self, name, cin, cout, k):
        r = self.rng
        self.store[name + '.conv.weight'] = r.normal(0, np.sqrt(2.0 / (cin * k * k)), size=(cout, cin, k, k))
        self.store[name + '.conv.bias'] = r.normal(0, 0.2, size=cout)
        self.store[name + '.bn.weight'] = r.uniform(0.5, 1.5, size=cout)
        self.store[name + '.bn.bias'] = r.normal(0, 0.3, size=cout)

    def layer(self, name, x, stride=1, activation=True):
        s = self.store
        y = batch_norm(conv2d_same(x, s[name + '.conv.weight'], s[name + '.conv.bias'], stride),
                       s[name + '.bn.weight'], s[name + '.bn.bias'])
        return leaky(y) if activation else y


def main():
    rng = np.random.Generator(np.random.PCG64(2024))
    out = {}
    # 1. one _conv_layer, 3x3 stride 1 and stride 2 on odd and even sizes, and 1x1 without activation
    p = Params(rng)
    x = rng.normal(0, 1, size=(2, 5, 9, 10))
    p.conv_bn('c3', 5, 7, 3)
    p.conv_bn('c3s2', 5, 6, 3)
    p.conv_bn('c1', 5, 4, 1)
    out['layer.x'] = x
    out['layer.c3.y'] = p.layer('c3', x)
    out['layer.c3s2.y'] = p.layer('c3s2', x, stride=2)
    out['layer.c1.y'] = p.layer('c1', x, activation=False)
    for k, v in p.store.items():
        out['layer.' + k] = v
    # 2. _darknet_block(filters 8, 2 residual units) on an odd grid
    p = Params(rng)
    x = rng.normal(0, 1, size=(2, 4, 11, 9))
    p.conv_bn('down', 4, 8, 3)
    for u in range(2):
        p.conv_bn('units.%d.0' % u, 8, 4, 1)
        p.conv_bn('units.%d.1' % u, 4, 8, 3)
    y = p.layer('down', x, stride=2)
    for u in range(2):
        y = y + p.layer('units.%d.1' % u, p.layer('units.%d.0' % u, y))
    out['block.x'], out['block.y'] = x, y
    for k, v in p.store.items():
        out['block.' + k] = v
    # 3. _yolo3_header(filters 8) with a pyramid input on a coarser grid (4 x 3 -> 7 x 6), 11 prediction channels
    p = Params(rng)
    bottom = rng.normal(0, 1, size=(2, 6, 7, 6))
    pyramid = rng.normal(0, 1, size=(2, 5, 4, 3))
    p.conv_bn('lateral', 5, 8, 1)
    cin = 6 + 8
    for name, ci, co, k in (('conv1', cin, 4, 1), ('conv2', 4, 8, 3), ('conv3', 8, 4, 1), ('conv4', 4, 8, 3),
                            ('conv5', 8, 4, 1), ('conv6', 4, 8, 3), ('pred', 8, 11, 1)):
        p.conv_bn(name, ci, co, k)
    up = resize_nearest(p.layer('lateral', pyramid, activation=False), 7, 6)
    h = np.concatenate([bottom, up], axis=1)
    for name in ('conv1', 'conv2', 'conv3', 'conv4', 'conv5'):
        h = p.layer(name, h)
    out['header.bottom'], out['header.pyramid'] = bottom, pyramid
    out['header.conv5'] = h
    out['header.pred'] = p.layer('pred', p.layer('conv6', h))
    for k, v in p.store.items():
        out['header.' + k] = v
    np.savez_compressed(os.path.join(HERE, 'conv_chain.npz'), **{k: np.asarray(v, np.float64) for k, v in out.items()})
    print('wrote conv_chain.npz: %d arrays' % len(out))


if __name__ == '__main__':
    main()
