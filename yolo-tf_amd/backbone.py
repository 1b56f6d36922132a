"""Convolution stacks that emit the head tensors of the path: Darknet-53 with the
three-level top-down header of YOLOv3 (YOLOv3.py:81-91, 387-417, 487-517) and
Darknet-19 with the pass-through header of YOLOv2 (YOLOv2.py:75-93, 255-303,
347-376), as torch.nn modules on MIOpen.

They implement the head-provider protocol of the YOLOv2 / YOLOv3 classes:
calling the object with a batch of images returns the NHWC head tensors the HIP
kernels read in place (the stack runs in channels_last memory format, so the
NHWC view of a prediction is contiguous and nothing is copied), and zero_grad /
step(lr) / l2_loss_sum / save / load / load_pretraining stand where the
reference has optimizer.minimize, the weight-decay term and its tf.train.Savers.

Reference behaviour kept on purpose:
  * every convolution is followed by batch normalisation (momentum 0.99,
    epsilon 1e-3: the tf.layers defaults), the prediction layers included; in
    YOLOv3 the prediction layer also passes through leaky ReLU (is_activation
    defaults to True, YOLOv3.py:416), in YOLOv2 it does not (YOLOv2.py:91);
  * batch normalisation always normalises with the statistics of the current
    batch: `training=self.is_training` is read once, while the graph is built,
    when it is True (YOLOv3.py:54, 515) - so also in test_one_image;
  * 'same' padding puts the odd pixel at the bottom / right (TensorFlow's rule),
    which differs from symmetric padding for the stride-2 convolutions;
  * the second header of YOLOv3 is built with 256 filters (YOLOv3.py:86).
Checkpoints are torch state_dicts: TensorFlow checkpoints cannot be read
without TensorFlow.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

IMAGE_MEAN = (123.68, 116.779, 103.979)        # YOLOv3.py:66 / YOLOv2.py:60
BN_EPSILON = 1e-3
BN_MOMENTUM = 0.99                             # TensorFlow's convention: moving = m * moving + (1 - m) * batch
LEAKY_SLOPE = 0.1


def same_padding(size, kernel, stride):
    """TensorFlow 'same' padding of one spatial dimension: (before, after)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    return total // 2, total - total // 2


def _pad_same(x, kernel, stride):
    top, bottom = same_padding(x.shape[2], kernel, stride)
    left, right = same_padding(x.shape[3], kernel, stride)
    if top or bottom or left or right:
        x = F.pad(x, (left, right, top, bottom))
    return x


def _variance_scaling_(weight):
    """tf.contrib.layers.variance_scaling_initializer() defaults (factor 2, FAN_IN,
    truncated normal): stddev sqrt(1.3 * 2 / fan_in), samples beyond two stddev redrawn."""
    fan_in = weight.shape[1] * weight.shape[2] * weight.shape[3]
    std = math.sqrt(1.3 * 2.0 / fan_in)
    nn.init.trunc_normal_(weight, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std)


class ConvBN(nn.Module):
    """conv2d('same', bias) -> batch normalisation -> optional leaky ReLU(0.1):
    _conv_layer of YOLOv3.py:496-509; YOLOv2.py:355-366 without the activation,
    which YOLOv2 applies at the call site."""

    def __init__(self, cin, cout, kernel, stride=1, activation=True, init='variance_scaling'):
        super().__init__()
        self.kernel, self.stride, self.activation = kernel, stride, activation
        self.conv = nn.Conv2d(cin, cout, kernel, stride, padding=0, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=BN_EPSILON, momentum=1.0 - BN_MOMENTUM)
        if init == 'variance_scaling':
            _variance_scaling_(self.conv.weight)
        else:                                               # tf.layers.conv2d default: glorot_uniform
            nn.init.xavier_uniform_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        x = self.bn(self.conv(_pad_same(x, self.kernel, self.stride)))
        return F.leaky_relu(x, LEAKY_SLOPE) if self.activation else x


class DarknetBlock(nn.Module):
    """_darknet_block (YOLOv3.py:487-494): a stride-2 3x3 convolution, then `blocks`
    residual units of 1x1 (filters / 2) and 3x3 (filters)."""

    def __init__(self, cin, filters, blocks):
        super().__init__()
        self.down = ConvBN(cin, filters, 3, 2)
        self.units = nn.ModuleList(nn.Sequential(ConvBN(filters, filters // 2, 1), ConvBN(filters // 2, filters, 3))
                                   for _ in range(blocks))

    def forward(self, x):
        x = self.down(x)
        for unit in self.units:
            x = x + unit(x)
        return x


class Yolo3Header(nn.Module):
    """_yolo3_header (YOLOv3.py:396-417).  With a pyramid input: 1x1 convolution of it
    (batch-normalised, no activation), nearest-neighbour resize to the bottom's grid,
    concatenation [bottom, resized]."""

    def __init__(self, bottom_channels, filters, final_units, pyramid_channels=None):
        super().__init__()
        cin = bottom_channels
        self.lateral = None
        if pyramid_channels is not None:
            self.lateral = ConvBN(pyramid_channels, filters, 1, 1, activation=False)
            cin += filters
        half = filters // 2
        self.conv1 = ConvBN(cin, half, 1)
        self.conv2 = ConvBN(half, filters, 3)
        self.conv3 = ConvBN(filters, half, 1)
        self.conv4 = ConvBN(half, filters, 3)
        self.conv5 = ConvBN(filters, half, 1)
        self.conv6 = ConvBN(half, filters, 3)
        self.pred = ConvBN(filters, final_units, 1)
        self.top_down_channels = half

    def forward(self, bottom, pyramid=None):
        x = bottom
        if self.lateral is not None:
            up = F.interpolate(self.lateral(pyramid), size=bottom.shape[2:], mode='nearest')
            x = torch.cat([bottom, up], dim=1)
        conv5 = self.conv5(self.conv4(self.conv3(self.conv2(self.conv1(x)))))
        return self.pred(self.conv6(conv5)), conv5


class _Provider(nn.Module):
    """What the two stacks share: image intake, optimiser, checkpoints."""

    def __init__(self, data_format, weight_decay, momentum):
        super().__init__()
        assert data_format in ('channels_first', 'channels_last')
        self.data_format = data_format
        self.weight_decay = float(weight_decay)
        self.momentum = float(momentum)
        self.register_buffer('image_mean', torch.tensor(IMAGE_MEAN, dtype=torch.float32).reshape(1, 3, 1, 1))
        self._velocity = None

    def _finish_init(self):
        self.to(memory_format=torch.channels_last)
        self.train()                        # batch statistics always, as in the reference's graph

    def _intake(self, images):
        """images: array or tensor [B, H, W, 3] (channels_last) or [B, 3, H, W]
        (channels_first), any real dtype.  Returns the mean-subtracted NCHW tensor
        in channels_last memory on the module's device."""
        x = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
        x = x.to(device=self.image_mean.device, dtype=torch.float32)
        if x.dim() != 4:
            raise ValueError('images must be a 4-d batch, got shape %s' % (tuple(x.shape),))
        if self.data_format == 'channels_last':
            x = x.permute(0, 3, 1, 2)                       # a view: NHWC memory is channels_last NCHW
        if x.shape[1] != 3:
            raise ValueError('images must have 3 channels in the %s position' % self.data_format)
        return (x - self.image_mean).contiguous(memory_format=torch.channels_last)

    def _emit(self, pred):
        """A prediction (NCHW, channels_last memory) as the caller's data_format: the
        NHWC view for channels_last (contiguous, nothing moves), NCHW otherwise (the
        class transposes it as YOLOv3.py:88-91 does)."""
        pred = pred.contiguous(memory_format=torch.channels_last)
        return pred.permute(0, 2, 3, 1) if self.data_format == 'channels_last' else pred

    # --- optimizer.minimize of the reference: MomentumOptimizer(lr, 0.9) on
    #     loss + weight_decay * sum(l2_loss(var)) (YOLOv3.py:312-316, YOLOv2.py:175-179)
    @torch.no_grad()
    def step(self, lr):
        params = [p for p in self.parameters() if p.requires_grad]
        if self._velocity is None:
            self._velocity = [torch.zeros_like(p) for p in params]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # data parallelism: the loss was divided by the global batch, so the sum of the ranks'
            # gradients is the gradient of the global mean; one all-reduce over one flat bucket
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            grads = [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)]
        for p, v, g in zip(params, self._velocity, grads):
            g = g + self.weight_decay * p                    # d/dp of weight_decay * sum(p^2) / 2
            v.mul_(self.momentum).add_(g)                    # accum = momentum * accum + grad
            p.sub_(v, alpha=float(lr))                       # var -= lr * accum

    @torch.no_grad()
    def l2_loss_sum(self):
        """sum over the trainable variables of tf.nn.l2_loss(var) = sum(var^2) / 2."""
        return float(sum((p.double() ** 2).sum() for p in self.parameters() if p.requires_grad) / 2)

    # --- tf.train.Saver stand-ins
    def save(self, path):
        if os.path.dirname(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location=self.image_mean.device, weights_only=True))

    def load_pretraining(self, path):
        """Restores the feature extractor only (scope 'backone', YOLOv3.py:377-378)."""
        state = torch.load(path, map_location=self.image_mean.device, weights_only=True)
        own = self.state_dict()
        picked = {k: v for k, v in state.items() if k.startswith('backone.') and k in own}
        if not picked:
            raise ValueError('%s holds no feature-extractor weights' % path)
        own.update(picked)
        self.load_state_dict(own)


class Darknet53Heads(_Provider):
    """images -> [pred1 (stride 32), pred2 (16), pred3 (8)], each
    [B, H, W, num_priors * (num_classes + 5)] (YOLOv3.py:81-91)."""

    def __init__(self, num_classes, num_priors, data_format='channels_last', weight_decay=0.0, momentum=0.9):
        super().__init__(data_format, weight_decay, momentum)
        final_units = (num_classes + 5) * num_priors
        # scope 'backone' (sic, YOLOv3.py:82): _feature_extractor, YOLOv3.py:387-394
        self.backone = nn.ModuleDict(dict(
            init_conv=ConvBN(3, 32, 3, 1),
            block1=DarknetBlock(32, 64, 1),
            block2=DarknetBlock(64, 128, 2),
            block3=DarknetBlock(128, 256, 8),
            block4=DarknetBlock(256, 512, 8),
            block5=DarknetBlock(512, 1024, 4)))
        pyd1 = Yolo3Header(1024, 1024, final_units)
        pyd2 = Yolo3Header(512, 256, final_units, pyramid_channels=pyd1.top_down_channels)
        pyd3 = Yolo3Header(256, 128, final_units, pyramid_channels=pyd2.top_down_channels)
        self.head = nn.ModuleDict(dict(pyd1=pyd1, pyd2=pyd2, pyd3=pyd3))
        self._finish_init()

    def forward(self, images):
        x = self.backone['init_conv'](self._intake(images))
        x = self.backone['block2'](self.backone['block1'](x))
        block3 = self.backone['block3'](x)
        block4 = self.backone['block4'](block3)
        block5 = self.backone['block5'](block4)
        pred1, top_down = self.head['pyd1'](block5)
        pred2, top_down = self.head['pyd2'](block4, top_down)
        pred3, _ = self.head['pyd3'](block3, top_down)
        return [self._emit(pred1), self._emit(pred2), self._emit(pred3)]


class Darknet19Head(_Provider):
    """images -> pred [B, H, W, num_priors * (num_classes + 5)] at stride 32
    (YOLOv2.py:75-93, 255-303)."""

    # (filters, kernel) per convolution; 'M' = 2x2 max pooling, stride 2, 'same'
    FEATURES = [(32, 3), 'M', (64, 3), 'M', (128, 3), (64, 1), (128, 3), 'M', (256, 3), (128, 1), (256, 3), 'M',
                (512, 3), (256, 1), (512, 3), (256, 1), (512, 3), 'M',
                (1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]
    HEAD = [(1024, 3), (512, 1), (1024, 3), (512, 1), (1024, 3)]

    def __init__(self, num_classes, num_priors, data_format='channels_last', weight_decay=0.0, momentum=0.9):
        super().__init__(data_format, weight_decay, momentum)
        final_units = (num_classes + 5) * num_priors
        layers, cin = [], 3
        for spec in self.FEATURES:
            if spec == 'M':
                layers.append(None)
            else:
                layers.append(ConvBN(cin, spec[0], spec[1], 1, activation=True, init='glorot'))
                cin = spec[0]
        self.pool_at = [i for i, l in enumerate(layers) if l is None]
        self.backone = nn.ModuleList(l for l in layers if l is not None)
        self.layout = ['M' if l is None else 'C' for l in layers]
        head, passthrough_channels = [], 512                  # lrelu17, the second-last feature layer
        for filters, kernel in self.HEAD:
            head.append(ConvBN(cin, filters, kernel, 1, activation=True, init='glorot'))
            cin = filters
        self.head = nn.ModuleList(head)
        self.predictions = ConvBN(cin + passthrough_channels, final_units, 1, 1, activation=False, init='glorot')
        self._finish_init()

    def forward(self, images):
        x = self._intake(images)
        convs = iter(self.backone)
        outputs = []
        for kind in self.layout:
            if kind == 'M':
                x = F.max_pool2d(_pad_same_pool(x), 2, 2)
            else:
                x = next(convs)(x)
                outputs.append(x)
        passthrough = outputs[-2]                             # YOLOv2.py:303: (lrelu18, lrelu17)
        for conv in self.head:
            x = conv(x)
        x = torch.cat([passthrough, x], dim=1)                # YOLOv2.py:90
        return [self._emit(self.predictions(x))]


def _pad_same_pool(x):
    """'same' padding of the 2x2 / stride-2 max pooling: an odd side gains one row or
    column of -inf at the bottom / right."""
    bottom, right = x.shape[2] & 1, x.shape[3] & 1
    return F.pad(x, (0, right, 0, bottom), value=float('-inf')) if (bottom or right) else x
