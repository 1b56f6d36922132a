"""YOLOv3 class API of the reference (YOLOv3.py:11-485) over the HIP
detection-head path.

Kept from the reference: constructor signature and config / data_provider keys,
train_one_epoch(lr), test_one_image(images), save_weight / load_weight /
load_pretraining_weight, _get_priors and _get_normlized_gn.  The head tensors
come from a pluggable provider; when none is configured and a batch of images
arrives, the reference's own convolution stack (backbone.Darknet53Heads) is
built and used:

  * config['head_provider'] (optional): callable(images) -> [pred1, pred2,
    pred3], NHWC float32 CUDA tensors [B, H, W, num_priors*(num_classes+5)]
    (or NCHW when data_format == 'channels_first', transposed here exactly as
    YOLOv3.py:88-91 does).  Without a provider `images` must already be that
    list of head tensors.
  * data_provider['train_generator'] = (initializer, iterator): `initializer`
    is a callable (or None) run at the start of an epoch, `iterator` yields
    (images_or_heads, ground_truth[B, pad, 5]) batches; they stand where the
    reference has a tf.data initializer op and iterator.

train_one_epoch takes an optimisation step through the fused loss backward
whenever the head tensors carry an autograd graph (the provider's zero_grad /
step(lr) are the optimiser); with plain tensors it evaluates the loss only.
"""
import datetime
import os
import sys

import numpy as np
import torch

from .head import DetectionHead
from . import parallel


class YOLOv3:
    def __init__(self, config, data_provider):
        assert len(config['data_shape']) == 3
        assert config['mode'] in ['train', 'test']
        assert config['data_format'] in ['channels_first', 'channels_last']
        self.config = config
        self.data_provider = data_provider
        self.data_shape = config['data_shape']
        self.num_classes = config['num_classes']
        self.weight_decay = config['weight_decay']
        self.prob = 1. - config['keep_prob']
        self.data_format = config['data_format']
        self.mode = config['mode']
        self.batch_size = config['batch_size'] if config['mode'] == 'train' else 1

        self.coord_sacle = config['coord_scale']
        self.noobj_scale = config['noobj_scale']
        self.obj_scale = config['obj_scale']
        self.class_scale = config['class_scale']
        self.num_priors = config['num_priors']

        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']

        # YOLOv3.py:37-41: list entry i is divided by stride[i] = 8, 16, 32 and
        # entry 0 is the one used with the first (coarsest) head (YOLOv3.py:111-113)
        priors = config['priors']
        self.stride = [8., 16., 32.]
        self.priors = [(np.asarray(priors[i], dtype=np.float32) / np.float32(self.stride[i])).reshape(1, 1, -1, 2)
                       for i in range(len(priors))]
        self.final_units = (self.num_classes + 5) * self.num_priors
        self.head_provider = config.get('head_provider')

        if self.mode == 'train':
            self.num_train = data_provider['num_train']
            self.train_generator = data_provider['train_generator']
            self.train_initializer, self.train_iterator = self.train_generator
            if data_provider['val_generator'] is not None:
                self.num_val = data_provider['num_val']
                self.val_generator = data_provider['val_generator']
                self.val_initializer, self.val_iterator = self.val_generator

        self.global_step = 0
        self.is_training = True
        # decode multipliers stride[-1], stride[-1], stride[-2] (YOLOv3.py:346-348);
        # loss rates stride[-1], stride[-2], stride[-3] (YOLOv3.py:118-120)
        self.head = DetectionHead(
            self.num_classes, [p.reshape(-1, 2) for p in self.priors],
            box_mult=[self.stride[-1], self.stride[-1], self.stride[-2]],
            gt_rate=[self.stride[-1], self.stride[-2], self.stride[-3]])
        self.detection_pred = None
        self.loss = None
        if self.mode == 'train' and self.train_initializer is not None:
            self.train_initializer()

    # ------------------------------------------------------------ helpers
    def _get_priors(self, pshape, priors):
        """YOLOv3.py:419-433.  pshape = (B, H, W, ...); priors [1,1,P,2] in grid
        units.  Returns abbox_yx, abbox_hw, abbox_y1x1, abbox_y2x2 as float32
        arrays [H, W, P, 2]."""
        H, W = int(pshape[1]), int(pshape[2])
        priors = np.asarray(priors, dtype=np.float32).reshape(1, 1, -1, 2)
        P = priors.shape[2]
        tl = np.stack(np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32),
                                  indexing='ij'), axis=-1).reshape(H, W, 1, 2)
        abbox_yx = np.broadcast_to(tl + np.float32(0.5), (H, W, P, 2)).astype(np.float32)
        abbox_hw = np.broadcast_to(priors, (H, W, P, 2)).astype(np.float32)
        abbox_y1x1 = abbox_yx - abbox_hw / np.float32(2)
        abbox_y2x2 = abbox_yx + abbox_hw / np.float32(2)
        return abbox_yx, abbox_hw, abbox_y1x1, abbox_y2x2

    def _get_normlized_gn(self, downsampling_rate, i):
        """YOLOv3.py:435-442 on the last ground-truth batch seen."""
        gt = np.asarray(self.ground_truth[i].detach().cpu().numpy()
                        if isinstance(self.ground_truth, torch.Tensor) else self.ground_truth[i], dtype=np.float32)
        slice_index = int(np.argmin(gt[:, 0]))
        gn = gt[:slice_index] / np.array([downsampling_rate] * 4 + [1], dtype=np.float32).reshape(1, 5)
        return gn[..., :2], gn[..., 2:4], gn[..., 4].astype(np.int32)

    def _build_network(self):
        """_build_graph's convolution stack (YOLOv3.py:81-91): built on first use, when
        images rather than head tensors arrive and no provider was configured."""
        from .backbone import Darknet53Heads
        net = Darknet53Heads(self.num_classes, self.num_priors, self.data_format, self.weight_decay)
        return net.to(torch.device('cuda', torch.cuda.current_device()))

    def _heads_from(self, images):
        if self.head_provider is None and not isinstance(images, (list, tuple)):
            self.head_provider = self._build_network()
        preds = self.head_provider(images) if self.head_provider is not None else images
        if not isinstance(preds, (list, tuple)) or len(preds) != 3:
            raise ValueError('three head tensors (pred1, pred2, pred3) are required')
        preds = list(preds)
        if self.data_format != 'channels_last':
            preds = [p.permute(0, 2, 3, 1).contiguous() for p in preds]      # YOLOv3.py:88-91
        return preds

    # -------------------------------------------------- the path itself
    def detect_from_heads(self, heads, max_candidates=None):
        """detection_pred of YOLOv3.py:319-368 for every image of the batch;
        returns head.Detections (device resident)."""
        return self.head.detect(heads, self.nms_score_threshold, self.nms_iou_threshold,
                                self.nms_max_boxes, max_candidates)

    def loss_from_heads(self, heads, ground_truth, want_parts=False):
        """total_loss of YOLOv3.py:116-311 (reduce_mean of the per-image
        losses) as a 0-dim CUDA tensor; with torch.distributed initialised the
        mean is taken over the global batch (one scalar all-reduce)."""
        out = self.head.loss(heads, ground_truth, self.coord_sacle, self.noobj_scale, self.obj_scale,
                             self.class_scale, ignore_threshold=0.5, normalize_by_num_gt=True,
                             want_parts=want_parts)
        total = parallel.global_mean_loss(out)
        return (total, out) if want_parts else total

    def _train_step(self, heads, ground_truth, lr, loss_factor):
        """One optimisation step when the head tensors carry an autograd graph
        (optimizer.minimize, YOLOv3.py:312-318): backward of loss_factor *
        total_loss through the fused loss into the provider's network, then the
        provider's `step(lr)`.  The weight-decay term belongs to the provider,
        which owns the weights.  Returns total_loss (global mean, detached)."""
        local_b = heads[0].shape[0]
        out, grads = self.head.loss_and_grad([h.detach() for h in heads], ground_truth, self.coord_sacle,
                                             self.noobj_scale, self.obj_scale, self.class_scale, 0.5, True,
                                             grad_scale=loss_factor / parallel.global_batch(local_b))
        zero = getattr(self.head_provider, 'zero_grad', None)
        if zero is not None:
            zero()
        torch.autograd.backward(heads, grads)                # d(loss_factor * total_loss) / d(head) into the network
        step = getattr(self.head_provider, 'step', None)
        if step is not None:
            step(lr)
        return parallel.global_mean_loss(out)

    def _loss_prepare_ahead(self, images, ground_truth):
        """The ground-truth half of the loss call, issued before the network runs (yh_loss_prepare): the batch's
        ground truth goes to the device first, and when the batch is shaped like the last one (so the head
        tensors' shapes are known before they exist) the loss's tables are built while the convolutions run.
        Returns the ground truth, on the device and in the form the loss call will find prepared."""
        last = getattr(self, '_last_step_shapes', None)
        if last is None or last[0] != tuple(getattr(images, 'shape', ())) or last[2].type != 'cuda':
            return ground_truth
        gt = ground_truth.to(last[2]).to(torch.float32).contiguous()
        if gt.dim() == 3 and gt.shape[0] == last[1][0][0] and gt.shape[2] == 5:
            self.head.loss_prepare(last[1], gt)
        return gt

    def _weight_l2(self):
        fn = getattr(self.head_provider, 'l2_loss_sum', None)
        return float(fn()) if fn is not None else 0.0

    # ------------------------------------------------------- public API
    def train_one_epoch(self, lr):
        self.is_training = True
        if self.train_initializer is not None:
            self.train_initializer()
        mean_loss = []
        num_iters = self.num_train // self.batch_size
        for i in range(num_iters):
            images, ground_truth = next(self.train_iterator)
            if not isinstance(ground_truth, torch.Tensor):          # e.g. NumPy batches of records.get_generator
                ground_truth = torch.as_tensor(np.asarray(ground_truth, dtype=np.float32))
            ground_truth = self._loss_prepare_ahead(images, ground_truth)
            heads = self._heads_from(images)
            ground_truth = ground_truth.to(heads[0].device)
            self.ground_truth = ground_truth
            self._last_step_shapes = (tuple(getattr(images, 'shape', ())), [tuple(h.shape) for h in heads], heads[0].device)
            if any(h.requires_grad for h in heads):
                total_loss = self._train_step(heads, ground_truth, lr, 0.5)
            else:                                            # heads without a graph: the loss value only
                total_loss = self.loss_from_heads(heads, ground_truth)
            # YOLOv3.py:313-315: .5 * total_loss + weight_decay * sum(l2_loss(var))
            loss = np.float32(.5 * float(total_loss.item()) + self.weight_decay * self._weight_l2())
            self.loss = loss
            self.global_step += 1
            sys.stdout.write('\r>> ' + 'iters '+str(i)+str('/')+str(num_iters)+' loss '+str(loss))
            sys.stdout.flush()
            mean_loss.append(loss)
        sys.stdout.write('\n')
        mean_loss = np.mean(mean_loss)
        return mean_loss

    def test_one_image(self, images):
        self.is_training = False
        print(datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S.%f'))
        heads = self._heads_from(images)
        heads = [h[:1] for h in heads]                       # YOLOv3.py:320-331: image 0 only
        scores, bbox, class_id, _ = self.detect_from_heads(heads).per_image()[0]
        pred = [scores, bbox, class_id]
        self.detection_pred = pred
        print(datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S.%f'))
        return pred

    def _provider_hook(self, name):
        """The head provider's save / load / load_pretraining; the weights live there, not in this
        path, so without the hook nothing can be written or read and saying "successfully" would lie."""
        fn = getattr(self.head_provider, name, None)
        if fn is None:
            raise RuntimeError('the head provider %r has no %s(path): there are no weights to %s'
                               % (type(self.head_provider).__name__, name, name.split('_')[0]))
        return fn

    def save_weight(self, mode, path):
        assert(mode in ['latest', 'best'])
        if os.path.dirname(path) and not os.path.exists(os.path.dirname(path)):
            os.makedirs(os.path.dirname(path))
            print(os.path.dirname(path), 'does not exist, create it done')
        self._provider_hook('save')('%s-%d' % (path, self.global_step))
        print('save', mode, 'model in', path, 'successfully')

    def load_weight(self, path):
        self._provider_hook('load')(path)
        print('load weight', path, 'successfully')

    def load_pretraining_weight(self, path):
        self._provider_hook('load_pretraining')(path)
        print('load pretraining weight', path, 'successfully')
