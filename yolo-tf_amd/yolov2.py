"""YOLOv2 class API of the reference (YOLOv2.py:10-345) over the HIP
detection-head path: one 32x-downsampled grid, priors given in grid units,
ignore threshold 0.6, no division of the loss by the ground-truth count.

Head tensors come from config['head_provider'] (callable(images) -> pred,
NHWC float32 CUDA tensor [B, H, W, len(priors)*(num_classes+5)]) or, without a
provider, `images` is either that tensor already or a batch of images, for which
the reference's own convolution stack (backbone.Darknet19Head) is built on first
use.  data_provider['train_generator'] = (initializer callable or None, iterator
of (images_or_head, ground_truth)).  train_one_epoch takes an optimisation step
through the fused loss backward when the head tensor carries an autograd graph.
"""
import os
import sys

import numpy as np
import torch

from .head import DetectionHead
from . import parallel


class YOLOv2:
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
        self.nms_score_threshold = config['nms_score_threshold']
        self.nms_max_boxes = config['nms_max_boxes']
        self.nms_iou_threshold = config['nms_iou_threshold']
        self.rescore_confidence = config['rescore_confidence']      # read and unused, as in YOLOv2.py:32
        self.num_priors = len(config['priors'])
        self.priors = np.asarray(config['priors'], dtype=np.float32).reshape(1, 1, self.num_priors, 2)
        self.final_units = (self.num_classes + 5) * self.num_priors
        self.head_provider = config.get('head_provider')
        self.downsampling_rate = 32.0                               # YOLOv2.py:302

        if self.mode == 'train':
            self.num_train = data_provider['num_train']
            self.num_val = data_provider['num_val']
            self.train_generator = data_provider['train_generator']
            self.train_initializer, self.train_iterator = self.train_generator
            if data_provider['val_generator'] is not None:
                self.val_generator = data_provider['val_generator']
                self.val_initializer, self.val_iterator = self.val_generator

        self.global_step = 0
        self.is_training = True
        self.head = DetectionHead(self.num_classes, [self.priors.reshape(-1, 2)],
                                  box_mult=[self.downsampling_rate], gt_rate=[self.downsampling_rate])
        self.detection_pred = None
        self.loss = None
        if self.mode == 'train' and self.train_initializer is not None:
            self.train_initializer()

    def _get_priors(self, pshape, priors):
        """YOLOv2.py:230-244: abbox_yx, abbox_hw, abbox_y1x1, abbox_y2x2, [H,W,P,2]."""
        H, W = int(pshape[1]), int(pshape[2])
        priors = np.asarray(priors, dtype=np.float32).reshape(1, 1, -1, 2)
        P = priors.shape[2]
        ys, xs = np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32)
        abbox_yx = np.empty((H, W, P, 2), dtype=np.float32)
        abbox_yx[..., 0] = (ys + np.float32(0.5))[:, None, None]
        abbox_yx[..., 1] = (xs + np.float32(0.5))[None, :, None]
        abbox_hw = np.broadcast_to(priors, (H, W, P, 2)).astype(np.float32)
        return abbox_yx, abbox_hw, abbox_yx - abbox_hw / np.float32(2), abbox_yx + abbox_hw / np.float32(2)

    def _get_normlized_gn(self, downsampling_rate, i):
        """YOLOv2.py:246-253 on the last ground-truth batch seen."""
        gt = self.ground_truth[i]
        gt = np.asarray(gt.detach().cpu().numpy() if isinstance(gt, torch.Tensor) else gt, dtype=np.float32)
        n = int(np.argmin(gt[:, 0]))
        gn = gt[:n].copy()
        gn[:, :4] = gn[:, :4] / np.float32(downsampling_rate)
        return gn[..., :2], gn[..., 2:4], gn[..., 4].astype(np.int32)

    def _build_network(self):
        """_build_graph's convolution stack (YOLOv2.py:75-93): built on first use, when
        images rather than a head tensor arrive and no provider was configured."""
        from .backbone import Darknet19Head
        net = Darknet19Head(self.num_classes, self.num_priors, self.data_format, self.weight_decay)
        return net.to(torch.device('cuda', torch.cuda.current_device()))

    def _is_image_batch(self, x):
        shape = getattr(x, 'shape', None)
        if shape is None or len(shape) != 4:
            return False
        return shape[3 if self.data_format == 'channels_last' else 1] == 3      # final_units is never 3

    def _head_from(self, images):
        if self.head_provider is None and self._is_image_batch(images):
            self.head_provider = self._build_network()
        pred = self.head_provider(images) if self.head_provider is not None else images
        if isinstance(pred, (list, tuple)):
            if len(pred) != 1:
                raise ValueError('YOLOv2 has a single head tensor')
            pred = pred[0]
        if self.data_format == 'channels_first':
            pred = pred.permute(0, 2, 3, 1).contiguous()             # YOLOv2.py:92-93
        return pred

    def detect_from_heads(self, head, max_candidates=None):
        """detection_pred of YOLOv2.py:182-210 for every image of the batch."""
        return self.head.detect([head], self.nms_score_threshold, self.nms_iou_threshold,
                                self.nms_max_boxes, max_candidates)

    def loss_from_heads(self, head, ground_truth, want_parts=False):
        """total_loss of YOLOv2.py:104-174, mean over the (global) batch."""
        out = self.head.loss([head], ground_truth, self.coord_sacle, self.noobj_scale, self.obj_scale,
                             self.class_scale, ignore_threshold=0.6, normalize_by_num_gt=False,
                             want_parts=want_parts)
        total = parallel.global_mean_loss(out)
        return (total, out) if want_parts else total

    def _train_step(self, head, ground_truth, lr):
        """One optimisation step when the head tensor carries an autograd graph
        (optimizer.minimize, YOLOv2.py:175-181); see YOLOv3._train_step."""
        local_b = head.shape[0]
        out, grads = self.head.loss_and_grad([head.detach()], ground_truth, self.coord_sacle, self.noobj_scale,
                                             self.obj_scale, self.class_scale, 0.6, False,
                                             grad_scale=1.0 / parallel.global_batch(local_b))
        zero = getattr(self.head_provider, 'zero_grad', None)
        if zero is not None:
            zero()
        torch.autograd.backward([head], grads)               # d(total_loss) / d(head) into the network
        step = getattr(self.head_provider, 'step', None)
        if step is not None:
            step(lr)
        return parallel.global_mean_loss(out)

    def _weight_l2(self):
        fn = getattr(self.head_provider, 'l2_loss_sum', None)
        return float(fn()) if fn is not None else 0.0

    def _loss_prepare_ahead(self, images, ground_truth):
        """The ground-truth half of the loss call, issued before the network runs (yh_loss_prepare): when the
        batch is shaped like the last one the loss's tables are built while the convolutions run.  Returns the
        ground truth, on the device and in the form the loss call will find prepared."""
        last = getattr(self, '_last_step_shapes', None)
        if last is None or last[0] != tuple(getattr(images, 'shape', ())) or last[2].type != 'cuda':
            return ground_truth
        gt = ground_truth.to(last[2]).to(torch.float32).contiguous()
        if gt.dim() == 3 and gt.shape[0] == last[1][0][0] and gt.shape[2] == 5:
            self.head.loss_prepare(last[1], gt)
        return gt

    def train_one_epoch(self, lr, writer=None):
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
            head = self._head_from(images)
            ground_truth = ground_truth.to(head.device)
            self.ground_truth = ground_truth
            self._last_step_shapes = (tuple(getattr(images, 'shape', ())), [tuple(head.shape)], head.device)
            if head.requires_grad:
                total_loss = self._train_step(head, ground_truth, lr)
            else:                                            # a head without a graph: the loss value only
                total_loss = self.loss_from_heads(head, ground_truth)
            # YOLOv2.py:176-178: total_loss + weight_decay * sum(l2_loss(var))
            loss = np.float32(float(total_loss.item()) + self.weight_decay * self._weight_l2())
            self.loss = loss
            self.global_step += 1
            sys.stdout.write('\r>> ' + 'iters '+str(i+1)+str('/')+str(num_iters)+' loss '+str(loss))
            sys.stdout.flush()
            mean_loss.append(loss)
            if writer is not None:
                writer.add_summary(loss, global_step=self.global_step)
        sys.stdout.write('\n')
        mean_loss = np.mean(mean_loss)
        return mean_loss

    def test_one_image(self, images):
        self.is_training = False
        head = self._head_from(images)[:1]                          # YOLOv2.py:183-186: image 0 only
        scores, bbox, class_id, _ = self.detect_from_heads(head).per_image()[0]
        pred = [scores, bbox, class_id]
        self.detection_pred = pred
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
        print('>> save', mode, 'model in', path, 'successfully')

    def load_weight(self, path):
        self._provider_hook('load')(path)
        print('>> load weight', path, 'successfully')

    def load_pretraining_weight(self, path):
        self._provider_hook('load_pretraining')(path)
        print('>> load pretraining weight', path, 'successfully')
