"""Host side of the detection-head path: owns device buffers (through torch),
describes the head geometry and calls libyolohead through ctypes.

torch is plumbing here (device memory, streams); every computation happens in
the HIP kernels behind the C ABI.  Without the shared library or without a HIP
device these functions raise; nothing falls back to the CPU."""
import contextlib
import ctypes

import numpy as np
import torch

from . import _abi


def make_config(grids, priors, box_mult, gt_rate, num_classes):
    """Builds struct yh_head_config.  grids: [(H, W)] per scale; priors: per
    scale [P][2] (h, w) in grid units."""
    S = len(grids)
    if not (len(priors) == len(box_mult) == len(gt_rate) == S):
        raise ValueError('grids, priors, box_mult and gt_rate must have one entry per scale')
    if S < 1 or S > _abi.YH_MAX_SCALES:
        raise ValueError('1..%d scales supported' % _abi.YH_MAX_SCALES)
    P = len(priors[0])
    if P < 1 or P > _abi.YH_MAX_PRIORS:
        raise ValueError('1..%d priors per scale supported' % _abi.YH_MAX_PRIORS)
    cfg = _abi.HeadConfig()
    cfg.num_scales = S
    cfg.num_priors = P
    cfg.num_classes = int(num_classes)
    for s in range(S):
        pr = np.asarray(priors[s], dtype=np.float32).reshape(-1, 2)
        if pr.shape[0] != P:
            raise ValueError('every scale must have the same number of priors')
        cfg.grid_h[s] = int(grids[s][0])
        cfg.grid_w[s] = int(grids[s][1])
        cfg.box_mult[s] = float(box_mult[s])
        cfg.gt_rate[s] = float(gt_rate[s])
        for p in range(P):
            cfg.prior_hw[s][p][0] = float(pr[p, 0])
            cfg.prior_hw[s][p][1] = float(pr[p, 1])
    return cfg


def _allocating_on(stream):
    """Allocations (and the zero fill of a new workspace) for launches that go to `stream`: made with that
    stream current, so that the fill is ordered before the launches and the caching allocator knows which
    stream uses the blocks.  (Filled on the caller's current stream, a new workspace could still hold what
    the block's last owner left in it when a launch on another stream read its counters: seen once in some
    thousand first calls of a DetectLanes as a stale key at the head of every class's selection.)"""
    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


class Detections:
    """Device-resident result of DetectionHead.detect / .nms for a batch."""

    def __init__(self, count, scores, boxes, class_id, index, class_candidates, status, rerun, stream=None):
        self.count = count                  # [B] int32
        self.scores = scores                # [B, C*max_boxes] float32, first count[b] valid
        self.boxes = boxes                  # [B, C*max_boxes, 4] float32 (y1, x1, y2, x2)
        self.class_id = class_id            # [B, C*max_boxes] int32
        self.index = index                  # [B, C*max_boxes] int32: box row of each detection
        self.class_candidates = class_candidates   # [B, C] int32 or None
        self.status = status                # [B] int32 per-image device status words
        self._rerun = rerun
        self._stream = stream               # the stream the producing launch was enqueued on, when not the current one

    def synchronize(self):
        """Waits for the launch that writes these tensors when it went to a stream of its own
        (a pipeline's or a lane's); reading them from the current stream before would race."""
        if self._stream is not None:
            self._stream.synchronize()

    def overflowed(self):
        """Synchronises and reports whether an image exceeded max_candidates."""
        self.synchronize()
        if self.status.numel() == 0:
            return False
        return bool(int((self.status & _abi.YH_STATUS_CANDIDATE_OVERFLOW).max().item()))

    def needed_candidates(self):
        """Synchronises; the largest candidate count an overflowed image reported (a multiple
        of 256, from the status words), 0 when no image overflowed."""
        self.synchronize()
        if self.status.numel() == 0:
            return 0
        st = self.status[(self.status & _abi.YH_STATUS_CANDIDATE_OVERFLOW) != 0]
        if st.numel() == 0:
            return 0
        return int((st >> _abi.YH_STATUS_COUNT_SHIFT).max().item()) * 256

    def resolved(self):
        """Synchronises; while the candidate capacity overflows, reruns with one sized by the
        counts the overflowed images reported (up to the worst case N*C) and returns the
        final result."""
        r = self
        while r.overflowed():
            if r._rerun is None:
                raise RuntimeError('candidate capacity overflow; raise max_candidates')
            r = r._rerun(r.needed_candidates())
        return r

    def per_image(self):
        """[(scores [K], bbox [K,4], class_id [K], index [K])] as NumPy arrays,
        the reference's detection_pred layout (YOLOv3.py:365-368) plus the box row."""
        r = self.resolved()
        count = r.count.cpu().numpy()
        scores = r.scores.cpu().numpy(); boxes = r.boxes.cpu().numpy()
        cls = r.class_id.cpu().numpy(); idx = r.index.cpu().numpy()
        out = []
        for b in range(count.shape[0]):
            k = int(count[b])
            out.append((scores[b, :k].copy(), boxes[b, :k].copy(), cls[b, :k].copy(), idx[b, :k].copy()))
        return out


class DetectionHead:
    """The reference's head geometry (priors, multipliers, rates) bound to the
    HIP kernels.  heads are NHWC float32 CUDA tensors [B, H, W, P*(C+5)]."""

    def __init__(self, num_classes, priors, box_mult, gt_rate):
        self.num_classes = int(num_classes)
        self.priors = [np.asarray(p, dtype=np.float32).reshape(-1, 2) for p in priors]
        self.num_priors = self.priors[0].shape[0]
        self.box_mult = [float(m) for m in box_mult]
        self.gt_rate = [float(r) for r in gt_rate]
        self._ws = {}
        self._configs = {}
        self._grad_weights = {}
        self._plans = {}
        self._grown = {}

    # ------------------------------------------------------------ helpers
    def _check_heads(self, heads):
        if len(heads) != len(self.priors):
            raise ValueError('%d head tensors given, %d scales configured' % (len(heads), len(self.priors)))
        K = self.num_priors * (self.num_classes + 5)
        B = heads[0].shape[0]
        dev = heads[0].device
        out = []
        for h in heads:
            if not isinstance(h, torch.Tensor):
                raise TypeError('heads must be torch tensors on a HIP device')
            if h.device.type != 'cuda':
                raise RuntimeError('head tensors must live on a HIP device (got %s); there is no CPU path' % h.device)
            if h.dtype != torch.float32 or h.dim() != 4 or h.shape[3] != K or h.shape[0] != B or h.device != dev:
                raise ValueError('every head must be float32 [B, H, W, %d] on one device' % K)
            out.append(h.contiguous())
        return out

    def config_for_shapes(self, shapes):
        """The yh_head_config of head tensors of these shapes ([B, H, W, P*(C+5)] each)."""
        return self._config_for_grids(tuple((int(s[1]), int(s[2])) for s in shapes))

    def config_for(self, heads):
        return self._config_for_grids(tuple((int(h.shape[1]), int(h.shape[2])) for h in heads))

    def _config_for_grids(self, grids):
        cfg = self._configs.get(grids)
        if cfg is None:
            cfg = self._configs[grids] = make_config(list(grids), self.priors, self.box_mult, self.gt_rate,
                                                     self.num_classes)
        return cfg

    def boxes_per_image(self, heads):
        return sum(int(h.shape[1]) * int(h.shape[2]) * self.num_priors for h in heads)

    def _workspace(self, key, nbytes, device):
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            # zero-filled once: the library keeps its counters clean from call to call
            ws = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    @staticmethod
    def _stream(device):
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    # ------------------------------------------------------------- decode
    def decode(self, heads, want_conf=True):
        """bbox [B,N,4] and confidence [B,N,C] of YOLOv3.py:338-350."""
        lib = _abi.load()
        heads = self._check_heads(heads)
        cfg = self.config_for(heads)
        B, N, C = heads[0].shape[0], self.boxes_per_image(heads), self.num_classes
        dev = heads[0].device
        boxes = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        conf = torch.empty((B, N, C), dtype=torch.float32, device=dev) if want_conf else None
        with torch.cuda.device(dev):
            _abi.check(lib.yh_decode(ctypes.byref(cfg), _abi.pointer_array([h.data_ptr() for h in heads]), B,
                                     boxes.data_ptr(), conf.data_ptr() if want_conf else None, self._stream(dev)))
        return boxes, conf

    # ------------------------------------------------------------- detect
    SMALL_CAPACITY = 512       # default capacity: the single-launch stage (<= 2048) with a small LDS footprint

    def default_max_candidates(self, N):
        """Per-image candidate capacity used when the caller gives none: the
        single-launch capacity, or whatever an earlier overflow on this head
        geometry grew it to (Detections.resolved)."""
        full = N * self.num_classes
        return int(min(full, self._grown.get(N, self.SMALL_CAPACITY)))

    @staticmethod
    def _grow(cap, needed, full):
        """The capacity of a second call after an overflow: a quarter above what the fullest
        image reported (rounded up to 2048: whole slices of the class binning), at least twice
        the old one, at most every (box, class) pair."""
        want = max(2 * cap, (needed + needed // 4 + 2047) // 2048 * 2048)
        return int(min(full, want))

    def _alloc_outputs(self, B, max_boxes, dev, want_class_candidates):
        C = self.num_classes
        cap = C * max_boxes
        count = torch.empty((B,), dtype=torch.int32, device=dev)
        scores = torch.empty((B, cap), dtype=torch.float32, device=dev)
        boxes = torch.empty((B, cap, 4), dtype=torch.float32, device=dev)
        cls = torch.empty((B, cap), dtype=torch.int32, device=dev)
        idx = torch.empty((B, cap), dtype=torch.int32, device=dev)
        cc = torch.empty((B, C), dtype=torch.int32, device=dev) if want_class_candidates else None
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        return count, scores, boxes, cls, idx, cc, status

    def _detect_plan(self, heads, cap, max_boxes, slot=0):
        """(cfg, B, N, workspace) for these head shapes and bounds, planned once;
        `slot` distinguishes workspaces of batches that are in flight together."""
        dev = heads[0].device
        key = ('detect', tuple(tuple(h.shape) for h in heads), cap, int(max_boxes), dev, slot, _abi.tuning_epoch)
        plan = self._plans.get(key)
        if plan is None:
            lib = _abi.load()
            cfg = self.config_for(heads)
            B, N = heads[0].shape[0], self.boxes_per_image(heads)
            nbytes = ctypes.c_size_t(0)
            _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, int(max_boxes), ctypes.byref(nbytes)))
            ws = self._workspace(('detect', B, N, cap, int(max_boxes), slot), nbytes.value, dev)
            plan = self._plans[key] = (cfg, B, N, ws)
        return plan

    @staticmethod
    def _output_pointers(outputs):
        count, scores, boxes, cls, idx, cc, status = outputs
        return [count.data_ptr(), scores.data_ptr(), boxes.data_ptr(), cls.data_ptr(), idx.data_ptr(),
                cc.data_ptr() if cc is not None else None, status.data_ptr()]

    def detect(self, heads, score_threshold, iou_threshold, max_boxes, max_candidates=None,
               want_class_candidates=False, outputs=None, stream_events=None, _auto=False, stream=None):
        """Decode + score filter + per-class NMS for every image (YOLOv3.py:320-368).
        Enqueues on `stream` (a torch.cuda.Stream; default: the current stream)
        and returns device tensors; nothing synchronises until the result is
        read.  outputs: a tuple from _alloc_outputs to write into (reused across
        calls by the caller).  One head object owns one workspace per shape:
        calls that may overlap on the device (different streams) need different
        DetectionHead objects.
        stream_events: a pair of torch.cuda.Event (timing enabled, recorded once
        before) bracketing the streaming kernel."""
        lib = _abi.load()
        heads = self._check_heads(heads)
        dev = heads[0].device
        N = self.boxes_per_image(heads)
        cap = int(max_candidates) if max_candidates is not None else self.default_max_candidates(N)
        with _allocating_on(stream):
            cfg, B, N, ws = self._detect_plan(heads, cap, max_boxes)
            if outputs is None:
                outputs = self._alloc_outputs(B, int(max_boxes), dev, want_class_candidates)
        args = [ctypes.byref(cfg), _abi.pointer_array([h.data_ptr() for h in heads]), B,
                float(score_threshold), float(iou_threshold), int(max_boxes), cap,
                ws.data_ptr(), ws.numel()] + self._output_pointers(outputs) + [
                self._stream(dev) if stream is None else ctypes.c_void_p(stream.cuda_stream)]
        with torch.cuda.device(dev):
            if stream_events is None:
                _abi.check(lib.yh_detect(*args))
            else:
                _abi.check(lib.yh_detect_timed(*args, ctypes.c_void_p(stream_events[0].cuda_event),
                                               ctypes.c_void_p(stream_events[1].cuda_event)))
        full = N * self.num_classes
        rerun = None
        if cap < full:
            auto = _auto or max_candidates is None

            def rerun(needed):
                grown = self._grow(cap, needed, full)
                if auto:                                    # remembered for later calls on this geometry
                    self._grown[N] = max(grown, self._grown.get(N, 0))
                return self.detect(heads, score_threshold, iou_threshold, max_boxes, grown,
                                   want_class_candidates, _auto=auto, stream=stream)
        return Detections(*outputs, rerun, stream=stream)

    def detect_static(self, heads, score_threshold, iou_threshold, max_boxes, max_candidates, outputs,
                      stream=None):
        """detect() through yh_detect_static: no workspace of the caller's and no allocation by the
        call (`outputs` from _alloc_outputs are the caller's, made beforehand) — for code under HIP
        graph capture or serving loops that must not allocate.  The library's scratch array is shared
        by every such call on the GPU: calls must not overlap in time.  Capacity is fixed by the
        caller; an overflow is reported in the status words and not rerun here."""
        lib = _abi.load()
        heads = self._check_heads(heads)
        dev = heads[0].device
        cfg = self.config_for(heads)
        args = [ctypes.byref(cfg), _abi.pointer_array([h.data_ptr() for h in heads]), heads[0].shape[0],
                float(score_threshold), float(iou_threshold), int(max_boxes), int(max_candidates)] + \
            self._output_pointers(outputs) + [self._stream(dev) if stream is None else ctypes.c_void_p(stream.cuda_stream)]
        with torch.cuda.device(dev):
            _abi.check(lib.yh_detect_static(*args))
        return Detections(*outputs, None, stream=stream)

    def nms(self, boxes, conf, score_threshold, iou_threshold, max_boxes, max_candidates=None,
            want_class_candidates=False, _auto=False, outputs=None, stream_events=None):
        """Score filter + per-class NMS on decoded inputs (YOLOv3.py:351-368).
        boxes [B,N,4], conf [B,N,C] float32 CUDA tensors.  outputs, stream_events: as in detect()
        (the events bracket the pass over the confidence matrix)."""
        lib = _abi.load()
        if boxes.device.type != 'cuda' or conf.device != boxes.device:
            raise RuntimeError('boxes and conf must live on one HIP device; there is no CPU path')
        boxes = boxes.contiguous(); conf = conf.contiguous()
        B, N, C = conf.shape
        if C != self.num_classes or boxes.shape != (B, N, 4):
            raise ValueError('expected boxes [B,N,4] and conf [B,N,%d]' % self.num_classes)
        dev = boxes.device
        full = N * C
        cap = int(max_candidates) if max_candidates is not None else self.default_max_candidates(N)
        nbytes = ctypes.c_size_t(0)
        _abi.check(lib.yh_nms_workspace_bytes(B, N, C, cap, int(max_boxes), ctypes.byref(nbytes)))
        ws = self._workspace(('nms', B, N, cap, int(max_boxes)), nbytes.value, dev)
        if outputs is None:
            outputs = self._alloc_outputs(B, int(max_boxes), dev, want_class_candidates)
        count, scores, oboxes, cls, idx, cc, status = outputs
        args = [B, N, C, boxes.data_ptr(), conf.data_ptr(), float(score_threshold), float(iou_threshold),
                int(max_boxes), cap, ws.data_ptr(), ws.numel()] + self._output_pointers(outputs) + [self._stream(dev)]
        with torch.cuda.device(dev):
            if stream_events is None:
                _abi.check(lib.yh_nms(*args))
            else:
                _abi.check(lib.yh_nms_timed(*args, ctypes.c_void_p(stream_events[0].cuda_event),
                                            ctypes.c_void_p(stream_events[1].cuda_event)))
        rerun = None
        if cap < full:
            auto = _auto or max_candidates is None

            def rerun(needed):
                grown = self._grow(cap, needed, full)
                if auto:
                    self._grown[N] = max(grown, self._grown.get(N, 0))
                return self.nms(boxes, conf, score_threshold, iou_threshold, max_boxes, grown,
                                want_class_candidates, _auto=auto)
        return Detections(count, scores, oboxes, cls, idx, cc, status, rerun)

    # --------------------------------------------------------------- loss
    def loss_prepare(self, head_shapes, ground_truth):
        """yh_loss_prepare: the ground-truth half of the next loss() / loss_and_grad() call for heads of these
        shapes ([B, H, W, P*(C+5)] per scale; tensors are accepted too) and this very ground_truth tensor
        ([B, pad, 5] float32, contiguous, on the device), issued on the current stream.  Optional: a training
        loop calls it as soon as the batch's ground truth is on the device, before the convolutions run."""
        lib = _abi.load()
        shapes = [tuple(h.shape) if isinstance(h, torch.Tensor) else tuple(int(v) for v in h) for h in head_shapes]
        gt = ground_truth
        if not (isinstance(gt, torch.Tensor) and gt.dtype is torch.float32 and gt.is_contiguous() and gt.is_cuda):
            raise ValueError('loss_prepare takes the float32, contiguous device tensor that loss() will be given')
        B = shapes[0][0]
        if gt.dim() != 3 or gt.shape[0] != B or gt.shape[2] != 5:
            raise ValueError('ground_truth must be [B, pad, 5]')
        cfg = self.config_for_shapes(shapes)
        dev, pad = gt.device, int(gt.shape[1])
        wkey = ('loss', tuple(shapes), pad)
        ws = self._ws.get(wkey)
        if ws is None or ws.device != dev:
            nbytes = ctypes.c_size_t(0)
            _abi.check(lib.yh_loss_workspace_bytes(ctypes.byref(cfg), B, pad, ctypes.byref(nbytes)))
            ws = self._workspace(wkey, nbytes.value, dev)
        with torch.cuda.device(dev):
            _abi.check(lib.yh_loss_prepare(ctypes.byref(cfg), gt.data_ptr(), B, pad, ws.data_ptr(), ws.numel(), self._stream(dev)))

    def loss(self, heads, ground_truth, coord_scale, noobj_scale, obj_scale, class_scale,
             ignore_threshold, normalize_by_num_gt, want_parts=False, want_iou_max=False, want_mask=False,
             out=None, grad_per_image=None):
        """Per-image training loss (YOLOv3.py:117-310 / YOLOv2.py:105-173).
        ground_truth: [B, pad, 5] float32 CUDA tensor, -1 padded.
        Returns dict(per_image [B], sum [1], mean [1] (sum / B), parts [B,4] | None, iou_max [B,N] | None,
        noobj_mask [B,N] uint8 | None: what loss_backward needs).  No autograd
        graph is recorded here: loss_with_grad does that.  out: the dict of an
        earlier call with the same shapes and options, whose tensors are then
        written again instead of new ones being allocated (a tight loop saves the
        allocations' host time).  grad_per_image [B]: the gradient
        d(sum_b grad_per_image[b] * loss_b) / d(head) is computed in the same pass
        (yh_loss_with_grad) and returned under 'grads'; no noobj_mask then."""
        lib = _abi.load()
        heads = self._check_heads(heads)
        cfg = self.config_for(heads)
        dev = heads[0].device
        if grad_per_image is not None and want_mask:
            raise ValueError('the fused gradient pass records no noobj_mask')
        if ground_truth.device != dev:
            raise RuntimeError('ground_truth must live on the heads\' device')
        gt = ground_truth.to(torch.float32).contiguous()
        B, N = heads[0].shape[0], self.boxes_per_image(heads)
        if gt.dim() != 3 or gt.shape[0] != B or gt.shape[2] != 5:
            raise ValueError('ground_truth must be [B, pad, 5]')
        pad = int(gt.shape[1])
        # keyed by the grid shapes: the size depends on the per-scale split of the rows, not on N alone
        wkey = ('loss', tuple(tuple(h.shape) for h in heads), pad)
        ws = self._ws.get(wkey)
        if ws is None or ws.device != dev:
            nbytes = ctypes.c_size_t(0)
            _abi.check(lib.yh_loss_workspace_bytes(ctypes.byref(cfg), B, pad, ctypes.byref(nbytes)))
            ws = self._workspace(wkey, nbytes.value, dev)
        if (out is not None and out['per_image'].shape == (B,) and out['per_image'].device == dev and
                (out['parts'] is not None) == want_parts and (out['iou_max'] is not None) == want_iou_max and
                (out['noobj_mask'] is not None) == want_mask and
                (not want_mask or out['noobj_mask'].shape == (B, N)) and (not want_iou_max or out['iou_max'].shape == (B, N))):
            per, total, parts, iou, mask = out['per_image'], out['_total'], out['parts'], out['iou_max'], out['noobj_mask']
        else:
            out = None
            per = torch.empty((B,), dtype=torch.float32, device=dev)
            total = torch.empty((2,), dtype=torch.float32, device=dev)      # sum, mean over this call's images
            parts = torch.empty((B, 4), dtype=torch.float32, device=dev) if want_parts else None
            iou = torch.empty((B, N), dtype=torch.float32, device=dev) if want_iou_max else None
            mask = torch.empty((B, N), dtype=torch.uint8, device=dev) if want_mask else None
        scales = (ctypes.c_float * 4)(float(coord_scale), float(noobj_scale), float(obj_scale), float(class_scale))
        common = [ctypes.byref(cfg), _abi.pointer_array([h.data_ptr() for h in heads]),
                  gt.data_ptr(), B, pad, scales, float(ignore_threshold),
                  1 if normalize_by_num_gt else 0, ws.data_ptr(), ws.numel(),
                  per.data_ptr(), parts.data_ptr() if want_parts else None,
                  iou.data_ptr() if want_iou_max else None, total.data_ptr()]
        grads = None
        if grad_per_image is not None:
            gw = grad_per_image.to(torch.float32).contiguous()
            if gw.shape != (B,) or gw.device != dev:
                raise ValueError('grad_per_image must be [B] on the heads\' device')
            grads = [torch.empty_like(h) for h in heads]
            with torch.cuda.device(dev):
                _abi.check(lib.yh_loss_with_grad(*common, gw.data_ptr(),
                                                 _abi.pointer_array([g.data_ptr() for g in grads]), self._stream(dev)))
        else:
            with torch.cuda.device(dev):
                _abi.check(lib.yh_loss_forward(*common, mask.data_ptr() if want_mask else None, self._stream(dev)))
        if out is not None:
            out['grads'] = grads
            if out['reduce_pair'].data_ptr() != ws.data_ptr():
                out['reduce_pair'] = ws[:16].view(torch.float64)
            return out
        # reduce_pair: the workspace's leading fp64 pair (sum of the per-image losses, image count), valid
        # until the next loss call on this head and shape: what parallel.LossReducer all-reduces
        return dict(per_image=per, sum=total[:1], mean=total[1:], parts=parts, iou_max=iou, noobj_mask=mask,
                    _total=total, grads=grads, reduce_pair=ws[:16].view(torch.float64))

    def loss_backward(self, heads, ground_truth, coord_scale, noobj_scale, obj_scale, class_scale,
                      normalize_by_num_gt, noobj_mask, grad_per_image):
        """d(sum_b grad_per_image[b] * loss_per_image[b]) / d(head) for every head
        tensor: the backward of loss() (what optimizer.minimize differentiates at
        YOLOv3.py:312-318), from the forward call's inputs and noobj_mask."""
        lib = _abi.load()
        heads = self._check_heads(heads)
        cfg = self.config_for(heads)
        dev = heads[0].device
        gt = ground_truth.to(torch.float32).contiguous()
        B, pad = heads[0].shape[0], int(gt.shape[1])
        gw = grad_per_image.to(torch.float32).contiguous()
        if gw.shape != (B,) or noobj_mask.shape != (B, self.boxes_per_image(heads)) or noobj_mask.dtype != torch.uint8:
            raise ValueError('grad_per_image must be [B] and noobj_mask the uint8 [B, N] of the forward call')
        grads = [torch.empty_like(h) for h in heads]
        scales = (ctypes.c_float * 4)(float(coord_scale), float(noobj_scale), float(obj_scale), float(class_scale))
        with torch.cuda.device(dev):
            _abi.check(lib.yh_loss_backward(ctypes.byref(cfg), _abi.pointer_array([h.data_ptr() for h in heads]),
                                            gt.data_ptr(), B, pad, scales, 1 if normalize_by_num_gt else 0,
                                            noobj_mask.contiguous().data_ptr(), gw.data_ptr(),
                                            _abi.pointer_array([g.data_ptr() for g in grads]), self._stream(dev)))
        return grads

    def loss_and_grad(self, heads, ground_truth, coord_scale, noobj_scale, obj_scale, class_scale,
                      ignore_threshold, normalize_by_num_gt, grad_scale=1.0):
        """Forward and backward in one library call (yh_loss_with_grad), without an
        autograd graph: (loss() result, [d(grad_scale * sum of the per-image
        losses) / d(head)]).  For training loops that hand the gradients on
        themselves (torch.autograd.backward(heads, grads)): the upstream gradient
        of reduce_mean is a constant known before the forward pass, so the
        gradient tensor is filled while the ignore mask is computed."""
        B, dev = heads[0].shape[0], heads[0].device
        key = (B, float(grad_scale), dev)
        gw = self._grad_weights.get(key)
        if gw is None:
            gw = self._grad_weights[key] = torch.full((B,), float(grad_scale), dtype=torch.float32, device=dev)
        out = self.loss(heads, ground_truth, coord_scale, noobj_scale, obj_scale, class_scale,
                        ignore_threshold, normalize_by_num_gt, grad_per_image=gw)
        return out, out['grads']

    def loss_with_grad(self, heads, ground_truth, coord_scale, noobj_scale, obj_scale, class_scale,
                       ignore_threshold, normalize_by_num_gt):
        """(sum of the per-image losses [1], per-image losses [B]) as tensors that
        carry an autograd graph back to `heads`."""
        out = _HeadLoss.apply(self, ground_truth, (float(coord_scale), float(noobj_scale), float(obj_scale),
                                                   float(class_scale)), float(ignore_threshold),
                              bool(normalize_by_num_gt), *heads)
        return out[0], out[1]


class _HeadLoss(torch.autograd.Function):
    """DetectionHead.loss / loss_backward as one differentiable op."""

    @staticmethod
    def forward(ctx, head_obj, ground_truth, scales, ignore_threshold, normalize, *heads):
        heads = [h.detach() for h in heads]
        out = head_obj.loss(heads, ground_truth, *scales, ignore_threshold, normalize, want_mask=True)
        ctx.head_obj, ctx.scales, ctx.normalize = head_obj, scales, normalize
        ctx.save_for_backward(ground_truth, out['noobj_mask'], *heads)
        return out['sum'], out['per_image']

    @staticmethod
    def backward(ctx, grad_sum, grad_per):
        ground_truth, mask = ctx.saved_tensors[0], ctx.saved_tensors[1]
        heads = list(ctx.saved_tensors[2:])
        # upstream weight of every per-image loss: through `per_image` and through `sum`
        gw = grad_per + grad_sum.reshape(())
        grads = ctx.head_obj.loss_backward(heads, ground_truth, *ctx.scales, ctx.normalize, mask, gw)
        return (None, None, None, None, None) + tuple(grads)



class DetectPipeline:
    """DetectionHead.detect software-pipelined over a sequence of equally shaped
    batches on one stream (yh_detect_pipelined): submit(heads) streams this
    batch and, inside the same kernel launch, selects the previous one, whose
    short latency-bound NMS stage hides behind the bandwidth-bound streaming
    pass.  submit returns the PREVIOUS batch's Detections (None on the first
    call); flush() returns the last batch's.  Two workspaces and two output
    sets alternate: a returned Detections object stays valid until the
    second submit after it.  Results per batch are those of detect(); a batch
    whose candidate capacity overflowed is flagged (Detections.overflowed) and
    has to be submitted again to a pipeline with a larger max_candidates.
    Thresholds, max_boxes and the stream are fixed at construction: the call of
    a slot is prepared once and reused while batches keep their shape."""

    def __init__(self, head, score_threshold, iou_threshold, max_boxes, max_candidates=None,
                 want_class_candidates=False, stream=None, tag=0):
        self.head = head
        self.tag = tag             # keeps the workspaces of several pipelines on one head apart
        self.thr, self.iou, self.max_boxes = float(score_threshold), float(iou_threshold), int(max_boxes)
        self.cap = max_candidates
        self.want_cc = want_class_candidates
        self.stream = stream
        self._i = 0
        self._prev = None          # (cfg, B, workspace, outputs, device) of the batch awaiting its select stage
        self._outputs = [None, None]
        self._steady = [None, None]    # per slot: the prepared call of a batch shaped like the last one
        # the head tensors of the batch awaiting its select stage: with a dense candidate capacity that stage
        # decodes the boxes it looks at from them, so they stay referenced (by the Detections object returned
        # for that batch) until the results are read
        self._prev_heads = None

    def _args(self, cfg, head_ptrs, B, cap, ws, prev_ws, nbytes, outputs, dev, events):
        ev = events or (None, None)
        st = self.head._stream(dev) if self.stream is None else ctypes.c_void_p(self.stream.cuda_stream)
        outs = self.head._output_pointers(outputs) if outputs is not None else [None] * 7
        return [ctypes.byref(cfg), head_ptrs, B, self.thr, self.iou, self.max_boxes, cap,
                ws, prev_ws, nbytes] + outs + [
                st, ctypes.c_void_p(ev[0].cuda_event) if ev[0] is not None else None,
                ctypes.c_void_p(ev[1].cuda_event) if ev[1] is not None else None]

    def _submit_steady(self, heads):
        """The call of submit() for a batch shaped like the last one on this slot, with
        everything but the head pointers prepared by that earlier call; None when the
        batch has to take the checked path."""
        slot = self._i & 1
        rec = self._steady[slot]
        prev = self._prev
        if rec is None or prev is None:
            return None
        shapes, dev, cap, prev_ws, prev_outputs, args, arr, mine, default_for, epoch = rec
        if epoch != _abi.tuning_epoch:
            return None
        # the prepared arguments hold the addresses of exactly these tensors
        if prev[2] is not prev_ws or prev[3] is not prev_outputs or self._outputs[slot] is not mine[3]:
            return None
        if len(heads) != len(shapes):
            return None
        if self.stream is None:
            args[-3] = self.head._stream(dev)
        for k, h in enumerate(heads):
            if not (isinstance(h, torch.Tensor) and h.shape == shapes[k] and h.dtype is torch.float32
                    and h.device == dev and h.is_contiguous()):
                return None
            arr[k] = h.data_ptr()
        if self.cap is None and self.head.default_max_candidates(default_for) != cap:
            return None
        lib = _abi.load()
        if torch.cuda.current_device() == dev.index:
            _abi.check(lib.yh_detect_pipelined(*args))
        else:
            with torch.cuda.device(dev):
                _abi.check(lib.yh_detect_pipelined(*args))
        done = Detections(*prev[3], None, stream=self.stream)
        done.keepalive = self._prev_heads
        self._prev_heads = list(heads)
        self._prev = mine
        self._i += 1
        return done

    def submit(self, heads, stream_events=None):
        if stream_events is None:
            done = self._submit_steady(heads)
            if done is not None:
                return done
        lib = _abi.load()
        hd = self.head
        given = heads
        heads = hd._check_heads(heads)
        if self.stream is not None:
            for h, g in zip(heads, given):
                if h is not g:                       # a packed copy made here: it is read on the pipeline's stream
                    self.stream.wait_stream(torch.cuda.current_stream(h.device))
                    h.record_stream(self.stream)
        dev = heads[0].device
        N = hd.boxes_per_image(heads)
        cap = int(self.cap) if self.cap is not None else hd.default_max_candidates(N)
        slot = self._i & 1
        with _allocating_on(self.stream):
            cfg, B, N, ws = hd._detect_plan(heads, cap, self.max_boxes, slot=('pipe', self.tag, slot))
        prev = self._prev
        if prev is not None and (prev[1] != B or prev[5] != cap or bytes(prev[0]) != bytes(cfg)):
            raise ValueError('a pipeline takes equally shaped batches; flush() before changing shape')
        if self._outputs[slot] is None or self._outputs[slot][0].shape[0] != B:
            with _allocating_on(self.stream):
                self._outputs[slot] = hd._alloc_outputs(B, self.max_boxes, dev, self.want_cc)
        arr = _abi.pointer_array([h.data_ptr() for h in heads])
        args = self._args(cfg, arr, B, cap, ws.data_ptr(),
                          prev[2].data_ptr() if prev is not None else None, ws.numel(),
                          prev[3] if prev is not None else None, dev, stream_events)
        with torch.cuda.device(dev):
            _abi.check(lib.yh_detect_pipelined(*args))
        done = Detections(*prev[3], None, stream=self.stream) if prev is not None else None
        if done is not None:
            done.keepalive = self._prev_heads
        self._prev_heads = list(heads)
        self._prev = (cfg, B, ws, self._outputs[slot], dev, cap)
        if prev is not None and stream_events is None:
            self._steady[slot] = (tuple(h.shape for h in heads), dev, cap, prev[2], prev[3], args, arr,
                                  self._prev, N, _abi.tuning_epoch)
        self._i += 1
        return done

    def flush(self):
        prev = self._prev
        if prev is None:
            return None
        lib = _abi.load()
        cfg, B, ws, outputs, dev, cap = prev
        args = self._args(cfg, None, B, cap, None, ws.data_ptr(), ws.numel(), outputs, dev, None)
        with torch.cuda.device(dev):
            _abi.check(lib.yh_detect_pipelined(*args))
        self._prev = None
        done = Detections(*outputs, None, stream=self.stream)
        done.keepalive, self._prev_heads = self._prev_heads, None
        return done


class DetectLanes:
    """Several DetectPipelines on streams of their own, fed in turn: the tail of a
    launch (the few surviving rows: latency-bound, little data) overlaps the start
    of the next lane's launch (the objectness column: bandwidth-bound), which one
    stream cannot do because its launches run one after the other.
    submit(heads) returns the Detections of the batch submitted `lanes` * 1 calls
    earlier on the same lane (None at first); flush() returns the outstanding ones
    in submission order.  The head tensors must be complete as far as the lane's
    stream can tell: produce them on `stream_of_next()`, or pass wait_current=True
    to make the lane wait for the caller's current stream (an event record and a
    stream wait per submit).  Leave a batch's head tensors unchanged until its
    Detections have been read: the lane's launches read them asynchronously (the
    select stage of a dense candidate set decodes boxes from them); the Detections
    object keeps them referenced."""

    def __init__(self, head, score_threshold, iou_threshold, max_boxes, lanes=2, max_candidates=None,
                 want_class_candidates=False, device=None):
        if lanes < 1:
            raise ValueError('at least one lane')
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        self.streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
        self.pipes = [DetectPipeline(head, score_threshold, iou_threshold, max_boxes, max_candidates,
                                     want_class_candidates, stream=s, tag=('lane', i))
                      for i, s in enumerate(self.streams)]
        self._i = 0

    def stream_of_next(self):
        return self.streams[self._i % len(self.streams)]

    def submit(self, heads, wait_current=False):
        lane = self._i % len(self.pipes)
        if wait_current:
            self.streams[lane].wait_stream(torch.cuda.current_stream(heads[0].device))
        self._i += 1
        return self.pipes[lane].submit(heads)

    def flush(self):
        n = len(self.pipes)
        order = [(self._i + k) % n for k in range(n)]          # oldest outstanding batch first
        return [d for d in (self.pipes[lane].flush() for lane in order) if d is not None]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
