#!/usr/bin/env python3
"""Benchmark of the detection-head hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one pass of the path over one batch of synthetic head tensors that
are resident in HBM before the timed region starts.  For N > 1 the driver
launches one rank per GPU (torch.distributed.run); images are sharded by rank
(weak scaling by default: every rank owns a full per-GPU batch; with
--global-batch G strong scaling: rank r owns the images parallel.shard_range(G,
r, N) of a fixed global batch), the detect workloads need no collective and the
loss workloads all-reduce one scalar pair over RCCL, one collective per step
(--steps-per-collective n gathers n steps per collective; reported in config).

Workloads (BASELINE.json configs):
  detect416  config 2  YOLOv3-416 COCO-80, batch 64 / GPU, decode + score filter + per-class NMS   [default]
  loss416    config 3  YOLOv3-416 loss, 100 GT / image, batch 32 / GPU
  train416   config 3 with the backward to the head tensors (reads the head twice, writes its gradient)
  detect_v2  config 1  YOLOv2-416 VOC-20, batch 1 (latency-bound: one image)
  detect608  config 4  YOLOv3-608 COCO-80, batch 64 / GPU
  stress     config 5  100x100 grid x 10 priors, ~1e5 candidates / image at thr 0.001, batch 128 / GPU
  nms_seam   config 5 entered at the post-decode seam: boxes [B,N,4] and conf [B,N,C] resident in HBM,
             score filter + per-class NMS only (yh_nms)

Rank 0 prints ONE JSON line.

`value` / `ms_per_step` come from the first timed region of exactly K steps: a barrier and a device
synchronisation before it; after it every rank synchronises its device and stops its clock, then the ranks
meet at a barrier again; the maximum over ranks counts.  `timing` adds the percentiles over that region
and several more of the same length.

`roofline` prices the dominant kernel, the pass over the head tensor:
  kernel_ms        its average launch duration, HIP events on the launch stream, launches back to
                   back on ONE stream (what a kernel trace shows);
  traffic          HBM bytes per launch from the committed rocprofv3 PMC pass of this (workload,
                   kernel) (profiles/traffic.json: FETCH_SIZE doubled per the gfx950 correction,
                   plus WRITE_SIZE), null when no pass is committed;
  achieved, frac   traffic / kernel_ms, and that over the 8 TB/s peak: a statement about the memory
                   system, at most 1 (with no PMC pass the bytes come from `traffic_model`);
  algorithmic_*    the figure that prices N*(C+5)*4 bytes per image whatever the kernel reads.  The
                   default detect kernel is objectness-first: it reads one cache line per row and only
                   the rows whose objectness can still reach the score threshold, so this figure
                   exceeds the peak at a detector's survival rate and depends on the data:
                   `row_survival` says how many rows survived, `survival_sweep` shows the kernel from
                   2 % to 100 % survival.
`cpu_baseline` times the C restatement of the reference's arithmetic (oracle/, "port") on this
host's cores for bounded samples of the same workload (rank 0, N = 1 only): all cores (OpenMP over
images), one thread, and the NumPy restatement.
"""
import argparse
import ctypes
import gc
import json
import re
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import yolo_tf_amd as y                                    # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
INFINITY_CACHE_BYTES = 256 << 20
PRIME_STEPS = 64                 # untimed steps before the W warm-up steps: pipelines and host caches reach steady state
EXTRA_REGIONS = 6                # timed regions after the first, for the percentiles

WORKLOADS = {
    'detect416': dict(kind='detect', size=416, C=80, batch=64, thr=0.5, iou=0.5, max_boxes=10, dist={},
                      name='YOLOv3 416x416 COCO-80 decode+NMS, synthetic head tensors, 10647 boxes/img'),
    'detect_v2': dict(kind='detect', size=416, C=20, batch=1, thr=0.5, iou=0.5, max_boxes=10, dist={}, version=2,
                      name='YOLOv2 416x416 VOC-20 batch 1 decode+NMS (the reference\'s own CPU-runnable case), 845 boxes/img'),
    'detect608': dict(kind='detect', size=608, C=80, batch=64, thr=0.5, iou=0.5, max_boxes=10, dist={},
                      name='YOLOv3 608x608 COCO-80 decode+NMS, synthetic head tensors, 22743 boxes/img'),
    'stress': dict(kind='detect', size=None, C=80, batch=128, thr=0.001, iou=0.5, max_boxes=100, dist=y.synth.STRESS,
                   name='NMS stress: 100x100 grid x 10 priors, ~1e5 candidates/img at score_thresh 0.001, 80 classes'),
    'nms_seam': dict(kind='nms', size=None, C=80, batch=128, thr=0.001, iou=0.5, max_boxes=100, dist=y.synth.STRESS,
                     name='NMS stress at the post-decode seam: decoded boxes [B,N,4] + conf [B,N,C] resident, score filter '
                          '+ per-class NMS (100x100 grid x 10 priors, ~1e5 candidates/img at score_thresh 0.001, 80 classes)'),
    'loss416': dict(kind='loss', size=416, C=80, batch=32, G=100, pad=101,
                    name='YOLOv3 416x416 training loss path: 100 GT boxes/img IoU ignore-mask + target assignment'),
    'train416': dict(kind='loss', size=416, C=80, batch=32, G=100, pad=101, backward=True,
                     name='YOLOv3 416x416 training loss path, forward + backward to the head tensors'),
}


def build_head(w):
    if w['size'] is None:
        rng = np.random.Generator(np.random.PCG64(99))
        priors = [rng.uniform(0.5, 4.0, size=(10, 2)).astype(np.float32)]
        grids = [(100, 100)]
        return grids, priors, (32.0,), (32.0,)
    if w.get('version') == 2:
        g = w['size'] // 32
        return [(g, g)], [np.asarray(y.synth.V2_PRIORS, np.float32)], (32.0,), (32.0,)
    s = [8., 16., 32.]
    priors = [(np.asarray(y.synth.V3_PRIORS[i], np.float32) / np.float32(s[i])) for i in range(3)]
    return y.synth.v3_grids(w['size']), priors, (32.0, 32.0, 16.0), (32.0, 16.0, 8.0)


# ----------------------------------------------------------------------------- CPU baselines
def _time_for(fn, budget_s, max_reps=10000):
    fn()                                                    # warm-up (page-in, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        fn(); reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= max_reps:
            return reps, el


def cpu_baseline(w, grids, priors, box_mult, gt_rate):
    """The CPU restatements of the reference's op sequence on this host, bounded samples."""
    from oracle import c_oracle as co, yolo_head_oracle as O
    threads = max(1, min(16, int(os.environ.get('OMP_NUM_THREADS', '0')) or (os.cpu_count() or 1)))
    cfg = co.make_config(grids, priors, box_mult, gt_rate, w['C'])
    P = len(priors[0])
    big = w['size'] is None                                  # 1e5 boxes per image: fewer images per pass

    def sample(n_img, nthreads, budget):
        heads = y.synth.make_heads(4242, n_img, grids, P, w['C'], **w.get('dist', {}))
        if w['kind'] == 'nms':                               # decoded once, outside the timed passes
            dboxes, dconf = co.decode(cfg, heads, True, threads)
            fn = lambda: co.nms(dboxes, dconf, w['thr'], w['iou'], w['max_boxes'], threads=nthreads)
        elif w['kind'] == 'detect':
            fn = lambda: co.detect(cfg, heads, w['thr'], w['iou'], w['max_boxes'], threads=nthreads, unpack=False)
        else:
            gt = y.synth.make_ground_truth(4243, n_img, w['G'], w['pad'], w['size'], w['C'])
            fn = lambda: co.loss(cfg, heads, gt, (1., 1., 5., 1.), 0.5, True, threads=nthreads)
        reps, el = _time_for(fn, budget)
        return n_img * reps / el, '%d synthetic images x %d passes (%.1f s)' % (n_img, reps, el), heads

    rate, what, _ = sample(threads if big else 2 * threads, threads, 10.0)
    res = dict(value=round(rate, 2), unit='images/sec', cores=threads, kind='port',
               sample=what + ' of the same workload, C restatement of the reference op sequence '
                             '(oracle/yolo_head_oracle.c), OpenMP over images')
    rate1, what1, heads1 = sample(1 if big else 4, 1, 5.0)
    res['single_thread'] = dict(value=round(rate1, 2), unit='images/sec', cores=1, kind='port', sample=what1 + ', the same C code on one thread')
    # NumPy restatement (oracle/yolo_head_oracle.py), one image per pass
    h0 = [h[0] for h in heads1]
    if w['kind'] == 'nms':
        nb, nc = O.decode(h0, priors, box_mult, w['C'])
        fn = lambda: O.nms_per_class(nb, nc, w['thr'], w['max_boxes'], w['iou'])
    elif w['kind'] == 'detect':
        fn = lambda: O.detect(h0, priors, box_mult, w['C'], w['thr'], w['max_boxes'], w['iou'])
    elif w.get('version') == 2:
        gt = y.synth.make_ground_truth(4243, 1, w['G'], w['pad'], w['size'], w['C'])
        fn = lambda: O.loss_v2_image(h0[0], gt[0], priors[0], w['C'])
    else:
        gt = y.synth.make_ground_truth(4243, 1, w['G'], w['pad'], w['size'], w['C'])
        fn = lambda: O.loss_v3_image(h0, gt[0], priors, w['C'])
    try:
        reps, el = _time_for(fn, 4.0)
        res['numpy'] = dict(value=round(reps / el, 2), unit='images/sec', cores=1, kind='port',
                            sample='1 synthetic image x %d passes (%.1f s), NumPy restatement op for op '
                                   '(oracle/yolo_head_oracle.py)' % (reps, el))
    except Exception as e:                                   # the baseline must not take the benchmark down
        res['numpy'] = dict(value=None, error=repr(e))
    return res


# ----------------------------------------------------------------------------- helpers
def measured_read_bandwidth(sets, dev, reps=20):
    """GB/s a read-only kernel reaches on these very buffers (yh_probe_read, HIP events): the
    achievable figure next to the nominal peak."""
    lib = y._abi.load()
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    bufs = [h for hs in sets for h in hs if h.numel() * 4 >= (32 << 20)] or [max((h for hs in sets for h in hs), key=lambda t: t.numel())]
    for h in bufs:
        y._abi.check(lib.yh_probe_read(h.data_ptr(), h.numel() * 4, st))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    nbytes = 0
    for i in range(reps):
        h = bufs[i % len(bufs)]
        y._abi.check(lib.yh_probe_read(h.data_ptr(), h.numel() * 4, st))
        nbytes += h.numel() * 4
    b.record()
    torch.cuda.synchronize(dev)
    return nbytes / (a.elapsed_time(b) * 1e-3) / 1e9


def detect_kernel(hd_cap, detect_kernel_knob):
    """(short name, description) of the dominant kernel of a detect workload, as
    csrc/yh_abi.hip::plan_stream picks it from the candidate capacity and the tuning."""
    if hd_cap > 2048:                                        # capacity beyond the single-launch select stage
        return 'k_stream', 'k_stream<detect> (workgroup tiles, reads the whole head: decode + score filter + candidate compaction)'
    if detect_kernel_knob in (0, 1):
        return 'k_detect_rows', ('k_detect_rows (objectness column, then the surviving rows: decode + score filter + '
                                 'candidate compaction)')
    return 'k_stream_wave', 'k_stream_wave (wave-private tiles, reads the whole head: decode + score filter + candidate compaction)'


def load_valu_busy(workload, kernel):
    """Vector-ALU occupancy of a kernel from the committed SQ counter pass (tools/pmc_sq.sh): the share of
    the SIMDs' cycles spent issuing VALU instructions.  SQ_ACTIVE_INST_VALU counts four-cycle issue slots
    summed over all SIMDs; SQ_BUSY_CYCLES is summed over the 32 shader engines of the 8 XCDs."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_%s_sq.txt' % workload)
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    vals = {}
    for line in lines:
        if kernel in line.split(' ', 1)[0]:
            for m in re.finditer(r'(\w+)=([-+.\deE]+)\(n=\d+\)', line):
                vals[m.group(1)] = float(m.group(2))
    if 'SQ_ACTIVE_INST_VALU' not in vals or 'SQ_BUSY_CYCLES' not in vals:
        return None
    simds, engines = 1024, 32
    busy = 4.0 * vals['SQ_ACTIVE_INST_VALU'] / (simds * vals['SQ_BUSY_CYCLES'] / engines)
    return dict(valu_instructions_per_launch=int(vals.get('SQ_INSTS_VALU', 0)), valu_busy_frac=round(busy, 3),
                source='profiles/r03_%s_sq.txt (rocprofv3 --pmc SQ_* passes of tools/pmc_sq.sh)' % workload,
                note='share of SIMD cycles issuing vector-ALU instructions while the kernel runs' +
                     ('; the loss kernel is bound by vector-ALU issue and its prologue latency, not by HBM'
                      if kernel == 'k_loss_main' else ''))


def load_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` in `workload` from the committed PMC passes, or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            rec = json.load(f).get(workload, {}).get(kernel)
    except (OSError, ValueError, AttributeError):
        return None
    return rec if isinstance(rec, dict) else None


def row_survival(sets, C, thr):
    """Fraction of box rows whose objectness can still reach the score threshold
    (sigmoid(obj) >= thr), over all input sets."""
    if thr <= 0.0:
        return 1.0
    if thr >= 1.0:
        return 0.0
    logit = math.log(thr / (1.0 - thr))
    n = k = 0
    for hs in sets:
        for h in hs:
            obj = h.reshape(-1, C + 5)[:, C + 4]
            k += int((obj >= logit).sum().item())
            n += obj.numel()
    return k / max(1, n)


def rows_traffic_model(rows, survival, K):
    """Bytes the objectness-first kernel must move per launch: one 128-byte line per row (the
    objectness column) plus the surviving rows."""
    return rows * 128.0 + rows * survival * K * 4.0


def percentiles(xs):
    a = np.asarray(xs, np.float64)
    return dict(p10=round(float(np.percentile(a, 10)), 5), p50=round(float(np.percentile(a, 50)), 5),
                p90=round(float(np.percentile(a, 90)), 5), n=int(a.size))


def rotation_for(touched_bytes_per_set, floor=4, cap=24):
    """Distinct input sets so that a line is re-read only after 3x the Infinity Cache of other
    lines has been touched."""
    return int(max(floor, min(cap, math.ceil(3.0 * INFINITY_CACHE_BYTES / max(1.0, touched_bytes_per_set)))))


def solo_kernel_ms(hd, w, sets, dev, per_run, runs=5):
    """Average duration of one launch of the dominant kernel with launches back to back on one
    stream (each launch also carries the previous batch's select stage when that rides along):
    HIP event pairs around `runs` runs of `per_run` launches; all runs returned."""
    solo = y.DetectPipeline(hd, w['thr'], w['iou'], w['max_boxes'], tag='solo')
    n = len(sets)
    for i in range(max(10, n)):
        solo.submit(sets[i % n])
    out = []
    for _ in range(runs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(per_run):
            solo.submit(sets[i % n])
        b.record()
        torch.cuda.synchronize(dev)
        out.append(a.elapsed_time(b) / per_run)
    last = solo.flush()
    return out, last


def survival_sweep(w, grids, priors, box_mult, gt_rate, dev, B, rows_per_launch, K):
    """The objectness-first kernel against the share of rows that survive its prefilter:
    objectness ~ N(mu, 2) with mu chosen for about 2 / 16 / 50 / 100 % at threshold 0.5, class
    logits too low for any candidate (so the candidate capacity plays no part).  Microseconds
    per launch (one stream, back to back) and the fraction of the HBM peak the bytes that must
    move (rows_traffic_model) reach."""
    out = []
    for mu in (-4.0, -2.0, 0.0, 9.0):
        surv_nominal = 0.5 * (1.0 + math.erf(mu / 2.0 / math.sqrt(2.0)))
        touched = rows_traffic_model(rows_per_launch, surv_nominal, K)
        n_sets = rotation_for(touched, floor=3, cap=10)
        sets = [y.synth.torch_heads(dev, 500 + i, B, grids, len(priors[0]), w['C'], obj_mean=mu, hot_mean=-6.0)
                for i in range(n_sets)]
        surv = row_survival(sets[:1], w['C'], w['thr'])
        hd = y.DetectionHead(w['C'], priors, box_mult, gt_rate)
        runs, last = solo_kernel_ms(hd, w, sets, dev, per_run=60, runs=3)
        ms = float(np.median(runs))
        model = rows_traffic_model(rows_per_launch, surv, K)
        out.append(dict(obj_mean=mu, row_survival=round(surv, 4), kernel_us=round(ms * 1e3, 2),
                        traffic_model_bytes=int(model), model_gbs=round(model / (ms * 1e-3) / 1e9, 1),
                        frac=round(model / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), input_sets=n_sets,
                        overflow=bool(last.overflowed())))
        del sets, hd
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='detect416', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='images per GPU (default: the workload\'s)')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling: a fixed global batch of this many images, rank r takes the contiguous range '
                         'parallel.shard_range(G, r, world) (BASELINE config 4: 512 -> 256/128/64 per GPU at 2/4/8 GPUs; '
                         'config 3: 256 -> 32 per GPU at 8); value = G * steps / time of the slowest rank')
    ap.add_argument('--steps-per-collective', type=int, default=1,
                    help='loss workloads: steps whose scalar pairs share one all-reduce (default 1: a collective per '
                         'step, what a loop that reads every step\'s loss pays)')
    ap.add_argument('--in-flight', type=int, default=2, choices=(1, 2),
                    help='detect workloads: 2 = software pipeline (a step streams batch i and selects batch '
                         'i-1 in one launch), 1 = every batch runs to completion before the next')
    ap.add_argument('--streams', type=int, default=None,
                    help='detect workloads, --in-flight 2: pipelines on streams of their own, fed in turn (the '
                         'latency-bound tail of one launch overlaps the bandwidth-bound start of the next); '
                         '1 = one stream, launches back to back; default 3')
    ap.add_argument('--detect-kernel', type=int, default=0, choices=(0, 1, 2),
                    help='yh_set_tuning.detect_kernel: 0/1 = objectness-first rows kernel, 2 = streaming kernel')
    ap.add_argument('--input-sets', type=int, default=0,
                    help='distinct input sets the steps rotate through (default: sized by the bytes a step touches, so '
                         'that a line is re-read only after 3x the Infinity Cache of other lines)')
    ap.add_argument('--row-steps', type=int, default=0,
                    help='yh_set_tuning.row_steps: steps of 64 rows per work unit of the rows kernel (0 = planned from the batch)')
    ap.add_argument('--stream-resident', type=int, default=0,
                    help='yh_set_tuning.stream_resident: workgroups of the tile kernel launched per compute unit (0 = one per cut)')
    ap.add_argument('--force-allreduce', action='store_true',
                    help='loss workloads with one rank: open a one-rank process group (RCCL) and all-reduce the loss '
                         'pair anyway, so that the collective runs on a single GPU')
    ap.add_argument('--loss-prepared', action='store_true',
                    help='loss workloads: every step issues yh_loss_prepare for its ground truth ahead of the loss call '
                         '(same stream, inside the timed region: nothing runs here that it could hide behind)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true', help='skip the survival sweep of the detect416 line')
    ap.add_argument('--regions', type=int, default=EXTRA_REGIONS, help='timed regions after the first (percentiles)')
    args = ap.parse_args()
    if os.environ.get('YH_BENCH_WATCHDOG'):                   # diagnostics: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['YH_BENCH_WATCHDOG']), exit=True)

    # YH_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    rank, world, local_rank = y.parallel.init_from_env(os.environ.get('YH_DIST_BACKEND'), even_alone=args.force_allreduce)
    if world != max(1, args.gpus):
        raise SystemExit('--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the path has no CPU fallback')
    dev = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if args.detect_kernel or args.stream_resident or args.row_steps:
        y._abi.set_tuning(detect_kernel=args.detect_kernel, stream_resident=args.stream_resident, row_steps=args.row_steps)

    w = dict(WORKLOADS[args.workload])
    B = args.batch or w['batch']
    if args.global_batch:
        if args.batch:
            raise SystemExit('--batch and --global-batch exclude each other')
        lo, hi = y.parallel.shard_range(args.global_batch, rank, world)
        B = hi - lo
        if B < 1:
            raise SystemExit('--global-batch %d leaves rank %d of %d without an image' % (args.global_batch, rank, world))
    grids, priors, box_mult, gt_rate = build_head(w)
    P, C = len(priors[0]), w['C']
    K = C + 5
    hd = y.DetectionHead(C, priors, box_mult, gt_rate)
    N = sum(h * ww * P for h, ww in grids)
    bytes_img = N * K * 4
    rows_per_launch = B * N
    steps, warmup = args.steps, args.warmup

    # ---- inputs: enough distinct sets that a step never re-reads lines the Infinity Cache still holds,
    # sized by the bytes a step TOUCHES (the objectness-first kernel and the loss touch a fraction of a set)
    decoded = None
    if w['kind'] == 'nms':
        # the post-decode seam: every input set is a decoded batch (boxes [B,N,4], conf [B,N,C]); the head
        # tensors it came from are dropped again
        def decoded_set(seed):
            hs = y.synth.torch_heads(dev, seed, B, grids, P, C, **w.get('dist', {}))
            out = hd.decode(hs)
            del hs
            return out
        probe, surv0 = None, None
        kname, kdesc = 'k_stream', 'k_stream<conf> (workgroup tiles over the confidence matrix: score filter + candidate compaction)'
        touched = float(rows_per_launch) * (C * 4 + 16)
        rotate = rotation_for(touched, cap=4)
        decoded = [decoded_set(1000 * rank + i) for i in range(rotate)]
        torch.cuda.empty_cache()
        hd.nms(decoded[0][0], decoded[0][1], w['thr'], w['iou'], w['max_boxes']).resolved()   # settles the capacity (untimed)
        sets = [[c] for _, c in decoded]                     # what the read-bandwidth yardstick streams
    elif w['kind'] == 'detect':
        probe = y.synth.torch_heads(dev, 1000 * rank, B, grids, P, C, **w.get('dist', {}))
        surv0 = row_survival([probe], C, w['thr'])
        hd.detect(probe, w['thr'], w['iou'], w['max_boxes']).resolved()     # settles the candidate capacity (untimed)
        kname, kdesc = detect_kernel(hd.default_max_candidates(N), args.detect_kernel)
        touched = rows_traffic_model(rows_per_launch, surv0, K) if kname == 'k_detect_rows' else float(B * bytes_img)
    else:
        probe, surv0 = None, None
        kname, kdesc = 'k_loss_main', 'k_loss_main (ignore-mask IoU loop + assignment + loss sums; reads the objectness column and the assigned rows)'
        touched = rows_per_launch * 128.0 * (3.0 if w.get('backward') else 1.0)
    if decoded is None:
        rotate = args.input_sets or rotation_for(touched, cap=24 if B * bytes_img < (1 << 30) else 4)
        sets = [probe if (i == 0 and probe is not None) else
                y.synth.torch_heads(dev, 1000 * rank + i, B, grids, P, C, **w.get('dist', {})) for i in range(rotate)]
    else:
        bytes_img = N * (C * 4 + 16)                         # what this seam reads per image: conf row + box
    rotation_note = ('%d distinct input sets of %.0f MB, %.0f MB of lines touched per step: %.1fx the 256 MiB Infinity '
                     'Cache between two reads of a line' % (rotate, B * bytes_img / 1e6, touched / 1e6,
                                                            rotate * touched / INFINITY_CACHE_BYTES))

    extra = {}
    pipe = None
    nstreams = 1
    if w['kind'] == 'nms':
        # yh_nms calls of consecutive batches go to streams of their own in turn (--streams, default 3), each with
        # its own head object (workspace) and outputs: the select stage's launches of one batch run beside the
        # pass over the next batch's confidence matrix, as in the stress workload
        outputs = hd._alloc_outputs(B, w['max_boxes'], dev, False)
        nstreams = max(1, args.streams or 3)
        nms_lanes = []
        for _ in range(nstreams):
            lane_hd = y.DetectionHead(C, priors, box_mult, gt_rate)
            lane_st = torch.cuda.Stream(dev)
            with torch.cuda.stream(lane_st):
                lane_hd.nms(decoded[0][0], decoded[0][1], w['thr'], w['iou'], w['max_boxes']).resolved()
            nms_lanes.append((lane_st, lane_hd, hd._alloc_outputs(B, w['max_boxes'], dev, False)))
        torch.cuda.synchronize(dev)
        extra.update(streams=nstreams)

        def step(i):
            bx, cf = decoded[i % rotate]
            lane_st, lane_hd, lane_out = nms_lanes[i % nstreams]
            with torch.cuda.stream(lane_st):
                return lane_hd.nms(bx, cf, w['thr'], w['iou'], w['max_boxes'], outputs=lane_out)
    elif w['kind'] == 'detect':
        # Software pipeline over the sequence of batches, as a serving loop would run the path: a
        # step streams batch i and, inside the same launch, selects batch i-1 (yh_detect_pipelined).
        # --in-flight 1 runs every batch to completion before the next (yh_detect).
        lanes = 2 if args.in_flight > 1 else 1
        outputs = hd._alloc_outputs(B, w['max_boxes'], dev, False)
        nstreams = max(1, args.streams or w.get('streams', 3)) if lanes > 1 else 1
        if lanes > 1:
            pipe = (y.DetectLanes(hd, w['thr'], w['iou'], w['max_boxes'], lanes=nstreams, device=dev) if nstreams > 1
                    else y.DetectPipeline(hd, w['thr'], w['iou'], w['max_boxes']))

        def step(i):
            if pipe is not None:
                return pipe.submit(sets[i % rotate])          # the inputs are resident: nothing to wait for
            return hd.detect(sets[i % rotate], w['thr'], w['iou'], w['max_boxes'], outputs=outputs)
        extra.update(batches_in_flight=lanes * nstreams, streams=nstreams)
    else:
        gts = [torch.from_numpy(y.synth.make_ground_truth(77 + 10 * rank + i, B, w['G'], w['pad'], w['size'], C)).to(dev)
               for i in range(rotate)]
        # persistent buffers: no allocation per step; one collective per --steps-per-collective steps
        reducer = y.parallel.LossReducer(dev, always_reduce=args.force_allreduce, batch=max(1, args.steps_per_collective))
        extra['steps_per_collective'] = reducer._batch if (world > 1 or args.force_allreduce) else None
        if args.force_allreduce:
            extra['allreduce'] = 'forced: %d-rank %s process group' % (world, dist.get_backend())
        reuse = {}                                           # the loss outputs are written in place step after step
        head_shapes = [tuple(h.shape) for h in sets[0]]
        extra['ground_truth_tables'] = ('yh_loss_prepare ahead of every loss call, on the same stream' if args.loss_prepared
                                        else 'built by the loss call')

        def step(i):
            if args.loss_prepared:
                hd.loss_prepare(head_shapes, gts[i % rotate])
            if w.get('backward'):
                # forward + backward in one library call: d(.5 * mean loss) / d(head) for every head
                out, _grads = hd.loss_and_grad(sets[i % rotate], gts[i % rotate], 1., 1., 5., 1., 0.5, True,
                                               grad_scale=0.5 / (args.global_batch or y.parallel.global_batch(B)))
            else:
                out = reuse['out'] = hd.loss(sets[i % rotate], gts[i % rotate], 1., 1., 5., 1., 0.5, True, out=reuse.get('out'))
            return reducer.global_mean(out)                  # the data-parallel scalar all-reduce

    def drain():
        """Everything this rank has enqueued is complete when this returns."""
        if w['kind'] == 'loss':
            reducer.flush()                                   # steps still gathered (--steps-per-collective > 1) go out inside the region
        torch.cuda.synchronize(dev)

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    gc.collect()
    gc.disable()                                            # no collector pause inside the few-millisecond timed regions
    # Untimed priming, whatever W is: every lane's pipeline reaches its steady state (both workspaces
    # and output sets allocated, the prepared call of each slot built), and the host path is warm.
    # In bursts closed by a synchronisation, like the timed regions: the first submits after the host
    # thread has slept in a synchronisation take two to three times as long as the rest until that path
    # has run a few times (25-30 us against 8), which a region of a few dozen steps would feel.
    last = None
    for burst in range(4):
        for i in range(PRIME_STEPS // 4):
            step(i)
        barrier()
    for i in range(warmup):
        last = step(i)
    # A timed region: all ranks leave a barrier with idle devices, every rank runs its K steps and stops its
    # clock when its own device has finished them, the ranks meet again, and the slowest rank's time counts
    # (the maximum is taken below).  The closing barrier's own latency, a collective of tens of microseconds
    # under RCCL, is not part of any rank's K steps.
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        last = step(i)
    enqueue = time.perf_counter() - t0
    drain()
    elapsed = time.perf_counter() - t0
    barrier()
    region_ms = [1e3 * elapsed / steps]
    for _ in range(max(0, args.regions)):                   # more regions of the same length, for the spread
        barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            last = step(i)
        drain()
        region_ms.append(1e3 * (time.perf_counter() - t1) / steps)
        barrier()
    gc.enable()

    kern_runs = None
    if w['kind'] == 'nms':
        if last.overflowed():
            raise SystemExit('candidate capacity overflow in the benchmark configuration')
        extra['detections_per_image_last_step'] = round(float(last.count.float().mean().item()), 2)
        extra['max_candidates'] = hd.default_max_candidates(N)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in evs:                                      # a first record creates the hipEvent_t handles
            a.record(); b.record()
        for i, (a, b) in enumerate(evs):
            bx, cf = decoded[i % rotate]
            hd.nms(bx, cf, w['thr'], w['iou'], w['max_boxes'], outputs=outputs, stream_events=(a, b))
        torch.cuda.synchronize(dev)
        kern_runs = [a.elapsed_time(b) for a, b in evs[2:]]
        timing = 'HIP event pairs around the kernel in %d yh_nms_timed calls after the timed regions' % len(kern_runs)
    elif w['kind'] == 'detect':
        if pipe is not None:
            last = pipe.flush()                             # the last batch's select stage (outside the timed regions)
            if isinstance(last, list):
                last = last[-1]
        if last.overflowed():
            raise SystemExit('candidate capacity overflow in the benchmark configuration')
        extra['host_enqueue_ms_per_step'] = round(1e3 * enqueue / steps, 5)
        extra['detections_per_image_last_step'] = round(float(last.count.float().mean().item()), 2)
        # one batch at a time (yh_detect, every call complete before the next), for reference
        n_serial = max(10, min(200, steps // 4))
        ts = time.perf_counter()
        for i in range(n_serial):
            hd.detect(sets[i % rotate], w['thr'], w['iou'], w['max_boxes'], outputs=outputs)
        barrier()
        extra['serial_ms_per_step'] = round(1e3 * (time.perf_counter() - ts) / n_serial, 5)
        if kname == 'k_stream':
            # several launches per step: the dominant kernel under its own event pairs (yh_detect_timed)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for a, b in evs:                                  # a first record creates the hipEvent_t handles
                a.record(); b.record()
            for a, b in evs:
                hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes'], outputs=outputs, stream_events=(a, b))
            evs = evs[2:]
            for i, (a, b) in enumerate(evs):
                hd.detect(sets[i % rotate], w['thr'], w['iou'], w['max_boxes'], outputs=outputs, stream_events=(a, b))
            torch.cuda.synchronize(dev)
            kern_runs = [a.elapsed_time(b) for a, b in evs]
            timing = 'HIP event pairs around the kernel in %d yh_detect_timed calls after the timed regions' % len(evs)
        else:
            kern_runs, _ = solo_kernel_ms(hd, w, sets, dev, per_run=max(40, min(200, steps // 5)))
            timing = ('HIP event pairs around 5 runs of back-to-back launches on one stream after the timed regions (median '
                      'run; one launch = this kernel with the previous batch\'s select stage riding in its first workgroups)')
    else:
        extra['mean_loss_last_step'] = float(last.item())
        # the loss kernel alone: event pairs around back-to-back yh_loss calls without the reducer
        runs = []
        if args.loss_prepared:
            # the preparation is spent by the call it serves: a pool of head objects (a workspace each) is prepared
            # first, then their loss calls run back to back between one pair of events
            pool = [y.DetectionHead(C, priors, box_mult, gt_rate) for _ in range(25)]
            outs_pool = [None] * len(pool)

            def call(j, h):
                if w.get('backward'):
                    h.loss_and_grad(sets[j % rotate], gts[j % rotate], 1., 1., 5., 1., 0.5, True, grad_scale=0.5 / B)
                else:
                    outs_pool[j] = h.loss(sets[j % rotate], gts[j % rotate], 1., 1., 5., 1., 0.5, True, out=outs_pool[j])
            for j, h in enumerate(pool):                      # workspaces and outputs exist before anything is timed
                call(j, h)
            for _ in range(6):
                for j, h in enumerate(pool):
                    h.loss_prepare(head_shapes, gts[j % rotate])
                torch.cuda.synchronize(dev)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for j, h in enumerate(pool):
                    call(j, h)
                b.record()
                torch.cuda.synchronize(dev)
                runs.append(a.elapsed_time(b) / len(pool))
            runs = runs[1:]
        for _ in range(0 if args.loss_prepared else 5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(100):
                if w.get('backward'):
                    hd.loss_and_grad(sets[i % rotate], gts[i % rotate], 1., 1., 5., 1., 0.5, True, grad_scale=0.5 / B)
                else:
                    reuse['out'] = hd.loss(sets[i % rotate], gts[i % rotate], 1., 1., 5., 1., 0.5, True, out=reuse.get('out'))
            b.record()
            torch.cuda.synchronize(dev)
            runs.append(a.elapsed_time(b) / 100)
        kern_runs = runs
        timing = (('HIP event pairs around 5 runs of 25 back-to-back library calls on one stream, each on a workspace prepared before the run (median run), no all-reduce: '
                   if args.loss_prepared else
                   'HIP event pairs around 5 runs of 100 back-to-back library calls on one stream (median run), no all-reduce: ') +
                  ('two launches per call, k_loss_main (loss sums and the gradient tensor in one pass) and '
                   'k_loss_grad_pos_finalize (the assigned rows\' terms); traffic adds up both'
                   if w.get('backward') else 'one launch per call, k_loss_main, whose last workgroup combines the partial sums'))

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if (world > 1 and dist.get_backend() == 'nccl') else 'cpu')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        images_per_step = args.global_batch or world * B
        value = images_per_step * steps / elapsed
        kern_ms = float(np.median(kern_runs))
        config = dict(workload=w['name'], workload_key=args.workload, batch_per_gpu=B, global_batch=images_per_step,
                      boxes_per_image=N, num_classes=C, bytes_per_image=bytes_img,
                      parallelism='images sharded across %d GPU(s), no data-path collective' % world
                      if w['kind'] in ('detect', 'nms') else
                      'images sharded across %d GPU(s), one all-reduce (RCCL) of the fp64 pair (loss sum, image count) per '
                      '%d step(s)' % (world, max(1, args.steps_per_collective)),
                      input_rotation=rotation_note, **extra)
        if surv0 is not None:
            config['row_survival'] = round(row_survival(sets, C, w['thr']), 5)
        res = {
            'metric': 'images/sec (decode+IoU+NMS) YOLOv3-416 COCO-80 at 1/2/4/8 GPU; % HBM peak',
            'value': round(value, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(1e3 * elapsed / steps, 5), 'higher_is_better': True, 'scaling': 'strong' if args.global_batch else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'config': config,
            'timing': dict(regions_ms_per_step=[round(r, 5) for r in region_ms], **percentiles(region_ms),
                           note='region 0 is the one value / ms_per_step report; every region is %d steps between '
                                'barriers' % steps),
        }
        # the committed counter passes are of the workload's own batch
        rec = load_traffic(args.workload, kname) if B == w['batch'] else None
        if rec and w.get('backward'):                       # the call's second launch
            rec2 = load_traffic(args.workload, 'k_loss_grad_pos')
            if rec2:
                rec = dict(rec, bytes_per_launch=rec['bytes_per_launch'] + rec2['bytes_per_launch'])
        algo = float(B * bytes_img)
        model = rows_traffic_model(rows_per_launch, config.get('row_survival', 1.0), K) if kname == 'k_detect_rows' else (
            rows_per_launch * 128.0 if kname == 'k_loss_main' else algo)
        traffic = float(rec['bytes_per_launch']) if rec else None
        priced = traffic if traffic is not None else model
        ach = priced / (kern_ms * 1e-3) / 1e9
        res['roofline'] = dict(
            bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4),
            traffic=int(traffic) if traffic is not None else None,
            traffic_source=(rec.get('source') if rec else 'no committed PMC pass for this (workload, kernel, batch): achieved / '
                                                           'frac price traffic_model instead'),
            traffic_model=int(model), kernel=kdesc, kernel_key=kname, kernel_ms=round(kern_ms, 5),
            kernel_ms_runs=[round(r, 5) for r in kern_runs], kernel_timing=timing,
            algorithmic_bytes_per_launch=int(algo), algorithmic_gbs=round(algo / (kern_ms * 1e-3) / 1e9, 1),
            algorithmic_frac=round(algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            step_traffic_gbs=round(priced / (1e-3 * region_ms[0]) / 1e9, 1),
            step_traffic_frac=round(priced / (1e-3 * region_ms[0]) / 1e9 / HBM_PEAK_GBS, 4),
            whole_pipeline_algorithmic_frac=round(value / world * bytes_img / 1e9 / HBM_PEAK_GBS, 4),
            note=('achieved / frac: bytes the kernel moves through HBM per launch (traffic) over its duration on its own; '
                  'algorithmic_*: N*(C+5)*4 bytes per image over the same duration' +
                  ('; the kernel reads a line per row plus the rows that survive the objectness prefilter (row_survival '
                   '%.4f), so the algorithmic figure can exceed the peak and falls as more rows survive (survival_sweep); '
                   'step_traffic_*: the same bytes over the step time of the timed region, where %d streams overlap launches'
                   % (config.get('row_survival', 0.0), nstreams) if kname == 'k_detect_rows' else '')))
        if surv0 is not None:
            res['roofline']['row_survival'] = config['row_survival']
        valu = load_valu_busy(args.workload, kname)
        if valu:
            res['roofline']['valu'] = valu
        res['roofline']['achievable_read_gbs'] = round(measured_read_bandwidth(sets, dev), 1)
        res['roofline']['frac_of_achievable'] = round(ach / res['roofline']['achievable_read_gbs'], 4)
        if args.workload == 'detect416' and kname == 'k_detect_rows' and not args.no_sweep and world == 1:
            del sets
            torch.cuda.empty_cache()
            res['survival_sweep'] = survival_sweep(w, grids, priors, box_mult, gt_rate, dev, B, rows_per_launch, K)
            worst = max(res['survival_sweep'], key=lambda r: r['kernel_us'])
            res['worst_case'] = dict(
                row_survival=worst['row_survival'], kernel_us=worst['kernel_us'],
                images_per_sec=round(B / (worst['kernel_us'] * 1e-6), 1),
                algorithmic_frac=round(algo / (worst['kernel_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                note='the slowest point of survival_sweep (every row passes the objectness prefilter, so the whole head is '
                     'read): launches back to back on one stream; images_per_sec = batch / kernel time, algorithmic_frac = '
                     'N*(C+5)*4 bytes per image over that time against the 8 TB/s peak; `value` above holds at row_survival '
                     '%.4f' % config.get('row_survival', 0.0))
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(w, grids, priors, box_mult, gt_rate)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
