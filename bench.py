#!/usr/bin/env python3
"""Benchmark of the detection-head hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one pass of the path over one batch of synthetic head tensors that
are resident in HBM before the timed region starts.  For N > 1 the driver
launches one rank per GPU (torch.distributed.run); images are sharded by rank
(weak scaling: every rank owns a full per-GPU batch), the detect workloads need
no collective and the loss workload all-reduces one scalar over RCCL.

Workloads (BASELINE.json configs):
  detect416  config 2  YOLOv3-416 COCO-80, batch 64 / GPU, decode + score filter + per-class NMS   [default]
  loss416    config 3  YOLOv3-416 loss, 100 GT / image, batch 32 / GPU
  train416   config 3 with the backward to the head tensors (reads the head twice, writes its gradient)
  detect_v2  config 1  YOLOv2-416 VOC-20, batch 1 (latency-bound: one image)
  detect608  config 4  YOLOv3-608 COCO-80, batch 64 / GPU
  stress     config 5  100x100 grid x 10 priors, ~1e5 candidates / image at thr 0.001, batch 16 / GPU

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the pass
over the head tensor: objectness-first by default, which does not read the
whole tensor, so `achieved` = algorithmic bytes / time can exceed the peak and
`traffic` tells what was read) with HIP events recorded around it on the
launch stream inside the timed region; `cpu_baseline` times the C restatement
of the reference's arithmetic (oracle/, "port") on this host's cores for a
bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import gc
import json
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
EVENT_EVERY = 10                 # steps between two event-bracketed launches of the dominant kernel
PRIME_STEPS = 64                 # untimed steps before the W warm-up steps: pipelines and host caches reach steady state
ROTATE = 4                       # distinct input sets, so that no step re-reads data the 256 MiB Infinity Cache still holds

WORKLOADS = {
    'detect416': dict(kind='detect', size=416, C=80, batch=64, thr=0.5, iou=0.5, max_boxes=10, dist={},
                      name='YOLOv3 416x416 COCO-80 decode+NMS, synthetic head tensors, 10647 boxes/img'),
    'detect_v2': dict(kind='detect', size=416, C=20, batch=1, thr=0.5, iou=0.5, max_boxes=10, dist={}, version=2,
                      name='YOLOv2 416x416 VOC-20 batch 1 decode+NMS (the reference\'s own CPU-runnable case), 845 boxes/img'),
    'detect608': dict(kind='detect', size=608, C=80, batch=64, thr=0.5, iou=0.5, max_boxes=10, dist={},
                      name='YOLOv3 608x608 COCO-80 decode+NMS, synthetic head tensors, 22743 boxes/img'),
    'stress': dict(kind='detect', size=None, C=80, batch=16, thr=0.001, iou=0.5, max_boxes=100, dist=y.synth.STRESS,
                   name='NMS stress: 100x100 grid x 10 priors, ~1e5 candidates/img at score_thresh 0.001, 80 classes'),
    'loss416': dict(kind='loss', size=416, C=80, batch=32, G=100, pad=101,
                    name='YOLOv3 416x416 training loss path: 100 GT boxes/img IoU ignore-mask + target assignment'),
    'train416': dict(kind='loss', size=416, C=80, batch=32, G=100, pad=101, backward=True,
                     name='YOLOv3 416x416 training loss path, forward + backward to the head tensors (autograd)'),
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


def cpu_baseline(w, grids, priors, box_mult, gt_rate, budget_s=12.0):
    """Times the C restatement on the host cores for a bounded sample."""
    from oracle import c_oracle as co
    threads = max(1, min(16, int(os.environ.get('OMP_NUM_THREADS', '0')) or (os.cpu_count() or 1)))
    cfg = co.make_config(grids, priors, box_mult, gt_rate, w['C'])
    P = len(priors[0])
    Bs = 2 * threads
    heads = y.synth.make_heads(4242, Bs, grids, P, w['C'], **w.get('dist', {}))
    if w['kind'] == 'detect':
        fn = lambda: co.detect(cfg, heads, w['thr'], w['iou'], w['max_boxes'], threads=threads, unpack=False)
    else:
        gt = y.synth.make_ground_truth(4243, Bs, w['G'], w['pad'], w['size'], w['C'])
        fn = lambda: co.loss(cfg, heads, gt, (1., 1., 5., 1.), 0.5, True, threads=threads)
    fn()                                                    # warm-up (page-in, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        fn(); reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 10000:
            break
    return dict(value=round(Bs * reps / el, 2), unit='images/sec', cores=threads, kind='port',
                sample='%d synthetic images x %d passes (%.1f s) of the same workload, C restatement of the '
                       'reference op sequence (oracle/yolo_head_oracle.c), OpenMP over images' % (Bs, reps, el))


def measured_read_bandwidth(sets, dev, reps=20):
    """GB/s a read-only kernel reaches on these very buffers (yh_probe_read, HIP events): the
    achievable figure next to the nominal peak."""
    import ctypes
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


def detect_kernel_name(workload):
    """The dominant kernel of a detect workload, as csrc/yh_abi.hip::plan_stream picks it."""
    if workload == 'stress':                                  # capacity beyond the single-launch select stage
        return 'k_stream<detect> (workgroup tiles: decode + score filter + candidate compaction'
    if os.environ.get('YH_DETECT_KERNEL', '1') == '1':
        return 'k_detect_rows (objectness column, then the surviving rows: decode + score filter + candidate compaction'
    return 'k_stream_wave (decode + score filter + candidate compaction'


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile, if one exists."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='detect416', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='images per GPU (default: the workload\'s)')
    ap.add_argument('--in-flight', type=int, default=2, choices=(1, 2),
                    help='detect workloads: 2 = software pipeline (a step streams batch i and selects batch '
                         'i-1 in one launch), 1 = every batch runs to completion before the next')
    ap.add_argument('--streams', type=int, default=None,
                    help='detect workloads, --in-flight 2: pipelines on streams of their own, fed in turn (the '
                         'latency-bound tail of one launch overlaps the bandwidth-bound start of the next); '
                         '1 = one stream, launches back to back; default 3 (measured 17.2 us per step against '
                         '17.7 with two and 25 with one on the default workload)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    # YH_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    rank, world, local_rank = y.parallel.init_from_env(os.environ.get('YH_DIST_BACKEND'))
    if world != max(1, args.gpus):
        raise SystemExit('--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the path has no CPU fallback')
    dev = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    w = dict(WORKLOADS[args.workload])
    B = args.batch or w['batch']
    grids, priors, box_mult, gt_rate = build_head(w)
    P, C = len(priors[0]), w['C']
    hd = y.DetectionHead(C, priors, box_mult, gt_rate)
    N = sum(h * ww * P for h, ww in grids)
    bytes_img = N * (C + 5) * 4

    sets = [y.synth.torch_heads(dev, 1000 * rank + i, B, grids, P, C, **w.get('dist', {})) for i in range(ROTATE)]
    steps, warmup = args.steps, args.warmup
    # HIP events around the dominant kernel on a sample of the timed steps: an event record costs a
    # few microseconds of stream time, comparable to the tail of the pipeline, so not on every step
    ev = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for i in range(0, steps, EVENT_EVERY)}
    for a, b in ev.values():                                # force creation of the hipEvent_t handles
        a.record(); b.record()

    if w['kind'] == 'detect':
        # Software pipeline over the sequence of batches, as a serving loop would run the path: a
        # step streams batch i and, inside the same launch, selects batch i-1 (yh_detect_pipelined).
        # --in-flight 1 runs every batch to completion before the next (yh_detect).
        lanes = 2 if args.in_flight > 1 else 1
        outputs = hd._alloc_outputs(B, w['max_boxes'], dev, False)
        nstreams = max(1, args.streams or w.get('streams', 3)) if lanes > 1 else 1
        pipe = None
        if lanes > 1:
            pipe = (y.DetectLanes(hd, w['thr'], w['iou'], w['max_boxes'], lanes=nstreams, device=dev) if nstreams > 1
                    else y.DetectPipeline(hd, w['thr'], w['iou'], w['max_boxes']))

        def step(i, events=None):
            if pipe is not None and nstreams > 1:
                return pipe.submit(sets[i % ROTATE])          # the inputs are resident: nothing to wait for
            if pipe is not None:
                return pipe.submit(sets[i % ROTATE], stream_events=events)
            return hd.detect(sets[i % ROTATE], w['thr'], w['iou'], w['max_boxes'], outputs=outputs,
                             stream_events=events)
        extra = {'batches_in_flight': lanes * nstreams, 'streams': nstreams}
    else:
        gts = [torch.from_numpy(y.synth.make_ground_truth(77 + 10 * rank + i, B, w['G'], w['pad'], w['size'], C)).to(dev)
               for i in range(ROTATE)]

        if w.get('backward'):
            for hs in sets:
                for h in hs:
                    h.requires_grad_(True)

        def step_autograd(i):
            hs = sets[i % ROTATE]
            for h in hs:
                h.grad = None
            total, _ = hd.loss_with_grad(hs, gts[i % ROTATE], 1., 1., 5., 1., 0.5, True)
            (0.5 * total.reshape(()) / y.parallel.global_batch(B)).backward()
            return y.parallel.global_mean_loss(total.detach(), B)

        def step(i, events=None):
            if w.get('backward'):
                # forward + backward in one library call: d(.5 * mean loss) / d(head) for every head
                out, _grads = hd.loss_and_grad(sets[i % ROTATE], gts[i % ROTATE], 1., 1., 5., 1., 0.5, True,
                                               grad_scale=0.5 / y.parallel.global_batch(B))
                return y.parallel.global_mean_loss(out['sum'], B, out['mean'])
            out = reuse['out'] = hd.loss(sets[i % ROTATE], gts[i % ROTATE], 1., 1., 5., 1., 0.5, True, out=reuse.get('out'))
            return y.parallel.global_mean_loss(out['sum'], B, out['mean'])   # the data-parallel scalar all-reduce
        reuse = {}                                           # the loss outputs are written in place step after step
        extra = {}
        if w.get('backward'):
            for i in range(5):
                step_autograd(i)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(30):
                step_autograd(i)
            torch.cuda.synchronize(dev)
            extra = {'through_autograd_ms_per_step': round((time.perf_counter() - t0) / 30 * 1e3, 5),
                     'note': 'timed: loss_and_grad (yh_loss_with_grad: the gradient tensor is filled during the '
                             'forward pass); through_autograd = the same step via torch.autograd '
                             '(loss_with_grad + backward(): forward, then backward), host-bound by the engine'}

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if w['kind'] == 'detect':
        hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()   # settles the candidate capacity (untimed)
    last = None
    # Pipelined detect: a step is exactly one launch of the dominant kernel, so one pair of events
    # around the K back-to-back launches gives its average duration (launch gaps included) without
    # putting event records between the launches.
    span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    one_launch_per_step = (w['kind'] == 'detect' and pipe is not None and
                           hd.default_max_candidates(N) <= hd.SMALL_CAPACITY)   # the select stage rides along
    # With several streams the pair sits on the first lane's stream and brackets that lane's launches:
    # their average duration while the other lanes' launches overlap them.
    multi = w['kind'] == 'detect' and pipe is not None and nstreams > 1
    span_stream = pipe.streams[0] if multi else torch.cuda.current_stream(dev)
    for e in span:                                          # creates the hipEvent_t handles now, not in the timed region
        e.record(span_stream)
    gc.collect()
    gc.disable()                                            # no collector pause inside the few-millisecond timed region
    # Untimed priming, whatever W is: every lane's pipeline reaches its steady state (both workspaces
    # and output sets allocated, the prepared call of each slot built), and the host path is warm.
    for i in range(PRIME_STEPS):
        step(i)
    for i in range(warmup):
        last = step(i)
    barrier()
    span_launches = sum(1 for i in range(steps) if (pipe._i + i) % nstreams == 0) if multi else steps
    t0 = time.perf_counter()
    if one_launch_per_step:
        span[0].record(span_stream)
    for i in range(steps):
        if w['kind'] == 'detect' and not one_launch_per_step and not multi and i in ev:
            last = step(i, ev[i])
        else:
            last = step(i)
    if one_launch_per_step:
        span[1].record(span_stream)
    enqueue = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()

    if w['kind'] == 'detect' and lanes > 1:
        # one batch at a time, for reference (untimed, after the timed region): every step waits for the one before it
        n_serial = max(10, steps // 4)
        ts = time.perf_counter()
        for i in range(n_serial):
            hd.detect(sets[i % ROTATE], w['thr'], w['iou'], w['max_boxes'], outputs=outputs)
        barrier()
        extra['serial_ms_per_step'] = round(1e3 * (time.perf_counter() - ts) / n_serial, 5)

    if w['kind'] == 'detect':
        if pipe is not None:
            last = pipe.flush()                             # the last batch's select stage (outside the timed region,
                                                            # which selected the last warm-up batch instead)
            if isinstance(last, list):
                last = last[-1]
        if last.overflowed():
            raise SystemExit('candidate capacity overflow in the benchmark configuration')
        extra['host_enqueue_ms_per_step'] = round(1e3 * enqueue / steps, 5)
        if one_launch_per_step and multi:
            # The launches of the timed region overlap each other across the streams; a kernel trace
            # (rocprofv3 serialises dispatches) shows the kernel on its own.  Both are reported: the
            # overlapped per-launch duration from the timed region, and the kernel's own duration from
            # a short single-stream section right after it (same inputs, same rotation), which is the
            # one `roofline` prices and the committed trace must agree with.
            extra['kernel_ms_overlapped'] = round(span[0].elapsed_time(span[1]) / span_launches, 5)
            solo = y.DetectPipeline(hd, w['thr'], w['iou'], w['max_boxes'], tag='solo')
            for i in range(10):
                solo.submit(sets[i % ROTATE])
            # five runs of back-to-back launches, each under its own event pair; the median run counts (a
            # run in which the host falls behind with enqueueing would time the host, not the kernel)
            per_run, runs = max(40, min(200, steps // 5)), []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(per_run):
                    solo.submit(sets[i % ROTATE])
                b.record()
                torch.cuda.synchronize(dev)
                runs.append(a.elapsed_time(b) / per_run)
            solo.flush()
            kern_ms, n_timed = float(np.median(runs)), 5 * per_run
            extra['kernel_ms_runs'] = [round(r, 5) for r in runs]
            timing = ('HIP event pairs around 5 runs of %d back-to-back launches on one stream right after the timed region '
                      '(median run; the timed region\'s launches overlap across %d streams: kernel_ms_overlapped)'
                      % (per_run, nstreams))
        elif one_launch_per_step:
            kern_ms, n_timed = span[0].elapsed_time(span[1]) / span_launches, span_launches
            timing = 'one HIP event pair around the %d back-to-back launches of the timed region' % steps
        elif multi:
            # several launches per step and several streams: the dominant kernel on its own, from event
            # pairs around every 10th launch of a single-stream section right after the timed region
            solo = y.DetectPipeline(hd, w['thr'], w['iou'], w['max_boxes'], tag='solo')
            for i in range(5):
                solo.submit(sets[i % ROTATE])
            for i in range(steps):
                solo.submit(sets[i % ROTATE], stream_events=ev.get(i))
            torch.cuda.synchronize(dev)
            solo.flush()
            kern_ms, n_timed = float(np.mean([a.elapsed_time(b) for a, b in ev.values()])), len(ev)
            timing = ('HIP event pairs around every %dth launch of a single-stream section right after the timed region '
                      '(whose steps overlap across %d streams)' % (EVENT_EVERY, nstreams))
        else:
            kern_ms, n_timed = float(np.mean([a.elapsed_time(b) for a, b in ev.values()])), len(ev)
            timing = 'HIP event pairs around every %dth launch of the timed region' % EVENT_EVERY
        extra['detections_per_image_last_step'] = round(float(last.count.float().mean().item()), 2)
    else:
        kern_ms = None
        extra['mean_loss_last_step'] = float(last.item())

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = world * B * steps / elapsed
        res = {
            'metric': 'images/sec (decode+IoU+NMS) YOLOv3-416 COCO-80 at 1/2/4/8 GPU; % HBM peak',
            'value': round(value, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(1e3 * elapsed / steps, 5), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': dict(workload=w['name'], workload_key=args.workload, batch_per_gpu=B, global_batch=B * world,
                           boxes_per_image=N, num_classes=C, bytes_per_image=bytes_img,
                           parallelism='images sharded across %d GPU(s), no data-path collective' % world
                           if w['kind'] == 'detect' else
                           'images sharded across %d GPU(s), one scalar all-reduce (RCCL) per step' % world,
                           input_rotation='%d distinct input sets (%.0f MB each) so the 256 MiB Infinity Cache cannot serve a step'
                                          % (ROTATE, B * bytes_img / 1e6), **extra),
        }
        if kern_ms is not None:
            overlapped = nstreams if multi else 1
            ach = B * bytes_img / (kern_ms * 1e-3) / 1e9
            res['roofline'] = dict(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                   frac=round(ach / HBM_PEAK_GBS, 4), traffic=load_traffic(args.workload),
                                   kernel=detect_kernel_name(args.workload)
                                          + ('; the previous batch\'s NMS stage rides in the same launch)' if one_launch_per_step else ')'),
                                   kernel_ms=round(kern_ms, 5), kernel_launches_timed=n_timed, kernel_timing=timing,
                                   algorithmic_bytes_per_launch=B * bytes_img,
                                   launches_in_flight=overlapped,
                                   whole_pipeline_frac=round(value / world * bytes_img / 1e9 / HBM_PEAK_GBS, 4))
            if overlapped > 1:
                res['roofline']['overlap_note'] = ('achieved / frac price the kernel on its own (kernel_ms, what a kernel trace shows); '
                                                   'in the timed region %d streams keep %d launches in flight together, so a step '
                                                   'takes less than kernel_ms: whole_pipeline_frac and traffic_gbs use the step time'
                                                   % (overlapped, overlapped))
            if res['roofline']['traffic']:
                step_ms = 1e3 * elapsed / steps if one_launch_per_step else kern_ms
                tr = res['roofline']['traffic'] / (step_ms * 1e-3) / 1e9
                res['roofline']['traffic_gbs'] = round(tr, 1)
                res['roofline']['traffic_frac'] = round(tr / HBM_PEAK_GBS, 4)
            if detect_kernel_name(args.workload).startswith('k_detect_rows'):
                res['roofline']['note'] = ('achieved prices the algorithmic bytes (the whole head, N*(C+5)*4 per image) per launch; '
                                           'the objectness-first kernel reads one cache line per row plus the surviving rows '
                                           '(see traffic), so achieved can exceed the HBM peak')
        else:
            ach = value / world * bytes_img / 1e9
            res['roofline'] = dict(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                   frac=round(ach / HBM_PEAK_GBS, 4), traffic=load_traffic(args.workload),
                                   kernel='whole loss step (k_loss_main + k_loss_finalize + the scalar mean), wall clock',
                                   algorithmic_bytes_per_launch=B * bytes_img)
        res['roofline']['achievable_read_gbs'] = round(measured_read_bandwidth(sets, dev), 1)
        res['roofline']['frac_of_achievable'] = round(res['roofline']['achieved'] / res['roofline']['achievable_read_gbs'], 4)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(w, grids, priors, box_mult, gt_rate)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
