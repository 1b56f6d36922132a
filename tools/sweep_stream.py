#!/usr/bin/env python3
"""Sweeps the streaming kernel's tuning knobs on config 2 (GPU box).

usage: sweep_stream.py NI:NBUF:UNITS [...]   (NI 4 or 8 = wave-private tiles, 1 = workgroup tiles;
UNITS = wavefront work units, or workgroups for NI 1).  Prints the kernel time by HIP events and
the whole-step time for each setting.
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import yolo_tf_amd as y
from yolo_tf_amd import _abi

dev = torch.device('cuda:0')
s = [8., 16., 32.]
priors = [(np.asarray(y.synth.V3_PRIORS[i], np.float32) / np.float32(s[i])) for i in range(3)]
hd = y.DetectionHead(80, priors, (32., 32., 16.), (32., 16., 8.))
grids = y.synth.v3_grids(416)
sets = [y.synth.torch_heads(dev, i, 64, grids, 3, 80) for i in range(4)]
outs = hd._alloc_outputs(64, 10, dev, False)
ref = None
for spec in sys.argv[1:]:
    ni, nbuf, units = spec.split(':')[:3]
    hd._ws.clear(); hd._plans.clear()
    wave = ni in ('4', '8')
    _abi.set_tuning(detect_kernel=2, stream_tile_kib=int(ni), stream_buffers=int(nbuf),
                    stream_units=int(units) if wave else 0, stream_blocks=0 if wave else int(units))
    for i in range(20):
        hd.detect(sets[i % 4], 0.5, 0.5, 10, max_candidates=2048, outputs=outs)
    torch.cuda.synchronize()
    n = 200
    evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in range(0, n, 10)}
    for a, b in evs.values():
        a.record(); b.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        e = evs[i] if i in evs else None
        r = hd.detect(sets[i % 4], 0.5, 0.5, 10, max_candidates=2048, outputs=outs, stream_events=e)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k = sorted(a.elapsed_time(b) * 1e3 for a, b in evs.values())
    sig = (int(r.count.sum().item()), float(r.scores.double().sum().item()))
    if ref is None:
        ref = sig
    print('dbg %s ni %s nbuf %s units %6s  kernel med %.2f min %.2f max %.2f us  step %.2f us  same_result %s' % (
        dbg, ni, nbuf, units, k[len(k) // 2], k[0], k[-1], 1e6 * el / n, sig == ref), flush=True)
