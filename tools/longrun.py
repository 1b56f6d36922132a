"""Long run of the default serving path (DetectLanes: three streams, a batch's select stage inside the next launch
of its lane) at a BASELINE shape: every batch's outputs are compared, bit for bit, with what detect() gave for the
same input set before the run.   usage: longrun.py [workload] [steps] [lanes]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y

wl = sys.argv[1] if len(sys.argv) > 1 else 'detect416'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
w = bench.WORKLOADS[wl]
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
nsets = 5
sets = [y.synth.torch_heads(dev, 900 + i, B, grids, P, C, **w.get('dist', {})) for i in range(nsets)]
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()              # settles the candidate capacity
want = []
for s in sets:
    d = hd.detect(s, w['thr'], w['iou'], w['max_boxes'])
    assert not d.overflowed()
    want.append([t.clone() for t in (d.count, d.scores, d.boxes, d.class_id, d.index)])
torch.cuda.synchronize()
pipe = y.DetectLanes(y.DetectionHead(C, priors, box_mult, gt_rate), w['thr'], w['iou'], w['max_boxes'], lanes=lanes, device=dev)
pipe.pipes[0].head._grown.update(hd._grown)                                     # the same capacity as the reference calls
bad, checked, t0 = 0, 0, time.time()

def check(d, i):
    global bad, checked
    ref = want[i % nsets]
    d._stream.synchronize()
    count = ref[0]
    ok = torch.equal(d.count, count)
    if ok:
        keep = torch.arange(d.scores.shape[1], device=dev)[None, :] < count[:, None].to(torch.int64)
        for got, exp in zip((d.scores, d.class_id, d.index), ref[1:2] + ref[3:5]):
            ok = ok and torch.equal(torch.where(keep, got, torch.zeros_like(got)), torch.where(keep, exp, torch.zeros_like(exp)))
        ok = ok and torch.equal(torch.where(keep[..., None], d.boxes, torch.zeros_like(d.boxes)), torch.where(keep[..., None], ref[2], torch.zeros_like(ref[2])))
    checked += 1
    if not ok:
        bad += 1
        print('step %d (set %d): outputs differ; counts equal: %s' % (i, i % nsets, torch.equal(d.count, count)), flush=True)

pending = []
for i in range(steps):
    d = pipe.submit(sets[i % nsets])
    pending.append(i)
    if d is not None:
        check(d, pending.pop(0))
    if i % 10000 == 9999:
        print('%d steps, %d checked, %d differ, %.0f s' % (i + 1, checked, bad, time.time() - t0), flush=True)
for d in pipe.flush():
    check(d, pending.pop(0))
print('done: %s, %d lanes, %d steps, %d checked, %d differ' % (wl, lanes, steps, checked, bad))
sys.exit(1 if bad else 0)
