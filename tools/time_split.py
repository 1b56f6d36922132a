"""The stress workload as two calls on two streams (yh_detect_candidates on one, yh_detect_select on another),
with and without a higher priority for the select stage's stream: does the select stage of batch i hide
behind the streaming pass of batch i+1?   usage: time_split.py [hi|same] [workspaces] [steps]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi
mode = sys.argv[1] if len(sys.argv) > 1 else 'hi'
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device('cuda:0')
w = bench.WORKLOADS['stress']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
lib = _abi.load()
sets = [y.synth.torch_heads(dev, i, B, grids, P, C, **(w.get('dist') or {})) for i in range(2)]
N = hd.boxes_per_image(sets[0])
d0 = hd.detect(sets[0], w['thr'], w['iou'], w['max_boxes']).resolved()       # grows the capacity
cap = hd.default_max_candidates(N)
cfg = hd.config_for(sets[0])
nbytes = ctypes.c_size_t(0)
_abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), B, cap, w['max_boxes'], ctypes.byref(nbytes)))
wss = [torch.zeros(nbytes.value, dtype=torch.uint8, device=dev) for _ in range(NW)]
outs = [hd._alloc_outputs(B, w['max_boxes'], dev, False) for _ in range(NW)]
s_cand = torch.cuda.Stream(dev)
s_sel = torch.cuda.Stream(dev, priority=-1) if mode == 'hi' else torch.cuda.Stream(dev)
ev_c = [torch.cuda.Event() for _ in range(NW)]
ev_s = [torch.cuda.Event() for _ in range(NW)]
ptrs = [_abi.pointer_array([h.data_ptr() for h in s]) for s in sets]
torch.cuda.synchronize()

def step(i):
    k = i % NW
    s_cand.wait_event(ev_s[k])                                # the workspace's last select has read it
    _abi.check(lib.yh_detect_candidates(ctypes.byref(cfg), ptrs[i % 2], B, w['thr'], w['max_boxes'], cap,
                                        wss[k].data_ptr(), wss[k].numel(), ctypes.c_void_p(s_cand.cuda_stream)))
    ev_c[k].record(s_cand)
    s_sel.wait_event(ev_c[k])
    _abi.check(lib.yh_detect_select(ctypes.byref(cfg), B, w['iou'], w['max_boxes'], cap, wss[k].data_ptr(), wss[k].numel(),
                                    *hd._output_pointers(outs[k]), ctypes.c_void_p(s_sel.cuda_stream)))
    ev_s[k].record(s_sel)

for i in range(20):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s priority, %d workspaces: %.1f us per step, %.0f images/s' % (mode, NW, 1e6 * dt / steps, B * steps / dt))
got = [o.clone() if o is not None else None for o in outs[(steps - 1) % NW][:5]]
ok = torch.equal(got[0], d0.count) if (steps - 1) % 2 == 0 else None
print('counts equal to yh_detect on the same set:', ok, 'overflow bits', int(outs[0][6].max().item()))
