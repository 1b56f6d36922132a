"""Per-workgroup timeline of the loss kernel's negative branch (measurement build: make -C
yolo-tf_amd/csrc trace).  Config-3 shape; microseconds since the first workgroup's start."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, yolo_tf_amd as y
from yolo_tf_amd import _abi
_abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), 'libyolohead_trace.so')

dev = torch.device('cuda:0')
w = bench.WORKLOADS['loss416']
grids, priors, box_mult, gt_rate = bench.build_head(w)
B, P, C = w['batch'], len(priors[0]), w['C']
hd = y.DetectionHead(C, priors, box_mult, gt_rate)
lib = _abi.load()
lib.yh_trace_loss_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
sets = [y.synth.torch_heads(dev, i, B, grids, P, C) for i in range(8)]
gts = [torch.from_numpy(y.synth.make_ground_truth(70 + i, B, w['G'], w['pad'], w['size'], C)).to(dev) for i in range(8)]
for i in range(10):
    if i == 9:
        torch.cuda.synchronize()
        assert lib.yh_trace_loss_read(None, 0, 1) == 0
    if os.environ.get('PREP') == '1':
        hd.loss_prepare(sets[i % 8], gts[i % 8])
    hd.loss(sets[i % 8], gts[i % 8], 1., 1., 5., 1., 0.5, True)
torch.cuda.synchronize()
tr = np.zeros((1 << 14, 8), np.uint64)
assert lib.yh_trace_loss_read(tr.ctypes.data, tr.nbytes, 0) == 0
tr_all_mask = (tr[:, 7] > 0) & ((tr[:, 6] & np.uint64(255)) != 255)
pos = tr[(tr[:, 7] > 0) & ((tr[:, 6] & np.uint64(255)) == 255)]
t_first = min(int(tr[tr[:, 7] > 0][:, 0].min()), int(pos[:, 0].min()))
tr = tr[tr_all_mask]
t = tr[:, :6].astype(np.float64) / 100.0
t -= t_first / 100.0
def q(x):
    return 'min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f' % (x.min(), *np.percentile(x, [10, 50, 90]), x.max())
print('%d negative-branch wavefronts' % len(t))
for name, x in (('start', t[:, 0]), ('GT staged at', t[:, 1]), ('GT converted (barrier) at', t[:, 2]), ('IoU loop done at', t[:, 3]),
                ('cross-entropy done at', t[:, 4]), ('end', t[:, 5]), ('staging', t[:, 1] - t[:, 0]), ('conversion', t[:, 2] - t[:, 1]),
                ('setup + IoU loop', t[:, 3] - t[:, 2]), ('cross-entropy', t[:, 4] - t[:, 3]), ('reduce + write', t[:, 5] - t[:, 4])):
    print('  %-28s %s' % (name, q(x)))
slow = (tr[:, 6] >> np.uint64(8)).astype(np.int64)
tr[:, 6] &= np.uint64(255)
print('  turns on the exact path: %d wavefronts with any, %d turns in all, most in one wavefront %d' % ((slow > 0).sum(), slow.sum(), slow.max()))
for s in range(3):
    m = tr[:, 6] == s
    print('  scale %d: %4d workgroups, IoU loop %s' % (s, m.sum(), q((t[:, 3] - t[:, 2])[m])))

# where the wavefronts ran: HW_ID bits 4-5 SIMD, 8-11 CU, 12 shader array, 13-15 shader engine; bits 40+ the XCD
hw = ((tr[:, 7] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
xcd = ((tr[:, 7] >> np.uint64(40)) & np.uint64(15)).astype(np.int64)
simd = (xcd << 12) | (((hw >> 8) & 0xff) << 4) | ((hw >> 4) & 3)
ids, inv, cnt = np.unique(simd, return_inverse=True, return_counts=True)
print('  %d distinct SIMDs; negative-branch wavefronts per SIMD: %s' % (len(ids), {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))}))
loop = t[:, 3] - t[:, 2]
for c in sorted(set(cnt)):
    m = cnt[inv] == c
    print('    SIMDs with %d: loop %s' % (c, q(loop[m])))
    print('                   done at %s' % q(t[:, 3][m]))
late = np.argsort(t[:, 5])[-8:]
for i in late:
    print('    late: wavefront %5d scale %d xcd %d hw %08x  start %5.2f setup %5.2f loop %5.2f end %5.2f  exact turns %d' % (i, tr[i, 6], xcd[i], hw[i], t[i, 0], t[i, 2], t[i, 3], t[i, 5], slow[i]))
# which workgroups share a CU (the scalar data cache is per CU): launch order -> placement
cuid = (xcd << 8) | ((hw >> 8) & 0xff)
blk = np.nonzero(tr_all_mask)[0] // 4
for c in list(np.unique(cuid))[:3] + list(np.unique(cuid))[100:102]:
    m = cuid == c
    print('    CU %04x: workgroups %s' % (c, sorted(set(blk[m].tolist()))))
print('    XCD of workgroups 0..15: %s' % [int(xcd[blk == i][0]) for i in range(16)])

# the positive branch's workgroups and the CUs of the late negative ones
pt = pos[:, [0, 5]].astype(np.float64) / 100.0 - t_first / 100.0
print('  %d positive-branch workgroups: start %s' % (len(pt), q(pt[:, 0])))
print('  %s end   %s' % (' ' * 30, q(pt[:, 1])))
phw = ((pos[:, 7] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
pcu = ((((pos[:, 7] >> np.uint64(40)) & np.uint64(15)).astype(np.int64)) << 8) | ((phw >> 8) & 0xff)
for i in late[-8::4]:
    c = cuid[i]
    m = cuid == c
    print('    CU %04x of late wavefront %d: negative workgroups %s end %s; positive workgroups end %s' % (
        c, i, sorted(set(blk[m].tolist())), np.round(sorted(set(t[:, 5][m].tolist()))[-3:], 2), np.round(pt[:, 1][pcu == c], 2)))
