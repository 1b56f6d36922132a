"""Worker of test_two_rank_gloo_loss_mean_matches_full_batch (not collected)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_tf_amd as y                                    # noqa: E402
from oracle import c_oracle as co                          # noqa: E402
from tests import cases                                    # noqa: E402

rank, world, _ = y.parallel.init_from_env(backend='gloo')
assert world == 2 and dist.get_backend() == 'gloo'
case = cases.v3(size=160, C=6)
B = 7                                                       # uneven split: 4 + 3
heads = case.heads(5, B)
gt = case.gt(6, B, 8, 12)
cfg = case.oracle_cfg(co)
per_full, _, _, mean_full = co.loss(cfg, heads, gt, (1., 1., 5., 1.), 0.5, True)
b0, b1 = y.parallel.shard_range(B, rank, world)
per_local, _, _, _ = co.loss(cfg, [h[b0:b1] for h in heads], gt[b0:b1], (1., 1., 5., 1.), 0.5, True)
assert np.array_equal(per_local, per_full[b0:b1])          # images are independent
local_sum = torch.tensor([float(np.sum(per_local.astype(np.float64)))], dtype=torch.float32)
mean = y.parallel.global_mean_loss(local_sum, b1 - b0)
np.testing.assert_allclose(float(mean), float(mean_full), rtol=1e-6)
gathered = [torch.zeros(()) for _ in range(world)]
dist.all_gather(gathered, mean)
assert float(gathered[0]) == float(gathered[1])             # identical on every rank
# the step loop's form: the loss kernel's fp64 pair, all-reduced asynchronously in a ring of persistent buffers
red = y.parallel.LossReducer(torch.device('cpu'), depth=4)
pending = []
for step in range(5):
    sum64 = float(np.sum(per_local.astype(np.float64))) * (step + 1)
    out = dict(sum=torch.tensor([sum64], dtype=torch.float32), mean=None, per_image=torch.zeros(b1 - b0),
               reduce_pair=torch.tensor([sum64, b1 - b0], dtype=torch.float64))
    pending.append(red.global_mean(out))
want64 = float(np.sum(per_full.astype(np.float64))) / B
try:                                                        # five steps through four buffers: the first one is gone
    pending[0].item()
    raise AssertionError('a stale PendingMean must raise')
except RuntimeError:
    pass
for step, p in list(enumerate(pending))[1:]:
    assert isinstance(p, y.parallel.PendingMean)
    np.testing.assert_allclose(p.item(), want64 * (step + 1), rtol=1e-6)
    assert p.tensor().dtype == torch.float32 and p.tensor().dim() == 0
# gathered steps: one collective per three steps, issued by the third call or by flush() on every rank; a read
# never issues one (a rank reading alone would otherwise enter the all-reduce alone)
red3 = y.parallel.LossReducer(torch.device('cpu'), depth=8, batch=3)
pend3 = []
for step in range(4):
    sum64 = float(np.sum(per_local.astype(np.float64))) * (step + 1)
    out = dict(sum=torch.tensor([sum64], dtype=torch.float32), mean=None, per_image=torch.zeros(b1 - b0),
               reduce_pair=torch.tensor([sum64, b1 - b0], dtype=torch.float64))
    pend3.append(red3.global_mean(out))
np.testing.assert_allclose(pend3[2].item(), want64 * 3, rtol=1e-6)      # steps 0..2 went out with the third call
try:
    pend3[3].item()
    raise AssertionError('reading a step whose collective has not been issued must raise')
except RuntimeError:
    pass
red3.flush()
np.testing.assert_allclose(pend3[3].item(), want64 * 4, rtol=1e-6)
print('GLOO_OK rank %d mean %.6f' % (rank, float(mean)))
dist.destroy_process_group()
