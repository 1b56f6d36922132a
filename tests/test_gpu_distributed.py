"""The data-parallel step on hardware: RCCL (torch.distributed backend nccl) in a one-rank
process group on the single MI355X, so that its initialisation and the all-reduce of the loss pair
really execute, and bench.py itself under two gloo ranks sharing the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith('{')]
    assert lines, text
    return json.loads(lines[-1])


def test_one_rank_rccl_group_runs_the_loss_all_reduce(gpu):
    """A child process (the process group is global state): LossReducer through RCCL against the
    kernel's own mean, then one bench.py loss416 run with the all-reduce forced."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import yolo_tf_amd as y
from tests import cases
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29745', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
rank, world, _ = y.parallel.init_from_env('nccl', even_alone=True)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1
dev = torch.device('cuda:0')
case = cases.v3(size=160, C=6)
hd = case.head_obj()
red = y.parallel.LossReducer(dev, always_reduce=True)
gt = torch.from_numpy(case.gt(6, 5, 8, 12)).to(dev)
pend, want = [], []
for step in range(3):
    heads = [torch.from_numpy(h).to(dev) for h in case.heads(50 + step, 5)]
    out = hd.loss(heads, gt, 1., 1., 5., 1., 0.5, True)
    pend.append(red.global_mean(out))
    want.append(out['per_image'].double().mean().item())
for p, w in zip(pend, want):
    assert isinstance(p, y.parallel.PendingMean)
    np.testing.assert_allclose(p.item(), w, rtol=1e-6)
m = y.parallel.global_mean_loss(out)
np.testing.assert_allclose(float(m), want[-1], rtol=1e-6)
dist.destroy_process_group()
print('RCCL_OK')
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'loss416', '--steps', '50',
                        '--warmup', '5', '--regions', '1', '--force-allreduce', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_PORT='29746'))
    assert r.returncode == 0, r.stdout + r.stderr
    line = _last_json(r.stdout)
    assert line['config']['allreduce'] == 'forced: 1-rank nccl process group'
    assert line['config']['steps_per_collective'] == 1
    assert line['value'] > 0 and line['config']['mean_loss_last_step'] > 0


@pytest.mark.parametrize('workload', ['detect416', 'loss416', 'detect608-strong'])
def test_bench_under_two_gloo_ranks(gpu, workload):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run), with the gloo backend
    because both ranks share this box's one GPU: sharding, barrier, max-over-ranks timing, the
    scalar all-reduce of the loss workload, one JSON line from rank 0."""
    # the watchdog makes a rank that hangs dump its threads' stacks and exit, so that the failure names its place
    env = dict(os.environ, YH_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', YH_BENCH_WATCHDOG='120')
    port = {'detect416': '29747', 'loss416': '29757', 'detect608-strong': '29767'}[workload]   # a port of its own: the last run's may still be closing
    strong = workload.endswith('-strong')                   # a fixed global batch of 16 images split over the ranks
    workload = workload.split('-')[0]
    size = ['--global-batch', '16'] if strong else ['--batch', '8']
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', port, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '20', '--warmup', '2', '--regions', '1', '--workload', workload,
                        *size, '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 16 and line['config']['batch_per_gpu'] == 8
    assert line['scaling'] == ('strong' if strong else 'weak')
    if workload == 'loss416':
        assert line['config']['steps_per_collective'] == 1
    assert line['value'] > 0 and sum(1 for l in r.stdout.splitlines() if l.startswith('{')) == 1
