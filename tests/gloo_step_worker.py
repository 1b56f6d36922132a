"""Worker of test_two_rank_gloo_optimizer_step (launched by torch.distributed.run, gloo,
CPU): two ranks hold the same small convolution stack and different gradients; after
step(lr) both hold the same weights, equal to a single-process step on the summed gradient."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_tf_amd as y                                       # noqa: E402
from yolo_tf_amd import backbone as bb                        # noqa: E402


class Tiny(bb._Provider):
    def __init__(self):
        super().__init__('channels_last', weight_decay=0.1, momentum=0.9)
        self.layer = bb.ConvBN(3, 4, 3, 2)
        self._finish_init()

    def forward(self, images):
        return [self._emit(self.layer(self._intake(images)))]


def grads_for(net, rank):
    g = torch.Generator().manual_seed(100 + rank)
    return [torch.randn(p.shape, generator=g) for p in net.parameters()]


def main():
    rank, world, _ = y.parallel.init_from_env('gloo')
    torch.manual_seed(7)
    net = Tiny()
    for p, g in zip(net.parameters(), grads_for(net, rank)):
        p.grad = g.clone()
    for _ in range(2):                                        # two steps: the momentum buffer is exercised too
        net.step(0.05)

    torch.manual_seed(7)
    ref = Tiny()
    summed = [sum(gs) for gs in zip(*[grads_for(ref, r) for r in range(world)])]
    old, ref_world = dist.get_world_size, None
    for p, g in zip(ref.parameters(), summed):
        p.grad = g.clone()
    dist.get_world_size = lambda *a, **k: 1                   # the reference step: one process, summed gradient
    try:
        for _ in range(2):
            ref.step(0.05)
    finally:
        dist.get_world_size = old
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert all(torch.equal(both[0], t) for t in both)
    print('GLOO_STEP_OK rank %d' % rank)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
