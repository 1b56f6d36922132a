"""Synthetic head tensors and ground truth of the benchmark configurations.

Seeded numpy.random.Generator(PCG64(seed)) streams, float32.  Head logits per
prior: t_y, t_x ~ N(0, 1); t_h, t_w ~ N(0, 0.5); objectness ~ N(obj_mean,
obj_std); one "hot" class per box ~ N(hot_mean, hot_std), the others
~ N(cold_mean, cold_std)."""
import numpy as np

V3_PRIORS = [[[10., 13.], [16, 30.], [33., 23.]],
             [[30., 61.], [62., 45.], [59., 119.]],
             [[116., 90.], [156., 198.], [373., 326.]]]            # testYOLOv3.py:36-38
V2_PRIORS = [[1.08, 1.19], [3.42, 4.41], [6.63, 11.38], [9.42, 5.11], [16.62, 10.52]]   # testYOLOv2.py:44

DEFAULT = dict(obj_mean=-4.0, obj_std=2.0, hot_mean=2.0, hot_std=1.0, cold_mean=-6.0, cold_std=1.0)
# about one (box, class) pair per box passes a 0.001 score threshold
STRESS = dict(obj_mean=0.0, obj_std=2.0, hot_mean=2.0, hot_std=1.0, cold_mean=-12.0, cold_std=1.0)


def v3_grids(size):
    return [(size // 32, size // 32), (size // 16, size // 16), (size // 8, size // 8)]


def make_head(rng, B, H, W, P, C, obj_mean, obj_std, hot_mean, hot_std, cold_mean, cold_std):
    """One NHWC head tensor [B, H, W, P*(C+5)]."""
    R = B * H * W * P
    t = np.empty((R, C + 5), dtype=np.float32)
    t[:, :C] = rng.normal(cold_mean, cold_std, size=(R, C)).astype(np.float32)
    hot = rng.integers(0, C, size=R)
    t[np.arange(R), hot] = rng.normal(hot_mean, hot_std, size=R).astype(np.float32)
    t[:, C:C + 2] = rng.normal(0.0, 1.0, size=(R, 2)).astype(np.float32)
    t[:, C + 2:C + 4] = rng.normal(0.0, 0.5, size=(R, 2)).astype(np.float32)
    t[:, C + 4] = rng.normal(obj_mean, obj_std, size=R).astype(np.float32)
    return t.reshape(B, H, W, P * (C + 5))


def make_heads(seed, B, grids, P, C, **dist):
    rng = np.random.Generator(np.random.PCG64(seed))
    d = dict(DEFAULT); d.update(dist)
    return [make_head(rng, B, H, W, P, C, **d) for (H, W) in grids]


def make_ground_truth(seed, B, G, pad, size, C):
    """[B, pad, 5] rows (yc, xc, h, w, class) in pixels: G valid rows then -1
    padding (utils/image_augmentor.py:228-243); pad > G is required by the
    reference's argmin slice.  yc, xc ~ U(1, size-2); h, w log-uniform in
    [8, size/2]."""
    if pad <= G:
        raise ValueError('pad must exceed G: the reference finds the GT count with argmin over a -1 row')
    rng = np.random.Generator(np.random.PCG64(seed))
    gt = np.full((B, pad, 5), -1.0, dtype=np.float32)
    gt[:, :G, 0:2] = rng.uniform(1.0, size - 2.0, size=(B, G, 2)).astype(np.float32)
    gt[:, :G, 2:4] = np.exp(rng.uniform(np.log(8.0), np.log(size / 2.0), size=(B, G, 2))).astype(np.float32)
    gt[:, :G, 4] = rng.integers(0, C, size=(B, G)).astype(np.float32)
    return gt


def torch_heads(device, seed, B, grids, P, C, **dist):
    """Same distribution generated on the device (for full-size benchmark
    batches; not bit-identical to make_heads)."""
    import torch
    d = dict(DEFAULT); d.update(dist)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    heads = []
    for (H, W) in grids:
        R = B * H * W * P
        t = torch.empty((R, C + 5), dtype=torch.float32, device=device)
        t[:, :C].normal_(d['cold_mean'], d['cold_std'], generator=g)
        hot = torch.randint(0, C, (R,), device=device, generator=g)
        hv = torch.empty((R,), dtype=torch.float32, device=device).normal_(d['hot_mean'], d['hot_std'], generator=g)
        t[torch.arange(R, device=device), hot] = hv
        t[:, C:C + 2].normal_(0.0, 1.0, generator=g)
        t[:, C + 2:C + 4].normal_(0.0, 0.5, generator=g)
        t[:, C + 4].normal_(d['obj_mean'], d['obj_std'], generator=g)
        heads.append(t.reshape(B, H, W, P * (C + 5)))
    return heads
