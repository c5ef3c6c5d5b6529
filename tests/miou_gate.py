"""Shared by oracle/gen_golden.py:gen_miou_gate (run next to the REFERENCE in the build container) and
tests/test_gpu_miou_gate.py (run on the GPU box): the synthetic, LEARNABLE Cityscapes-shaped task of the
north_star's "mIoU after one epoch" gate.  Nothing is stored but the seed: both sides rebuild the tensors with
numpy's PCG64 generator (bit-stable across machines and numpy versions).

Task: 19-class label space (Cityscapes trainIds), of which six classes occur; a scene is a grid of 24-pixel cells,
each cell one class, the image is the class colour (well separated in normalised RGB) + N(0, 0.35) pixel noise + a
smooth illumination ramp, the first 4 rows of every label map are `ignore` (255).  One epoch = 40 steps of 2 labeled
+ 2 unlabeled 193x193 crops; validation = 50 images of 193x193.  A randomly initialised R101-DeepLabv3+ moves from
~1 % mIoU to well above it within the epoch, so "within +-0.3 points of the reference" is not trivially true."""
import numpy as np
import torch

GATE = dict(arch="resnet101", S=193, B=2, C=19, steps=40, n_val=50, epochs=1, min_kept=20000, class_thr=0.3,
            data_seed=2025, init_seed=0, np_seed=31, torch_seed=41, dropout_seed=977)
USED = (0, 1, 2, 8, 10, 13)           # road, sidewalk, building, vegetation, sky, car
COLOUR = {0: (-1.2, -1.2, -1.2), 1: (1.2, -1.2, -1.2), 2: (-1.2, 1.2, -1.2), 8: (-1.2, -1.2, 1.2), 10: (1.2, 1.2, -1.2),
          13: (1.2, -1.2, 1.2)}
CELL = 24


def _scene(rng, S):
    g = (S + CELL - 1) // CELL
    cls = np.asarray(USED)[rng.integers(0, len(USED), (g, g))]
    lab = np.kron(cls, np.ones((CELL, CELL), dtype=np.int64))[:S, :S]
    img = np.zeros((3, S, S), np.float32)
    for c in USED:
        m = lab == c
        for k in range(3):
            img[k][m] = COLOUR[c][k]
    ramp = np.linspace(-0.3, 0.3, S, dtype=np.float32)
    img += ramp[None, :, None] * np.float32(rng.uniform(-1, 1)) + ramp[None, None, :] * np.float32(rng.uniform(-1, 1))
    img += rng.standard_normal((3, S, S), dtype=np.float32) * np.float32(0.35)
    lab = lab.copy()
    lab[:4] = 255
    return img, lab


def _batch(rng, B, S, labels=True):
    imgs, labs = zip(*[_scene(rng, S) for _ in range(B)])
    x = torch.from_numpy(np.stack(imgs))
    return (x, torch.from_numpy(np.stack(labs))) if labels else x


def gate_data(seed=GATE["data_seed"], steps=GATE["steps"], B=GATE["B"], S=GATE["S"]):
    """-> [(image_l, label_l, image_u)] * steps"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(steps):
        il, ll = _batch(rng, B, S)
        out.append((il, ll, _batch(rng, B, S, labels=False)))
    return out


def gate_val(seed=GATE["data_seed"] + 1, n=GATE["n_val"], S=GATE["S"], bs=5):
    """-> [(images, labels)] validation batches (50 images)"""
    rng = np.random.default_rng(seed)
    return [_batch(rng, bs, S) for _ in range(n // bs)]


def data_digest(data, val):
    """a few numbers that pin the regenerated tensors to the ones the fixture was written from"""
    return np.float64([float(data[0][0].double().sum()), float(data[-1][2].double().sum()), float(data[len(data) // 2][1].double().sum()),
                       float(val[0][0].double().sum()), float(val[-1][1].double().sum())])
