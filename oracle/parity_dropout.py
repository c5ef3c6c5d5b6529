"""TEST INFRASTRUCTURE ONLY -- explicit Dropout2d keep-masks for parity runs (hard part 10).

The six ``nn.Dropout2d(0.1)`` layers of the reference decoder / aux head
(u2pl/models/decoder.py:79,86,90,100,104,136) are active in the student and in the
train-mode teacher (train_semi.py:339-363).  torch draws their masks from the
CPU mt19937 stream on the reference and we draw them from the device generator,
so the streams cannot match.  In parity mode BOTH sides take the keep-mask of
layer ``tag`` at its ``k``-th train-mode call from this pure function of
``(seed, tag, k, N, C, p)``; everything else (the multiply by mask/(1-p), where
the layers sit, which passes are in train mode) is still the code under test.

    reference / torch-CPU port:  ``with patched_torch_dropout2d(KeyedMasks(seed)):`` replaces
        ``nn.Dropout2d.forward`` for modules tagged by ``tag_model`` (no reference file is modified);
    HIP product:                 ``u2pl_amd.nn.DROPOUT_HOOK = masks.hook``  (tests install it).
"""
import contextlib
import zlib

import torch
import torch.nn as nn


def tag_model(model, role):
    """role: 'student' | 'teacher'.  Tags every Dropout2d with '<role>:<module name>' (names are the
    reference's state-dict prefixes: decoder.head.3, decoder.classifier.3/.7, decoder.representation.3/.7,
    auxor.aux.3)."""
    tags = []
    for name, m in model.named_modules():
        if isinstance(m, nn.Dropout2d):
            if name.startswith("module."):
                name = name[len("module."):]
            m._parity_tag = f"{role}:{name}"
            tags.append(m._parity_tag)
    return tags


class KeyedMasks:
    def __init__(self, seed):
        self.seed = int(seed)
        self.calls = {}
        self.log = []

    def scale(self, tag, N, C, p):
        """(N, C) float32 CPU tensor: 0 or 1/(1-p)."""
        k = self.calls.get(tag, 0)
        self.calls[tag] = k + 1
        g = torch.Generator().manual_seed(zlib.crc32(f"{self.seed}|{tag}|{k}".encode()))
        keep = torch.rand((N, C), generator=g) >= p
        self.log.append((tag, k, N, C, int(keep.sum())))
        return keep.to(torch.float32) / (1.0 - p)

    def hook(self, mod, N, C):
        """u2pl_amd.nn.DROPOUT_HOOK signature: -> CPU (N, C) scale, or None to let the product draw."""
        tag = getattr(mod, "_parity_tag", None)
        if tag is None:
            return None
        return self.scale(tag, N, C, mod.p)


@contextlib.contextmanager
def patched_torch_dropout2d(masks):
    orig = nn.Dropout2d.forward

    def forward(self, x):
        tag = getattr(self, "_parity_tag", None)
        if tag is None or not self.training or self.p <= 0:
            return orig(self, x)
        s = masks.scale(tag, x.shape[0], x.shape[1], self.p).to(x.dtype)
        return x * s[:, :, None, None]

    nn.Dropout2d.forward = forward
    try:
        yield masks
    finally:
        nn.Dropout2d.forward = orig
