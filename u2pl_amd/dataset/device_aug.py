"""Device-side training data pipeline.  The decoded uint8 sample goes to the GPU as it is; the
reference's per-sample CPU transform chain (augmentation.py:51-266, composed in cityscapes.py:47-77:
ToTensor -> Normalize -> RandResize -> RandomHorizontalFlip -> Crop) runs as ONE fused HIP gather per batch
(`u2pl_augment_u8_f32`).  The random numbers are still drawn on the host with python `random` in the
reference's order, so a seeded run consumes the RNG stream exactly like `builder.Pipeline` does."""
import random

import numpy as np
import torch

from .._lib import call


class AugmentPlan:
    """Draws the per-sample geometry of builder.Pipeline.__call__ without touching pixels."""

    def __init__(self, cfg):
        self.mean = np.asarray(cfg["mean"], np.float32).copy()
        self.std = np.asarray(cfg["std"], np.float32).copy()
        self.rand_resize = cfg.get("rand_resize", False)
        self.flip = bool(cfg.get("flip", False))
        self.crop = cfg.get("crop", False)
        if cfg.get("resize", False):
            raise NotImplementedError("fixed `resize` is only used by val pipelines; the device pipeline is train-only")
        for k in ("rand_rotation", "GaussianBlur", "cutout", "cutmix"):
            if cfg.get(k, False):
                raise NotImplementedError(f"dataset option '{k}' is not enabled by any shipped config")
        if not self.crop:
            raise NotImplementedError("the device pipeline emits fixed-size crops (every train config crops)")

    def out_size(self):
        return tuple(self.crop["size"])

    def draw(self, h, w):
        """-> int32[8] = {rh, rw, flip, pad_top, pad_left, crop_y, crop_x, 0}; same `random` calls, same order."""
        rh, rw = h, w
        if self.rand_resize:
            lo, hi = self.rand_resize
            s = lo + (1.0 - lo) * random.random() if random.random() < 0.5 else 1.0 + (hi - 1.0) * random.random()
            rh, rw = int(h * s), int(w * s)
        flip = int(self.flip and random.random() < 0.5)
        ch, cw = self.crop["size"]
        ph, pw = max(ch - rh, 0), max(cw - rw, 0)
        H2, W2 = rh + ph, rw + pw
        if self.crop["type"] == "rand":
            ho, wo = random.randint(0, H2 - ch), random.randint(0, W2 - cw)
        else:
            ho, wo = (H2 - ch) // 2, (W2 - cw) // 2
        return np.array([rh, rw, flip, ph // 2, pw // 2, ho, wo, 0], np.int32)


def augment_batch(plan, images_u8, labels_u8, params):
    """images_u8 (B,H,W,3) uint8 and labels_u8 (B,H,W) uint8 on the GPU, params (B,8) int32 (host or device)
    -> (B,3,Sh,Sw) float32 normalised crops, (B,Sh,Sw) int64 labels."""
    from ..hipops import h2d

    dev = images_u8.device
    B, H, W, _ = images_u8.shape
    Sh, Sw = plan.out_size()
    if not params.is_cuda:
        params = h2d(params.contiguous(), dev)
    out = torch.empty((B, 3, Sh, Sw), dtype=torch.float32, device=dev)
    lab = torch.empty((B, Sh, Sw), dtype=torch.int64, device=dev)
    call("u2pl_augment_u8_f32", images_u8.contiguous(), labels_u8.contiguous(), params, B, H, W, Sh, Sw,
         plan.mean.ctypes.data, plan.std.ctypes.data, out, lab)
    return out, lab


class RawSegDataset(torch.utils.data.Dataset):
    """Same sample list / resampling as builder.SegDataset, but __getitem__ returns the decoded uint8 sample
    plus the drawn geometry; `augment_batch` finishes the job on the GPU.  All images of the list must share
    one size (Cityscapes: 1024 x 2048) so that the default collate can stack them."""

    def __init__(self, base, plan):
        self.base, self.plan = base, plan

    def __len__(self):
        return len(self.base)

    def __getitem__(self, i):
        import os

        from PIL import Image

        ip, lp = self.base.samples[i]
        with open(os.path.join(self.base.root, ip), "rb") as f:
            image = np.asarray(Image.open(f).convert("RGB")).copy()
        with open(os.path.join(self.base.root, lp), "rb") as f:
            label = np.asarray(Image.open(f).convert("L")).copy()
        params = self.plan.draw(image.shape[0], image.shape[1])
        return torch.from_numpy(image), torch.from_numpy(label), torch.from_numpy(params)
