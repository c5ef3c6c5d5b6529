"""get_loader(cfg, seed) -> (sup, unsup, val) or (sup, val) loaders
(reference: u2pl/dataset/builder.py:9-43, cityscapes.py, pascal_voc.py, base.py, augmentation.py)."""
import copy
import math
import os
import random

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

TOTAL_TRAIN = {"cityscapes": 2975, "pascal": 10582}   # cityscapes.py:116, pascal_voc.py:109


def parse_list(path):
    """base.py:12-35: naming scheme is picked from the LIST PATH."""
    lines = [l.strip() for l in open(path) if l.strip()]
    if "cityscapes" in path:
        return [(l, "gtFine/" + l[12:-15] + "gtFine_labelTrainIds.png") for l in lines], "cityscapes"
    if "pascal" in path or "VOC" in path:
        return [(f"JPEGImages/{l}.jpg", f"SegmentationClassAug/{l}.png") for l in lines], "pascal"
    raise ValueError("unknown dataset list: " + path)


class Pipeline:
    """ToTensor -> Normalize -> [Resize] -> [RandResize] -> [Flip] -> [Crop] on (1,C,H,W) tensors;
    python `random` draws in the reference's order (augmentation.py:51-266, cityscapes.py:47-77)."""

    def __init__(self, cfg):
        self.mean = torch.tensor(np.float32(cfg["mean"]))[None, :, None, None]
        self.std = torch.tensor(np.float32(cfg["std"]))[None, :, None, None]
        self.resize = cfg.get("resize", False)
        self.rand_resize = cfg.get("rand_resize", False)
        self.flip = bool(cfg.get("flip", False))
        self.crop = cfg.get("crop", False)
        for k in ("rand_rotation", "GaussianBlur", "cutout", "cutmix"):
            if cfg.get(k, False):
                raise NotImplementedError(f"dataset option '{k}' is not enabled by any shipped config")

    def __call__(self, image, label):
        image = torch.from_numpy(np.asarray(image).copy().transpose(2, 0, 1)[None]).float()
        label = torch.from_numpy(np.asarray(label).copy()[None, None]).float()
        image = (image - self.mean) / self.std
        if self.resize:
            image = F.interpolate(image, size=self.resize, mode="bilinear", align_corners=False)
            label = F.interpolate(label, size=self.resize, mode="nearest")
        if self.rand_resize:
            lo, hi = self.rand_resize
            s = lo + (1.0 - lo) * random.random() if random.random() < 0.5 else 1.0 + (hi - 1.0) * random.random()
            h, w = image.shape[-2:]
            size = (int(h * s), int(w * s))
            image = F.interpolate(image, size=size, mode="bilinear", align_corners=False)
            label = F.interpolate(label, size=size, mode="nearest")
        if self.flip and random.random() < 0.5:
            image, label = torch.flip(image, [3]), torch.flip(label, [3])
        if self.crop:
            ch, cw = self.crop["size"]
            h, w = image.shape[-2:]
            ph, pw = max(ch - h, 0), max(cw - w, 0)
            if ph or pw:  # labels are padded with 0, not ignore_label (augmentation.py:241-245)
                border = (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
                image, label = F.pad(image, border, value=0.0), F.pad(label, border, value=0)
            h, w = image.shape[-2:]
            if self.crop["type"] == "rand":
                ho, wo = random.randint(0, h - ch), random.randint(0, w - cw)
            else:
                ho, wo = (h - ch) // 2, (w - cw) // 2
            image, label = image[:, :, ho:ho + ch, wo:wo + cw], label[:, :, ho:ho + ch, wo:wo + cw]
        return image[0].contiguous(), label[0, 0].long().contiguous()


class SegDataset(Dataset):
    def __init__(self, data_root, data_list, transform, seed, n_sup, split, kind_hint=None):
        self.samples, self.kind = parse_list(data_list)
        self.root, self.transform = data_root, transform
        random.seed(seed)
        if split == "train" and (self.kind == "cityscapes" or n_sup is not None):
            if len(self.samples) < n_sup:   # tile then sample (cityscapes.py:24-31)
                self.samples = self.samples * math.ceil(n_sup / len(self.samples))
            self.samples = random.sample(self.samples, n_sup)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        ip, lp = self.samples[i]
        with open(os.path.join(self.root, ip), "rb") as f:
            image = Image.open(f).convert("RGB")
        with open(os.path.join(self.root, lp), "rb") as f:
            label = Image.open(f).convert("L")
        return self.transform(image, label)


def _loader(dset, cfg, train):
    # the reference always samples through DistributedSampler (shuffle=True, re-seeded by set_epoch: cityscapes.py:143-163);
    # with explicit num_replicas / rank it needs no process group, so single-GPU runs shuffle per epoch too
    ddp = torch.distributed.is_available() and torch.distributed.is_initialized()
    world, rank = (torch.distributed.get_world_size(), torch.distributed.get_rank()) if ddp else (1, 0)
    sampler = DistributedSampler(dset, num_replicas=world, rank=rank, shuffle=train)
    return DataLoader(dset, batch_size=cfg.get("batch_size", 1), num_workers=cfg.get("workers", 2), sampler=sampler,
                      shuffle=False, pin_memory=True, drop_last=train)


def get_loader(cfg, seed=0):
    d = cfg["dataset"]
    kind = "cityscapes" if d["type"].startswith("cityscapes") else "pascal"
    semi = d["type"].endswith("_semi")

    def split_cfg(split):
        c = copy.deepcopy(d)
        c.update(c.get(split, {}))
        return c

    tc, vc = split_cfg("train"), split_cfg("val")
    val = SegDataset(vc["data_root"], vc["data_list"], Pipeline(vc), seed, None, "val")
    if not semi:
        n_sup = tc.get("n_sup", TOTAL_TRAIN[kind]) if kind == "cityscapes" else None
        sup = SegDataset(tc["data_root"], tc["data_list"], Pipeline(tc), seed, n_sup, "train")
        return _loader(sup, tc, True), _loader(val, vc, False)
    # both sets are resampled to (total - n_sup) items (cityscapes.py:116-141, pascal_voc.py:109-134; Q12)
    n = TOTAL_TRAIN[kind] - tc.get("n_sup", TOTAL_TRAIN[kind])
    sup = SegDataset(tc["data_root"], tc["data_list"], Pipeline(tc), seed, n, "train")
    unsup = SegDataset(tc["data_root"], tc["data_list"].replace("labeled.txt", "unlabeled.txt"), Pipeline(tc), seed, n,
                       "train")
    if d.get("device_aug", False):
        # decoded uint8 samples + host-drawn geometry; the transform chain runs fused on the GPU (device_aug.py).
        # Needs one image size per list (Cityscapes); engine.run finishes the batches with augment_batch.
        from .device_aug import AugmentPlan, RawSegDataset
        plan = AugmentPlan(tc)
        ls, lu = _loader(RawSegDataset(sup, plan), tc, True), _loader(RawSegDataset(unsup, plan), tc, True)
        ls.device_plan = lu.device_plan = plan
        return ls, lu, _loader(val, vc, False)
    return _loader(sup, tc, True), _loader(unsup, tc, True), _loader(val, vc, False)
