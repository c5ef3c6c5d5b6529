"""Deterministic synthetic datasets in the reference's on-disk layouts (SURVEY hard part 11):
Cityscapes (leftImg8bit/<split>/<city>/*_leftImg8bit.png + gtFine/..._gtFine_labelTrainIds.png, list lines as in
data/splits/cityscapes/*/labeled.txt, list path contains "cityscapes") and VOC (JPEGImages / SegmentationClassAug)."""
import os
import sys

import numpy as np
import yaml
from PIL import Image


def scene(rng, H, W, C, cell=16):
    g = rng.integers(0, C, ((H + cell - 1) // cell, (W + cell - 1) // cell))
    lab = np.kron(g, np.ones((cell, cell), dtype=np.int64))[:H, :W].astype(np.uint8)
    lab[:4] = 255
    pal = (np.arange(C)[:, None] * np.array([37, 91, 151]) % 256).astype(np.uint8)
    img = pal[np.where(lab == 255, 0, lab)] + rng.integers(0, 40, (H, W, 3), dtype=np.uint8)
    return img.astype(np.uint8), lab


def make_cityscapes(root, n_l=4, n_u=4, n_val=4, H=140, W=200, C=19, seed=0):
    rng = np.random.default_rng(seed)
    droot = os.path.join(root, "data", "cityscapes")
    sroot = os.path.join(root, "data", "splits", "cityscapes", str(n_l))
    os.makedirs(sroot, exist_ok=True)
    lists = {"labeled": [], "unlabeled": [], "val": []}
    for split, name, n in (("train", "labeled", n_l), ("train", "unlabeled", n_u), ("val", "val", n_val)):
        for i in range(n):
            city = "synth"
            stem = f"{city}_{name}_{i:06d}_000019"
            ip = f"leftImg8bit/{split}/{city}/{stem}_leftImg8bit.png"
            lp = f"gtFine/{split}/{city}/{stem}_gtFine_labelTrainIds.png"
            img, lab = scene(rng, H, W, C)
            for p, a in ((ip, img), (lp, lab)):
                os.makedirs(os.path.dirname(os.path.join(droot, p)), exist_ok=True)
                Image.fromarray(a).save(os.path.join(droot, p))
            lists[name].append(ip)
    for k, v in lists.items():
        path = os.path.join(sroot if k != "val" else os.path.dirname(sroot), k + ".txt")
        open(path, "w").write("\n".join(v) + "\n")
    return droot, sroot


def make_voc(root, n=8, H=96, W=120, C=21, seed=0):
    rng = np.random.default_rng(seed)
    droot = os.path.join(root, "data", "VOC2012")
    sroot = os.path.join(root, "data", "splits", "pascal", str(n))
    for d in ("JPEGImages", "SegmentationClassAug"):
        os.makedirs(os.path.join(droot, d), exist_ok=True)
    os.makedirs(sroot, exist_ok=True)
    names = []
    for i in range(n):
        img, lab = scene(rng, H, W, C)
        Image.fromarray(img).save(os.path.join(droot, "JPEGImages", f"s{i:04d}.jpg"), quality=95)
        Image.fromarray(lab).save(os.path.join(droot, "SegmentationClassAug", f"s{i:04d}.png"))
        names.append(f"s{i:04d}")
    for k in ("labeled", "val"):
        open(os.path.join(sroot if k != "val" else os.path.dirname(sroot), k + ".txt"), "w").write("\n".join(names) + "\n")
    return droot, sroot


def write_city_config(root, droot, sroot, crop=97, arch="resnet50", epochs=1, n_sup=4, total_hack=None):
    exp = os.path.join(root, "experiments", "cityscapes", str(n_sup), "ours")
    os.makedirs(exp, exist_ok=True)
    ref = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "city_semi_template.yaml")))
    ref["dataset"]["train"].update(data_root=droot, data_list=os.path.join(sroot, "labeled.txt"),
                                   crop=dict(type="rand", size=[crop, crop]))
    ref["dataset"]["val"].update(data_root=droot, data_list=os.path.join(os.path.dirname(sroot), "val.txt"),
                                 crop=dict(type="center", size=[crop, crop]))
    ref["dataset"]["n_sup"] = 2975 - 4          # both loaders are resampled to 2975 - n_sup = 4 items (Q12)
    ref["dataset"]["workers"] = 0
    ref["trainer"]["epochs"] = epochs
    ref["criterion"]["kwargs"]["min_kept"] = 3000
    ref["net"]["sync_bn"] = False
    ref["net"]["encoder"]["type"] = f"u2pl.models.resnet.{arch}"
    ref["net"]["encoder"]["kwargs"]["pretrained"] = False
    path = os.path.join(exp, "config.yaml")
    yaml.safe_dump(ref, open(path, "w"))
    return path


if __name__ == "__main__":
    r = sys.argv[1]
    d, s = make_cityscapes(r)
    print(write_city_config(r, d, s))
