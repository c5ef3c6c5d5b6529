#!/usr/bin/env python
"""Evaluation CLI with the reference's surface (eval.py:26-156): --config --model_path --base_size --scales
--save_folder --crop; Cityscapes lists -> sliding-window evaluation, VOC lists -> whole-image evaluation."""
import argparse
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def get_parser():
    p = argparse.ArgumentParser(description="U2PL evaluation (MI355X HIP path)")
    p.add_argument("--base_size", type=int, default=2048)
    p.add_argument("--scales", type=float, default=[1.0], nargs="+")
    p.add_argument("--config", type=str, default="config.yaml")
    p.add_argument("--model_path", type=str, default="checkpoints/ckpt_best.pth")
    p.add_argument("--save_folder", type=str, default="checkpoints/results/")
    p.add_argument("--names_path", type=str, default="")
    p.add_argument("--crop", action="store_true", default=False)
    return p


def data_list(cfg):
    d = cfg["dataset"]["val"]
    root, out = d["data_root"], []
    for line in open(d["data_list"]):
        line = line.strip()
        if not line:
            continue
        if "cityscapes" in root:
            arr = [line, "gtFine/" + line[12:-15] + "gtFine_labelTrainIds.png"]
        else:
            arr = [f"JPEGImages/{line}.jpg", f"SegmentationClassAug/{line}.png"]
        out.append([os.path.join(root, a) for a in arr])
    return out


def main():
    from PIL import Image

    from u2pl_amd import evaluate as E
    from u2pl_amd.engine import load_state
    from u2pl_amd.models.model_helper import ModelBuilder

    args = get_parser().parse_args()
    cfg = yaml.load(open(args.config), Loader=yaml.Loader)
    ds = cfg["dataset"]
    mean, std = np.asarray(ds["mean"], np.float32), np.asarray(ds["std"], np.float32)
    classes = cfg["net"]["num_classes"]
    crop = ds["val"]["crop"]["size"]
    gray = os.path.join(args.save_folder, "gray")
    os.makedirs(gray, exist_ok=True)
    items = data_list(cfg)
    cfg["net"]["sync_bn"] = False
    model = ModelBuilder(cfg["net"])
    ck = torch.load(args.model_path, map_location="cpu")
    load_state(args.model_path, model, "teacher_state" if "teacher_state" in ck else "model_state")
    model = model.cuda()

    def samples():
        for ip, lp in items:
            img = (np.asarray(Image.open(ip).convert("RGB")).astype(np.float32) - mean) / std
            yield torch.from_numpy(img).permute(2, 0, 1).contiguous(), np.asarray(Image.open(lp).convert("L")).astype(np.uint8)

    def dump(i, pred):
        Image.fromarray(pred).save(os.path.join(gray, os.path.basename(items[i][0]).split(".")[0] + ".png"))

    city = "cityscapes" in ds["type"]
    miou, iou = E.evaluate(model, samples(), classes, args.base_size, crop, args.scales, use_crop=city or args.crop,
                           ignore=ds.get("ignore_label", 255), on_prediction=dump)
    for c, v in enumerate(iou):
        print(f" * class [{c}] IoU {v * 100:.2f}")
    print(f" * mIoU {miou * 100:.2f}")


if __name__ == "__main__":
    main()
