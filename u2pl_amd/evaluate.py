"""Sliding-window / whole-image evaluation (reference eval.py:158-320) on the HIP forward path.

  net_process          eval.py:158-181   model(x)["pred"] -> bilinear(align_corners=True) to the input size
  scale_crop_process   eval.py:184-224   zero-pad to the crop size, windows at stride ceil(crop * 2/3) (the last
                                         window is pulled back inside the image), sum of window logits / window
                                         count, un-pad, bilinear to the label size
  validate_city        eval.py:235-306   per image: sum over scales, argmax, intersection / union histograms
  mIoU                 eval.py:302-305   mean(I / (U + 1e-10))
Host side = the reference's control flow; every tensor op is a libu2pl_hip kernel (forward stack, bilinear,
window accumulate / normalise, confusion histogram).
"""
import math

import numpy as np
import torch

from . import hipops as H
from ._lib import call


@torch.no_grad()
def net_process(model, image):
    """image (1,3,h,w) on the GPU -> logits (1,C,h,w), planar."""
    out = model(image, need_aux=False, need_rep=False)["pred"]
    return H.bilinear_up(out, image.shape[2:])


def window_grid(new_h, new_w, crop_h, crop_w, stride_rate=2 / 3):
    """[(s_h, s_w)] window origins in the reference's visiting order."""
    stride_h, stride_w = int(math.ceil(crop_h * stride_rate)), int(math.ceil(crop_w * stride_rate))
    grid_h = int(math.ceil(float(new_h - crop_h) / stride_h) + 1)
    grid_w = int(math.ceil(float(new_w - crop_w) / stride_w) + 1)
    wins = []
    for ih in range(grid_h):
        for iw in range(grid_w):
            e_h = min(ih * stride_h + crop_h, new_h)
            e_w = min(iw * stride_w + crop_w, new_w)
            wins.append((e_h - crop_h, e_w - crop_w))
    return wins


@torch.no_grad()
def scale_crop_process(model, image, classes, crop_h, crop_w, h, w, stride_rate=2 / 3):
    """image (1,3,H,W) GPU tensor -> logits (classes, h, w)."""
    ori_h, ori_w = image.shape[2:]
    pad_h, pad_w = max(crop_h - ori_h, 0), max(crop_w - ori_w, 0)
    ph, pw = int(pad_h / 2), int(pad_w / 2)
    if pad_h > 0 or pad_w > 0:
        padded = torch.zeros((1, image.shape[1], ori_h + pad_h, ori_w + pad_w), dtype=torch.float32, device=image.device)
        padded[:, :, ph:ph + ori_h, pw:pw + ori_w] = image
        image = padded
    new_h, new_w = image.shape[2:]
    pred = torch.zeros((1, classes, new_h, new_w), dtype=torch.float32, device=image.device)
    count = torch.zeros((new_h, new_w), dtype=torch.float32, device=image.device)
    for s_h, s_w in window_grid(new_h, new_w, crop_h, crop_w, stride_rate):
        crop = image[:, :, s_h:s_h + crop_h, s_w:s_w + crop_w].contiguous()
        logits = net_process(model, crop).contiguous()
        call("u2pl_window_accumulate_f32", pred, count, classes, new_h, new_w, logits, s_h, s_w, crop_h, crop_w)
    call("u2pl_window_normalize_f32", pred, count, classes, new_h, new_w)
    pred = pred[:, :, ph:ph + ori_h, pw:pw + ori_w]
    return H.bilinear_up(pred.contiguous(), (h, w))[0]


@torch.no_grad()
def scale_whole_process(model, image, h, w):
    return H.bilinear_up(net_process(model, image), (h, w))[0]


@torch.no_grad()
def predict_image(model, image, classes, base_size, crop, scales=(1.0,), use_crop=True):
    """image (1,3,h,w) normalised GPU tensor -> summed logits (classes, h, w) (validate_city's inner loop)."""
    h, w = image.shape[2:]
    total = torch.zeros((classes, h, w), dtype=torch.float32, device=image.device)
    for scale in scales:
        long_size = round(scale * base_size)
        new_h = new_w = long_size
        if h > w:
            new_w = round(long_size / float(h) * w)
        else:
            new_h = round(long_size / float(w) * h)
        scaled = image if (new_h, new_w) == (h, w) else H.bilinear_up(image.contiguous(), (new_h, new_w))
        if use_crop:
            total += scale_crop_process(model, scaled, classes, crop[0], crop[1], h, w)
        else:
            total += scale_whole_process(model, scaled, h, w)
    return total


@torch.no_grad()
def evaluate(model, samples, classes, base_size, crop, scales=(1.0,), use_crop=True, ignore=255, on_prediction=None):
    """samples: iterable of (image (3,h,w) float tensor already mean/std normalised, label (h,w) integer array).
    Returns (mIoU, per-class IoU).  on_prediction(i, uint8 map) receives every argmax map (gray / colour dumps)."""
    model.eval()
    dev = next(model.parameters()).device
    hist = torch.zeros(3 * classes, dtype=torch.int64, device=dev)
    for i, (image, label) in enumerate(samples):
        image = torch.as_tensor(image, dtype=torch.float32).unsqueeze(0).to(dev)
        logits = predict_image(model, image, classes, base_size, crop, scales, use_crop)
        lab = torch.as_tensor(np.asarray(label)).to(dev).long().contiguous().unsqueeze(0)
        h, w = lab.shape[1:]
        call("u2pl_confusion_hist_f32", logits.contiguous(), lab, ignore, 1, classes, h, w, hist)
        if on_prediction is not None:
            on_prediction(i, logits.argmax(0).to(torch.uint8).cpu().numpy())
    hh = hist.cpu().double().reshape(3, classes)
    inter, union = hh[0], hh[1] + hh[2] - hh[0]
    iou = (inter / (union + 1e-10)).numpy()
    return float(np.mean(iou)), iou
