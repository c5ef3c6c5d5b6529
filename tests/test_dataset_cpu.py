"""CPU tests of the host data pipeline against the reference's conventions."""
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_synth_dataset as M  # noqa: E402


def test_cityscapes_semi_loaders(tmp_path):
    from u2pl_amd.dataset import get_loader
    from u2pl_amd.dataset.builder import parse_list

    d, s = M.make_cityscapes(str(tmp_path), H=70, W=100)
    cfgp = M.write_city_config(str(tmp_path), d, s, crop=65)
    cfg = yaml.load(open(cfgp), Loader=yaml.Loader)
    samples, kind = parse_list(os.path.join(s, "labeled.txt"))
    assert kind == "cityscapes" and samples[0][1].startswith("gtFine/train/synth/") and samples[0][1].endswith("_gtFine_labelTrainIds.png")
    sup, unsup, val = get_loader(cfg, seed=2)
    assert len(sup.dataset) == len(unsup.dataset) == 4 and len(val.dataset) == 4   # both resampled to 2975 - n_sup
    img, lab = next(iter(sup))
    assert img.shape == (2, 3, 65, 65) and lab.shape == (2, 65, 65) and lab.dtype == torch.int64
    assert set(np.unique(lab.numpy())) <= set(range(19)) | {255}
    img_v, lab_v = next(iter(val))
    assert img_v.shape == (2, 3, 65, 65)
    assert abs(float(img.mean())) < 3.0   # normalised


def test_voc_sup_loader(tmp_path):
    from u2pl_amd.dataset import get_loader

    d, s = M.make_voc(str(tmp_path))
    cfg = dict(dataset=dict(type="pascal", batch_size=4, workers=0, mean=[123.675, 116.28, 103.53],
                            std=[58.395, 57.12, 57.375], ignore_label=255,
                            train=dict(data_root=d, data_list=os.path.join(s, "labeled.txt"), flip=True,
                                       rand_resize=[0.5, 2.0], crop=dict(type="rand", size=[65, 65])),
                            val=dict(data_root=d, data_list=os.path.join(os.path.dirname(s), "val.txt"),
                                     crop=dict(type="center", size=[65, 65]))))
    sup, val = get_loader(cfg, seed=0)
    assert len(sup.dataset) == 8       # whole list is used (pascal_voc.py:85)
    img, lab = next(iter(sup))
    assert img.shape == (4, 3, 65, 65) and lab.shape == (4, 65, 65)


def test_device_pipeline_plan_draws_like_the_cpu_pipeline():
    """dataset/device_aug.AugmentPlan consumes python `random` exactly like builder.Pipeline (the reference's
    augmentation order), and its geometry reproduces the Pipeline's output size / padding / crop origin."""
    import random

    import numpy as np
    from PIL import Image

    from u2pl_amd.dataset.builder import Pipeline
    from u2pl_amd.dataset.device_aug import AugmentPlan

    cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], rand_resize=[0.5, 2.0], flip=True,
               crop=dict(type="rand", size=[97, 113]))
    img = np.zeros((96, 150, 3), np.uint8)
    # a coordinate image: label value encodes nothing, the image's red channel encodes x so the flip is visible
    img[..., 0] = np.arange(150, dtype=np.uint8)[None, :]
    lab = np.full((96, 150), 7, np.uint8)
    for seed in range(12):
        random.seed(seed)
        out_img, out_lab = Pipeline(cfg)(Image.fromarray(img), Image.fromarray(lab))
        after_cpu = random.random()
        random.seed(seed)
        p = AugmentPlan(cfg).draw(96, 150)
        assert random.random() == after_cpu
        rh, rw, flip, pt, pl, ho, wo, _ = [int(v) for v in p]
        assert tuple(out_lab.shape) == (97, 113)
        # rows / columns of the crop that fall on the zero padding carry label 0 (augmentation.py:241-245)
        ys = np.arange(97) + ho - pt
        xs = np.arange(113) + wo - pl
        inside = ((ys >= 0) & (ys < rh))[:, None] & ((xs >= 0) & (xs < rw))[None, :]
        assert np.array_equal(out_lab.numpy() == 7, inside), (seed, p)
