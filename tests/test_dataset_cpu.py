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
