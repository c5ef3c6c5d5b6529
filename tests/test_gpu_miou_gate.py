"""-m gpu: north_star acceptance gate -- mIoU on a fixed 50-image val subset after ONE epoch of
semi-supervised training must be within +-0.3 points of the CPU reference (here: the CPU port of the
reference step, oracle/step_ref.py) on a deterministic synthetic Cityscapes-layout dataset; both sides
start from the same initial weights and see the same batches, CutMix boxes and sampling indices
(dropout ON, p = 0.1: both sides take the keyed keep-masks of oracle/parity_dropout)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
DEV = "cuda"


def test_miou_after_one_epoch_matches_cpu_reference(tmp_path):
    import make_synth_dataset as M
    from oracle.step_ref import CpuStepRef, validate_ref
    from u2pl_amd.dataset import get_loader
    from u2pl_amd.engine import validate
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion

    d, s = M.make_cityscapes(str(tmp_path), n_l=6, n_u=6, n_val=50, H=110, W=150)
    cfgp = M.write_city_config(str(tmp_path), d, s, crop=97, epochs=1)
    cfg = yaml.load(open(cfgp), Loader=yaml.Loader)
    cfg["dataset"]["n_sup"] = 2975 - 6
    cfg["trainer"]["contrastive"]["current_class_threshold"] = 0.055   # near-uniform softmax at init
    sup, unsup, val = get_loader(cfg, seed=2)
    batches = list(zip(iter(sup), iter(unsup)))
    val_batches = list(iter(val))
    assert len(batches) == 3 and sum(b[0].shape[0] for b in val_batches) == 50
    torch.manual_seed(0)
    model, teacher = ModelBuilder(copy.deepcopy(cfg["net"])), ModelBuilder(copy.deepcopy(cfg["net"]))
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    teacher.load_state_dict(sd)
    from oracle.parity_dropout import KeyedMasks, tag_model
    from u2pl_amd import nn as Kn
    tag_model(model, "student"), tag_model(teacher, "teacher")
    assert all(m.p == 0.1 for m in model.modules() if isinstance(m, nn.Dropout2d))
    model, teacher = model.to(DEV), teacher.to(DEV)
    Kn.DROPOUT_HOOK = KeyedMasks(77).hook
    tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=len(batches))
    ref = CpuStepRef(arch="resnet50", num_classes=19, aux=True, epochs=1, steps_per_epoch=len(batches),
                     ohem=(0.7, cfg["criterion"]["kwargs"]["min_kept"]), p_drop=0.1,
                     contra=copy.deepcopy(cfg["trainer"]["contrastive"]), state_dict={k: v.clone() for k, v in sd.items()},
                     dropout_masks=KeyedMasks(77))
    for step, ((il, ll), (iu, _)) in enumerate(batches):
        g1, g2 = torch.Generator().manual_seed(90 + step), torch.Generator().manual_seed(90 + step)
        np.random.seed(40 + step)
        ref.step(il, ll, iu, 0, randint=lambda hi, n, g=g1: torch.randint(hi, size=(n,), generator=g).numpy())
        np.random.seed(40 + step)
        tr.train_step(il.to(DEV), ll.to(DEV), iu.to(DEV), 0, randint=lambda hi, n, g=g2: torch.randint(hi, size=(n,), generator=g))
    Kn.DROPOUT_HOOK = None
    miou_ref, iou_ref = validate_ref(ref.teacher, val_batches, 19)
    miou_gpu, iou_gpu = validate(teacher, val_batches, cfg, torch.device(DEV))
    print("mIoU cpu-reference %.4f  hip %.4f" % (miou_ref * 100, miou_gpu * 100))
    assert abs(miou_gpu - miou_ref) * 100 <= 0.3
    assert np.abs(iou_gpu - iou_ref).max() * 100 <= 1.0
