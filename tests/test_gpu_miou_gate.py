"""-m gpu: north_star acceptance gate -- "mIoU on a fixed 50-image val subset within +-0.3 of the CPU reference after
1 epoch".

The CPU side is the REFERENCE ITSELF: tests/golden/miou_gate.npz was written in the build container by the reference's
own train() (train_semi.py:234-592; one epoch = 40 steps, R101-DeepLabv3+, 193x193, 2 labeled + 2 unlabeled images,
OHEM + aux, CutMix, contrastive bank, dropout ON with keyed keep-masks) followed by its own validate()
(train_semi.py:595-654) of the EMA teacher on 50 validation images (oracle/gen_golden.py:gen_miou_gate; inputs are
regenerated from a seed by tests/miou_gate.py on both sides, pinned by a digest).  The task is learnable: the fixture
also holds the mIoU of the initial weights, and the gate asserts that the epoch moved it by >= 2 points on BOTH sides,
so agreement is not the trivial agreement of two untrained networks.  The fixture's noise-floor run (the reference's
epoch again, from weights perturbed by one fp32 rounding) says what two fp32 implementations may differ by after 40
chaotic steps; the HIP path (production default: Winograd F(4x4) + split-fp32 products) is held to the north_star's
+-0.3 points."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_miou_after_one_epoch_matches_the_reference():
    import miou_gate as MG
    from oracle.parity_dropout import KeyedMasks, tag_model
    from u2pl_amd import configs
    from u2pl_amd import nn as Kn
    from u2pl_amd.engine import validate
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion

    g = golden("miou_gate")
    G = MG.GATE
    init_seed, data_seed, np_seed, torch_seed, dropout_seed = (int(x) for x in g["seeds"])
    steps = int(g["steps"])
    data = MG.gate_data(data_seed, steps, G["B"], G["S"])
    val = MG.gate_val(data_seed + 1, G["n_val"], G["S"])
    assert np.array_equal(MG.data_digest(data, val), g["digest"]), "regenerated inputs differ from the fixture's"
    cfg = configs.cityscapes_semi(arch=G["arch"], crop=G["S"], batch_size=G["B"], sync_bn=False, epochs=G["epochs"])
    cfg["criterion"]["kwargs"]["min_kept"] = G["min_kept"]
    assert cfg["trainer"]["contrastive"]["current_class_threshold"] == G["class_thr"]
    torch.manual_seed(init_seed)
    model = ModelBuilder(copy.deepcopy(cfg["net"]))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    teacher = ModelBuilder(copy.deepcopy(cfg["net"]))
    teacher.load_state_dict(sd)
    assert all(m.p == 0.1 for m in model.modules() if isinstance(m, nn.Dropout2d))
    tag_model(model, "student"), tag_model(teacher, "teacher")
    model, teacher = model.to(DEV), teacher.to(DEV)
    miou_init, _ = validate(teacher, val, cfg, torch.device(DEV))
    Kn.DROPOUT_HOOK = KeyedMasks(dropout_seed).hook
    try:
        tr = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=steps)
        np.random.seed(np_seed)
        torch.manual_seed(torch_seed)        # compute_contra_memobank_loss draws from the global CPU generator
        losses = []
        for il, ll, iu in data:
            losses.append(tr.train_step(il.to(DEV), ll.to(DEV), iu.to(DEV), 0).cpu().numpy())
    finally:
        Kn.DROPOUT_HOOK = None
    miou_t, iou_t = validate(teacher, val, cfg, torch.device(DEV))
    miou_s, _ = validate(model, val, cfg, torch.device(DEV))
    ref_t, ref_s, ref_init = float(g["miou_teacher"]), float(g["miou_student"]), float(g["miou_init"])
    noise = abs(float(g["noise_miou_teacher"]) - ref_t) if "noise_miou_teacher" in g.files else float("nan")
    losses = np.asarray(losses)
    rep = dict(miou_init=(100 * miou_init, 100 * ref_init), miou_teacher=(100 * miou_t, 100 * ref_t),
               miou_student=(100 * miou_s, 100 * ref_s), reference_noise_floor_points=100 * noise,
               first_losses=(losses[0].tolist(), g["meters"][0, 2:5].tolist()),
               last_losses=(losses[-1].tolist(), g["meters"][-1, 2:5].tolist()),
               iou_teacher_max_diff_points=float(100 * np.abs(iou_t - g["iou_teacher"]).max()))
    print("MIOU_GATE (hip, reference)", rep)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    import json
    json.dump(rep, open(os.path.join(out, "miou_gate.json"), "w"), indent=1)
    # step 0 starts from identical weights: the north_star loss tolerance
    for a, b in zip(losses[0], g["meters"][0, 2:5]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), rep
    # the epoch trained: both sides moved the validation mIoU by at least 2 points ...
    assert 100 * (ref_t - ref_init) >= 2.0 and 100 * (miou_t - miou_init) >= 2.0, rep
    assert abs(miou_init - ref_init) * 100 <= 0.05, rep
    # ... and the gate itself
    assert abs(miou_t - ref_t) * 100 <= 0.3, rep
    # per class: the reference's own noise-floor run moves single classes by up to 1.6 points (class 0: 85.3 vs 86.9) while the
    # mean moves by 0.02 -- a loose per-class bound, the mean is the gate
    assert np.abs(iou_t - g["iou_teacher"]).max() * 100 <= 5.0, rep
