"""-m gpu parity of the fused device-side data pipeline (u2pl_augment_u8_f32) against the per-sample CPU
transform chain (u2pl_amd.dataset.builder.Pipeline == reference augmentation.py composition) under the same
python-`random` seed: identical geometry draws, labels bit-exact, normalised pixels within 2e-6 (1-2 ulp)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], rand_resize=[0.5, 2.0], flip=True,
           crop=dict(type="rand", size=[97, 113]))


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_device_pipeline_equals_cpu_pipeline(seed):
    from PIL import Image
    from u2pl_amd.dataset.builder import Pipeline
    from u2pl_amd.dataset.device_aug import AugmentPlan, augment_batch

    rng = np.random.default_rng(seed)
    H, W = 96 + 8 * (seed % 3), 150
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    lab = rng.integers(0, 19, (H, W), dtype=np.uint8)
    lab[:5] = 255
    random.seed(100 + seed)
    ref_img, ref_lab = Pipeline(CFG)(Image.fromarray(img), Image.fromarray(lab))
    random.seed(100 + seed)
    plan = AugmentPlan(CFG)
    params = torch.from_numpy(plan.draw(H, W))[None]
    state_after = random.random()
    out, ol = augment_batch(plan, torch.from_numpy(img)[None].to(DEV), torch.from_numpy(lab)[None].to(DEV), params)
    random.seed(100 + seed)
    Pipeline(CFG)(Image.fromarray(img), Image.fromarray(lab))
    assert random.random() == state_after          # both consume the RNG stream identically
    assert torch.equal(ol[0].cpu(), ref_lab), (params, (ol[0].cpu() != ref_lab).sum())
    err = (out[0].cpu() - ref_img).abs().max().item()
    print('max abs err', err)
    assert err < 2e-6, (params, err)


def test_device_pipeline_batch_and_center_crop_padding():
    """batch of 3, image smaller than the crop (zero padding of image AND label, augmentation.py:241-245)"""
    from PIL import Image
    from u2pl_amd.dataset.builder import Pipeline
    from u2pl_amd.dataset.device_aug import AugmentPlan, augment_batch
    cfg = dict(CFG, rand_resize=False, flip=False, crop=dict(type="center", size=[80, 120]))
    rng = np.random.default_rng(9)
    imgs = rng.integers(0, 256, (3, 64, 100, 3), dtype=np.uint8)
    labs = rng.integers(1, 19, (3, 64, 100), dtype=np.uint8)
    plan = AugmentPlan(cfg)
    params = torch.stack([torch.from_numpy(plan.draw(64, 100)) for _ in range(3)])
    out, ol = augment_batch(plan, torch.from_numpy(imgs).to(DEV), torch.from_numpy(labs).to(DEV), params)
    for b in range(3):
        ri, rl = Pipeline(cfg)(Image.fromarray(imgs[b]), Image.fromarray(labs[b]))
        assert torch.equal(ol[b].cpu(), rl)
        assert (out[b].cpu() - ri).abs().max().item() < 2e-5
        assert int((rl == 0).sum()) > 0            # the padded border carries label 0, not ignore
