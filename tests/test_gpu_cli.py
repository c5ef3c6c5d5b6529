"""-m gpu: the train_semi.py / train_sup.py command lines end to end on a synthetic
Cityscapes-layout dataset: reference YAML surface, loaders, training steps, device validate(),
checkpoint wire format ('module.' prefix, teacher_state) and auto-resume."""
import os
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _run(script, cfg):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--config", cfg, "--seed", "2"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout + r.stderr


def test_train_semi_cli_checkpoint_and_resume(tmp_path):
    import make_synth_dataset as M

    d, s = M.make_cityscapes(str(tmp_path), H=110, W=150)
    cfgp = M.write_city_config(str(tmp_path), d, s, crop=97, epochs=1)
    out = _run("train_semi.py", cfgp)
    assert "mIoU" in out
    ck = torch.load(os.path.join(os.path.dirname(cfgp), "checkpoints", "ckpt.pth"), map_location="cpu")
    assert set(ck) >= {"epoch", "model_state", "teacher_state", "best_miou"} and ck["epoch"] == 1
    assert all(k.startswith("module.") for k in ck["model_state"])
    assert "module.encoder.layer3.5.conv2.weight" in ck["model_state"] and ck["model_state"]["module.encoder.conv1.0.weight"].shape == (64, 3, 3, 3)
    cfg = yaml.load(open(cfgp), Loader=yaml.Loader)
    cfg["trainer"]["epochs"] = 2
    yaml.safe_dump(cfg, open(cfgp, "w"))
    _run("train_semi.py", cfgp)                       # auto-resume from epoch 1
    assert torch.load(os.path.join(os.path.dirname(cfgp), "checkpoints", "ckpt.pth"), map_location="cpu")["epoch"] == 2


def test_train_sup_cli(tmp_path):
    import make_synth_dataset as M

    d, s = M.make_cityscapes(str(tmp_path), H=110, W=150)
    cfgp = M.write_city_config(str(tmp_path), d, s, crop=97, epochs=1)
    cfg = yaml.load(open(cfgp), Loader=yaml.Loader)
    cfg["dataset"]["type"] = "cityscapes"
    cfg["dataset"]["n_sup"] = 4
    cfg["net"]["decoder"]["kwargs"]["rep_head"] = False
    for k in ("unsupervised", "contrastive"):
        cfg["trainer"].pop(k)
    yaml.safe_dump(cfg, open(cfgp, "w"))
    out = _run("train_sup.py", cfgp)
    assert "mIoU" in out


def test_train_semi_cli_with_device_side_data_pipeline(tmp_path):
    """dataset.device_aug: uint8 batches + host-drawn geometry, transform chain fused on the GPU"""
    import make_synth_dataset as M

    d, s = M.make_cityscapes(str(tmp_path), H=110, W=150)
    cfgp = M.write_city_config(str(tmp_path), d, s, crop=97, epochs=1)
    cfg = yaml.load(open(cfgp), Loader=yaml.Loader)
    cfg["dataset"]["device_aug"] = True
    yaml.safe_dump(cfg, open(cfgp, "w"))
    out = _run("train_semi.py", cfgp)
    assert "mIoU" in out
