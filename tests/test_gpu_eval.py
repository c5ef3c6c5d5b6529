"""-m gpu parity of the sliding-window evaluation (u2pl_amd/evaluate.py, reference eval.py:158-320) against
goldens produced by the reference's own scale_crop_process, and of evaluate() against the CPU restatement."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden
from model_utils import formula_state_dict, net_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model():
    from u2pl_amd.models.model_helper import ModelBuilder
    m = ModelBuilder(net_cfg("resnet50", 19, True))
    m.load_state_dict(formula_state_dict(m))
    return m.to(DEV).eval()


@pytest.mark.parametrize("tag", ["70x100", "50x90"])
def test_sliding_window_matches_reference_golden(tag):
    from u2pl_amd import evaluate as E
    g = golden("evalwin_" + tag)
    x = torch.from_numpy(g["x"]).to(DEV)
    crop = int(g["crop"])
    H, W = x.shape[2:]
    out = E.scale_crop_process(_model(), x, 19, crop, crop, H, W).cpu().double()
    ref32, ref64 = torch.from_numpy(g["out"]).double(), torch.from_numpy(g["out64"]).double()
    e_ref = (ref32 - ref64).abs().max().item()
    err = (out - ref64).abs()
    scale = ref64.abs().max().item()
    print(f"|hip-f64| {err.max().item():.3e}  |ref32-f64| {e_ref:.3e}  scale {scale:.3e}")
    # formula weights are ill-conditioned on purpose; Winograd F(4x4) layers may add up to ~8x the fp32 noise
    assert (err > 32.0 * e_ref + 1e-6 * scale).double().mean().item() <= 0.01
    assert (out.argmax(0) == ref32.argmax(0)).double().mean().item() > 0.97


def test_evaluate_miou_matches_cpu_restatement():
    from oracle.model_ref import RefNet
    from oracle.step_ref import sliding_window_ref
    from u2pl_amd import evaluate as E
    torch.manual_seed(3)
    from u2pl_amd.models.model_helper import ModelBuilder
    m = ModelBuilder(net_cfg("resnet50", 19, True))          # reference-identical seeded initialisation
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    ref = RefNet("resnet50", 19, True, p_drop=0.0)
    ref.load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    samples = []
    for (h, w) in [(70, 100), (60, 66)]:
        img = torch.randn(3, h, w, generator=g)
        lab = torch.randint(0, 19, (h, w), generator=g).numpy().astype(np.uint8)
        lab[:3] = 255
        samples.append((img, lab))
    miou, iou = E.evaluate(m, samples, 19, base_size=100, crop=(65, 65), scales=(1.0,), use_crop=True)
    # CPU restatement of validate_city with the same scaling rule
    inter, union = np.zeros(19), np.zeros(19)
    import torch.nn.functional as F
    agree = []
    for img, lab in samples:
        h, w = img.shape[1:]
        long_size = 100
        new_h = new_w = long_size
        if h > w:
            new_w = round(long_size / float(h) * w)
        else:
            new_h = round(long_size / float(w) * h)
        xs = F.interpolate(img.unsqueeze(0), size=(new_h, new_w), mode="bilinear", align_corners=True)
        pr = sliding_window_ref(ref, xs, 19, 65, 65, h, w).argmax(0).numpy()
        out = np.where(lab == 255, 255, pr)
        hit = out[out == lab]
        ai = np.bincount(hit[hit != 255], minlength=19)[:19]
        ao = np.bincount(out[out != 255], minlength=19)[:19]
        at = np.bincount(lab[lab != 255], minlength=19)[:19]
        inter += ai
        union += ao + at - ai
    ref_iou = inter / (union + 1e-10)
    print("mIoU hip", miou, "cpu", float(ref_iou.mean()))
    assert abs(miou - float(ref_iou.mean())) < 3e-3


def test_confusion_hist_kernel_vs_reference_intersection_and_union_golden():
    """u2pl_confusion_hist_f32 (argmax + the three class histograms of validate(), train_semi.py:620-641) against
    tests/golden/miou_hist.npz, written by the reference's utils.intersectionAndUnion (utils.py:568-580): ignored
    pixels (random + a band), a class that never occurs in the ground truth; logits are built so that their arg-max is
    the fixture's prediction map, once with a clear margin and once with the winner only one ulp ahead."""
    from u2pl_amd._lib import call
    g = golden("miou_hist")
    out, tgt = g["out"].astype(np.int64), g["tgt"].astype(np.int64)
    N, H, W = out.shape
    C = 19
    rng = np.random.default_rng(3)
    for margin in (1.0, None):
        logits = rng.standard_normal((N, C, H, W)).astype(np.float32)
        top = logits.max(1)
        win = top + margin if margin is not None else np.nextafter(top, np.float32(np.inf), dtype=np.float32)
        np.put_along_axis(logits, out[:, None], win[:, None], axis=1)
        assert np.array_equal(logits.argmax(1), out)
        hist = torch.zeros(3 * C, dtype=torch.int64, device=DEV)
        for rep in range(2):      # the kernel ACCUMULATES (validate() sums over batches): two calls = twice the counts
            call("u2pl_confusion_hist_f32", torch.from_numpy(logits).to(DEV), torch.from_numpy(tgt).to(DEV), 255, N, C, H, W, hist)
        h = hist.cpu().numpy().reshape(3, C).astype(np.float64) / 2
        inter, union, target = h[0], h[1] + h[2] - h[0], h[2]
        assert np.array_equal(inter, g["inter"]) and np.array_equal(union, g["union"]) and np.array_equal(target, g["target"])
