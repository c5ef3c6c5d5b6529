"""Shared helpers for model tests (no reference import)."""
import copy

import torch


def net_cfg(arch, num_classes, aux, sync_bn=False):
    net = dict(
        num_classes=num_classes, sync_bn=sync_bn, ema_decay=0.99,
        encoder=dict(type=f"u2pl.models.resnet.{arch}",
                     kwargs=dict(multi_grid=True, zero_init_residual=True, fpn=True,
                                 replace_stride_with_dilation=[False, True, True], pretrained=False)),
        decoder=dict(type="u2pl.models.decoder.dec_deeplabv3_plus", kwargs=dict(inner_planes=256, dilations=[12, 24, 36])),
    )
    if aux:
        net["aux_loss"] = dict(aux_plane=1024, loss_weight=0.4)
    return copy.deepcopy(net)


def formula_state_dict(model, seed=1234):
    """Same closed-form weights as oracle/gen_golden.py:formula_state_dict (kept in
    sync by tests/test_oracle_golden.py::test_formula_state_dict_in_sync)."""
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        if v.dtype == torch.long:
            out[k] = v.clone()
            continue
        h = (sum((i + 1) * ord(c) for i, c in enumerate(k)) * 2654435761 + seed) % (2 ** 31)
        n = v.numel()
        idx = torch.arange(n, dtype=torch.float64)
        u = torch.frac(torch.sin(idx * 12.9898 + (h % 10007) * 0.618) * 43758.5453).abs()
        if k.endswith("running_var"):
            val = 0.5 + u
        elif k.endswith("running_mean"):
            val = (u - 0.5) * 0.2
        elif k.endswith("weight") and v.dim() == 1:
            val = 0.5 + u
        elif k.endswith("bias"):
            val = (u - 0.5) * 0.2
        else:
            fan_in = v[0].numel()
            val = (u - 0.5) * 2 * (3.0 / fan_in) ** 0.5 * 1.4
        out[k] = val.reshape(v.shape).to(v.dtype)
    return out
