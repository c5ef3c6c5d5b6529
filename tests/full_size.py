"""Shared helpers of the BASELINE-size parity tests (tests/golden/train_full_*.npz).

The inputs of those goldens are never stored (28 MB per step at 769^2): both the generator
(oracle/gen_golden.py:survey_step_inputs, run next to the reference) and the tests rebuild them from the seed
with torch's CPU generator (bit-reproducible across machines for one torch version; the fixture also stores
the labels' pseudo-label side so a drift would be caught as a label mismatch, not as a silent loss error)."""
import numpy as np
import torch


def survey_step_inputs(seed, B, S, C, n):
    """SURVEY 8(d) synthetic step inputs: images N(0,1); labels randint(0,C) on an (S//16+1)^2 grid,
    nearest-up-sampled, first 8 rows 255."""
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        il, iu = torch.randn(B, 3, S, S, generator=gen), torch.randn(B, 3, S, S, generator=gen)
        gsz = S // 16 + 1
        coarse = torch.randint(0, C, (B, gsz, gsz), generator=gen)
        iy = (torch.arange(S) * gsz // S).clamp(max=gsz - 1)
        ll = coarse[:, iy][:, :, iy].contiguous()
        ll[:, :8] = 255
        out.append((il, ll, iu))
    return out


def unpack(g, key, shape):
    n = int(np.prod(shape))
    return np.unpackbits(g[key])[:n].reshape(shape).astype(bool)


def golden_step(g, i, S, B, s):
    """-> dict of the reference's masks at semi-supervised step i of a train_full_* fixture."""
    d = dict(label_u=g[f"s{i}_label_u"].astype(np.int64), dropped=unpack(g, f"s{i}_dropped", (B, S, S)))
    tgt = d["label_u"].copy()
    tgt[d["dropped"]] = 255
    d["target_u"] = tgt
    if f"s{i}_low" in g.files:
        d["low"] = unpack(g, f"s{i}_low", (2 * B, 1, s, s))
        d["high"] = unpack(g, f"s{i}_high", (2 * B, 1, s, s))
        d["lbits"] = g[f"s{i}_lbits"]
        d["bank_len"] = g[f"s{i}_bank_len"]
    return d


def cfg_for(tag, configs):
    """product config matching oracle/gen_golden.py:FULL_SIZE[tag] / _train_cfg."""
    voc, arch, S, B, C, steps, epochs_run = FULL[tag]
    big = S > 200
    if voc:
        cfg = configs.pascal_semi(arch=arch, crop=S, batch_size=B, sync_bn=False, epochs=200 if big else 20)
    else:
        cfg = configs.cityscapes_semi(arch=arch, crop=S, batch_size=B, sync_bn=False, epochs=200 if big else 20)
        cfg["criterion"]["kwargs"]["min_kept"] = 100000 if big else 4000
    return cfg


FULL = {   # mirrors oracle/gen_golden.py:FULL_SIZE (voc, arch, S, B, C, steps per epoch, epochs)
    "city769": (False, "resnet101", 769, 2, 19, 1, [0]),
    "voc513": (True, "resnet101", 513, 4, 21, 1, [0, 1]),
    "city97": (False, "resnet50", 97, 2, 19, 2, [0]),
}


def port_for_full(tag, g):
    """CpuStepRef configured like oracle/gen_golden.py:gen_train_full(tag): reference-identical seeded init,
    classifier last layer x sharpen, dropout p = 0.1 with the keyed keep-masks."""
    import torch
    from oracle.parity_dropout import KeyedMasks
    from oracle.step_ref import CpuStepRef
    from u2pl_amd import configs
    from u2pl_amd.models.model_helper import ModelBuilder

    voc, arch, S, B, C, steps, epochs_run = FULL[tag]
    cfg = cfg_for(tag, configs)
    torch.manual_seed(int(g["seeds"][0]))
    sd = {k: v.detach().clone() for k, v in ModelBuilder(cfg["net"]).state_dict().items()}
    sd["decoder.classifier.8.weight"] = sd["decoder.classifier.8.weight"] * float(g["sharpen"])
    ok = cfg["trainer"]["optimizer"]["kwargs"]
    ohem = None if voc else (0.7, cfg["criterion"]["kwargs"]["min_kept"])
    ref = CpuStepRef(arch=arch, num_classes=C, aux=not voc, epochs=cfg["trainer"]["epochs"], steps_per_epoch=steps,
                     lr=ok["lr"], weight_decay=ok["weight_decay"], lr_times=10 if voc else 1,
                     sup_only_epoch=1 if voc else 0, ohem=ohem, p_drop=0.1, contra=dict(cfg["trainer"]["contrastive"]),
                     state_dict=sd, dropout_masks=KeyedMasks(int(g["seeds"][4])))
    return ref, cfg, sd
