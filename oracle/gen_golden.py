"""TEST INFRASTRUCTURE ONLY -- generate golden vectors from the REFERENCE itself.

Runs only in the build container (needs /root/reference, imported read-only via
``oracle/ref_shim.py``); writes small ``tests/golden/*.npz`` fixtures that are
committed and travel to the GPU box.  Re-run:  ``python oracle/gen_golden.py``.

Every fixture stores its inputs AND the reference's outputs, so tests never
depend on RNG reproducibility across machines.  What calls what:

  unsup_*        u2pl.utils.loss_helper.compute_unsupervised_loss   (loss_helper.py:30-48)
  ohem_*         u2pl.utils.loss_helper.CriterionOhem               (loss_helper.py:323-360,451-531)
  relsplit_*     the inline block train_semi.py:397-465 executed with the same
                 torch / numpy calls and the reference's label_onehot (utils.py:50-59)
  contra_*       u2pl.utils.loss_helper.compute_contra_memobank_loss (loss_helper.py:51-235)
                 incl. memory-bank state before/after and d loss / d rep
  bank_seq       u2pl.utils.utils.dequeue_and_enqueue sequence        (utils.py:27-47)
  cutmix         u2pl.dataset.augmentation.generate_unsup_data        (augmentation.py:471-541)
  pseudo_*       train_semi.py:320-324 (bilinear up, softmax, max)
  sgd_ema        torch.optim.SGD + LRScheduler (lr_helper.py:78-113) + EMA (train_semi.py:531-548)
  model_*        u2pl.models.model_helper.ModelBuilder fwd/bwd on tiny inputs with
                 formula-initialised weights (see formula_state_dict)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CONTRA_CFG = dict(
    negative_high_entropy=True, low_rank=3, high_rank=20, current_class_threshold=0.3,
    current_class_negative_threshold=1, unsupervised_entropy_ignore=80,
    low_entropy_threshold=20, num_negatives=50, num_queries=256, temperature=0.5,
)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def block_labels(B, S, C, gen, ignore_rows=4, cell=8):
    g = (S + cell - 1) // cell
    coarse = torch.randint(0, C, (B, 1, g, g), generator=gen).float()
    lab = F.interpolate(coarse, size=(g * cell, g * cell), mode="nearest")[:, 0, :S, :S].long()
    lab[:, :ignore_rows] = 255
    return lab.contiguous()


def assert_no_ties(prob, k=4):
    top = torch.sort(prob, 1, True)[0][:, : k + 1]
    assert (top[:, :-1] - top[:, 1:]).min() > 0, "tie among top probabilities; change seed"


# ----------------------------------------------------------------------------
def gen_unsup(ns, seed, S, s, C, percent, tag):
    gen = torch.Generator().manual_seed(seed)
    low_t = torch.randn(2, C, s, s, generator=gen) * 3
    low_s = torch.randn(2, C, s, s, generator=gen) * 2
    pred_teacher = F.interpolate(low_t, (S, S), mode="bilinear", align_corners=True)
    predict = F.interpolate(low_s, (S, S), mode="bilinear", align_corners=True).requires_grad_(True)
    target = block_labels(2, S, C, gen)
    tgt = target.clone()
    loss = ns.loss_helper.compute_unsupervised_loss(predict, tgt, percent, pred_teacher)
    loss.backward()
    prob = torch.softmax(pred_teacher, 1)
    entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
    g = predict.grad
    save(f"unsup_{tag}", low_teacher=low_t, low_student=low_s, size=np.int64(S),
         target=target.to(torch.uint8), percent=np.float64(percent), loss=loss,
         new_target=tgt.to(torch.uint8), grad_sub=g[:, :, ::5, ::5], grad_sum_c=g.double().sum((0, 2, 3)),
         grad_abs_sum=g.double().abs().sum(), entropy=entropy)


def gen_ohem(ns, seed, S, s, C, min_kept, tag, ignore_frac_rows=4):
    gen = torch.Generator().manual_seed(seed)
    low = torch.randn(2, C, s, s, generator=gen) * 2
    low_aux = torch.randn(2, C, s, s, generator=gen) * 2
    target = block_labels(2, S, C, gen, ignore_rows=ignore_frac_rows)
    # make the prediction partly agree with the target so p[target] spreads over (0,1)
    onehot = F.one_hot(torch.where(target == 255, 0, target), C).permute(0, 3, 1, 2).float()
    main = (F.interpolate(low, (S, S), mode="bilinear", align_corners=True) + 2.5 * onehot).requires_grad_(True)
    aux = (F.interpolate(low_aux, (S, S), mode="bilinear", align_corners=True) + 1.0 * onehot).requires_grad_(True)
    crit = ns.loss_helper.CriterionOhem(0.4, thresh=0.7, min_kept=min_kept, ignore_index=255)
    loss = crit([main, aux], target.clone())
    loss.backward()
    crit1 = ns.loss_helper.OhemCrossEntropy2dTensor(255, 0.7, min_kept)
    loss_main = crit1(main.detach(), target.clone())
    gm, ga = main.grad, aux.grad
    save(f"ohem_{tag}", low=low, low_aux=low_aux, size=np.int64(S), target=target.to(torch.uint8),
         min_kept=np.int64(min_kept), thresh=np.float64(0.7), aux_weight=np.float64(0.4), loss=loss,
         loss_main=loss_main, grad_main_sub=gm[:, :, ::5, ::5], grad_aux_sub=ga[:, :, ::5, ::5],
         grad_main_abs_sum=gm.double().abs().sum(), grad_aux_abs_sum=ga.double().abs().sum(),
         n_kept_main=(gm.abs().sum(1) > 0).sum())


def relsplit_reference(ns, pred_u_large_teacher, label_u_aug, label_l, alpha_t, out_hw, C):
    """train_semi.py:397-465 with the same calls (torch + numpy + reference label_onehot)."""
    with torch.no_grad():
        prob = torch.softmax(pred_u_large_teacher, dim=1)
        entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
        low_thresh = np.percentile(entropy[label_u_aug != 255].cpu().numpy().flatten(), alpha_t)
        low_entropy_mask = entropy.le(low_thresh).float() * (label_u_aug != 255).bool()
        high_thresh = np.percentile(entropy[label_u_aug != 255].cpu().numpy().flatten(), 100 - alpha_t)
        high_entropy_mask = entropy.ge(high_thresh).float() * (label_u_aug != 255).bool()
        low_mask_all = torch.cat(((label_l.unsqueeze(1) != 255).float(), low_entropy_mask.unsqueeze(1)))
        low_mask_all = F.interpolate(low_mask_all, size=out_hw, mode="nearest")
        high_mask_all = torch.cat(((label_l.unsqueeze(1) != 255).float(), high_entropy_mask.unsqueeze(1)))
        high_mask_all = F.interpolate(high_mask_all, size=out_hw, mode="nearest")
        label_l_small = F.interpolate(ns.utils.label_onehot(label_l, C), size=out_hw, mode="nearest")
        label_u_small = F.interpolate(ns.utils.label_onehot(label_u_aug, C), size=out_hw, mode="nearest")
    return dict(entropy=entropy, low_thresh=np.float32(low_thresh), high_thresh=np.float32(high_thresh),
                low_mask_all=low_mask_all, high_mask_all=high_mask_all,
                label_l_small=label_l_small.long(), label_u_small=label_u_small.long())


def make_step_inputs(seed, B, S, s, C, D=256, cutout=False):
    """Synthetic tensors shaped like the ones train_semi.py hands to the loss block."""
    gen = torch.Generator().manual_seed(seed)
    low_t_train = torch.randn(2 * B, C, s, s, generator=gen) * 3      # teacher (train mode) logits, all images
    low_t_eval = low_t_train[B:] + 1.5 * torch.randn(B, C, s, s, generator=gen)  # teacher eval logits (unlabeled)
    label_l = block_labels(B, S, C, gen)
    conf, label_u = torch.max(torch.softmax(
        F.interpolate(low_t_eval, (S, S), mode="bilinear", align_corners=True), 1), 1)
    if cutout:
        label_u = label_u.clone()
        label_u[:, S // 3: S // 2, S // 4: S // 2] = 255
    rep = torch.round(torch.randn(2 * B, D, s, s, generator=gen) * 64) / 64   # coarse grid: fixture compresses
    rep_t = torch.round(torch.randn(2 * B, D, s, s, generator=gen) * 64) / 64
    prob_all = torch.softmax(low_t_train, 1)
    assert_no_ties(prob_all)
    pred_u_large_teacher = F.interpolate(low_t_train[B:], (S, S), mode="bilinear", align_corners=True)
    return dict(low_t_train=low_t_train, low_t_eval=low_t_eval, label_l=label_l, label_u_aug=label_u,
                conf_u=conf, rep=rep, rep_teacher=rep_t, prob_all=prob_all,
                pred_u_large_teacher=pred_u_large_teacher)


def gen_relsplit(ns, seed, B, S, s, C, alpha_t, tag, cutout=False):
    inp = make_step_inputs(seed, B, S, s, C, D=8, cutout=cutout)
    out = relsplit_reference(ns, inp["pred_u_large_teacher"], inp["label_u_aug"], inp["label_l"], alpha_t, (s, s), C)
    out = {k: (v.to(torch.uint8) if isinstance(v, torch.Tensor) and k != "entropy" else v) for k, v in out.items()}
    save(f"relsplit_{tag}", low_t_train=inp["low_t_train"], label_l=inp["label_l"].to(torch.uint8),
         label_u_aug=inp["label_u_aug"].to(torch.uint8), size=np.int64(S), alpha_t=np.float64(alpha_t), **out)


def formula_bank(c, n, D):
    """closed-form, machine-independent bank rows (tests rebuild them; not stored)."""
    r = torch.arange(n, dtype=torch.int64)[:, None]
    d = torch.arange(D, dtype=torch.int64)[None, :]
    return (((r * 131 + d * 31 + c * 17 + (r * d) % 7) % 1000).float() / 500.0 - 1.0)


def gen_contra(ns, seed, B, S, s, C, alpha_t, tag, prefill=0, steps=1, queue_size=3000, D=256, temperature=None,
               queue_size0=None, prefill0=None):
    """steps>1: call the reference repeatedly on fresh inputs (bank carries over).
    temperature: overrides CONTRA_CFG's 0.5 (stored in the fixture); queue_size0 / prefill0: capacity / pre-fill of class 0
    (the real run uses 30000 and 50000 for class 0, train_semi.py:161-169; with a pre-fill just below capacity the
    enqueue of step 0 wraps the ring and step 1 samples from a wrapped ring)."""
    torch.manual_seed(seed + 1000)  # global generator used by torch.randint inside the reference
    cfg = dict(CONTRA_CFG)
    if temperature is not None:
        cfg["temperature"] = temperature
    fill = [prefill + 3 * i for i in range(C)]
    if prefill0 is not None:     # "just below capacity" fixtures: every class `prefill` rows, class 0 `prefill0`
        fill = [prefill0] + [prefill] * (C - 1)
    memobank = [[torch.zeros(0, D)] for _ in range(C)]
    if prefill:
        memobank = [[formula_bank(i, fill[i], D)] for i in range(C)]
    queue_ptrlis = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
    queue_size_l = [queue_size] * C
    queue_size_l[0] = queue_size + 500 if queue_size0 is None else queue_size0
    fx = dict(num_steps=np.int64(steps), queue_size=np.array(queue_size_l), alpha_t=np.float64(alpha_t),
              prefill=np.int64(prefill), D=np.int64(D), temperature=np.float64(cfg["temperature"]),
              fill=np.array(fill if prefill else [0] * C))
    for st in range(steps):
        inp = make_step_inputs(seed + 31 * st, B, S, s, C, D=D)
        rs = relsplit_reference(ns, inp["pred_u_large_teacher"], inp["label_u_aug"], inp["label_l"], alpha_t, (s, s), C)
        rep = inp["rep"].clone().requires_grad_(True)
        rng_state = torch.get_rng_state()
        new_keys, loss = ns.loss_helper.compute_contra_memobank_loss(
            rep, rs["label_l_small"], rs["label_u_small"], inp["prob_all"][:B], inp["prob_all"][B:],
            rs["low_mask_all"], rs["high_mask_all"], cfg, memobank, queue_ptrlis, queue_size_l,
            inp["rep_teacher"])
        loss.backward()
        p = f"s{st}_"
        fx.update({
            p + "rep": inp["rep"], p + "rep_teacher": inp["rep_teacher"], p + "prob_all": inp["prob_all"],
            p + "label_l": inp["label_l"].to(torch.uint8), p + "label_u_aug": inp["label_u_aug"].to(torch.uint8),
            p + "low_t_train": inp["low_t_train"],
            p + "low_mask_all": rs["low_mask_all"].to(torch.uint8), p + "high_mask_all": rs["high_mask_all"].to(torch.uint8),
            p + "label_l_small": rs["label_l_small"].to(torch.uint8), p + "label_u_small": rs["label_u_small"].to(torch.uint8),
            p + "loss": loss, p + "grad_rep": rep.grad, p + "new_keys": np.array(new_keys),
            p + "rng_state": rng_state,
            p + "bank_len": np.array([memobank[c][0].shape[0] for c in range(C)]),
            p + "queue_ptr": np.array([int(queue_ptrlis[c][0]) for c in range(C)]),
        })
        print(tag, "step", st, "loss", float(loss), "new_keys", new_keys)
    for c in range(C):
        b = memobank[c][0]
        fx[f"bankF_{c}_sum"] = b.double().sum(0)
        fx[f"bankF_{c}_head"] = b[:4]
        fx[f"bankF_{c}_tail"] = b[-4:]
    save(f"contra_{tag}", **fx)


def gen_bank_seq(ns, seed):
    gen = torch.Generator().manual_seed(seed)
    queue = [torch.zeros(0, 16)]
    ptr = torch.zeros(1, dtype=torch.long)
    sizes = [5, 0, 7, 30, 3, 64, 1]
    fx = dict(sizes=np.array(sizes), queue_size=np.int64(40))
    for i, n in enumerate(sizes):
        keys = torch.randn(n, 16, generator=gen)
        ret = ns.utils.dequeue_and_enqueue(keys, queue, ptr, 40)
        fx[f"keys{i}"] = keys
        fx[f"queue{i}"] = queue[0].clone()
        fx[f"ptr{i}"] = np.int64(int(ptr[0]))
        fx[f"ret{i}"] = np.int64(ret)
    save("bank_seq", **fx)


def gen_cutmix(ns, seed, B, S):
    gen = torch.Generator().manual_seed(seed)
    data = torch.randn(B, 3, S, S, generator=gen)
    target = torch.randint(0, 19, (B, S, S), generator=gen)
    logits = torch.rand(B, S, S, generator=gen)
    np.random.seed(seed)
    state = np.random.get_state()
    nd, nt, nl = ns.augmentation.generate_unsup_data(data, target.clone(), logits.clone(), mode="cutmix")
    # recover the boxes the reference drew, by replaying np.random in the same order
    np.random.set_state(state)
    boxes = []
    for _ in range(B):
        area = S * S / 2
        w = np.random.randint(S / 2 + 1, S)
        h = np.round(area / w)
        x0 = np.random.randint(0, S - w + 1)
        y0 = np.random.randint(0, S - h + 1)
        boxes.append((int(y0), int(y0 + h), int(x0), int(x0 + w)))
    save("cutmix", data=data, target=target, logits=logits, seed=np.int64(seed), boxes=np.array(boxes),
         new_data=nd, new_target=nt, new_logits=nl)


def gen_strong_aug(ns, seed, B, S):
    """reference generate_unsup_data for the two other modes (augmentation.py:486-541): "cutout" (boxes replayed from
    np.random like gen_cutmix) and "classmix" (the per-image selected classes replayed from torch.randperm)."""
    gen = torch.Generator().manual_seed(seed)
    data = torch.randn(B, 3, S, S, generator=gen)
    data[0, 0, :3, :3] = -0.0            # signed zero / x*0 behaviour is part of the contract
    target = block_labels(B, S, 19, gen, ignore_rows=0, cell=16)
    target[1][target[1] == 3] = 5        # images with different class sets
    logits = torch.rand(B, S, S, generator=gen)
    np.random.seed(seed)
    state = np.random.get_state()
    nd, nt, nl = ns.augmentation.generate_unsup_data(data, target.clone(), logits.clone(), mode="cutout")
    np.random.set_state(state)
    boxes = []
    for _ in range(B):
        area = S * S / 2
        w = np.random.randint(S / 2 + 1, S)
        h = np.round(area / w)
        x0 = np.random.randint(0, S - w + 1)
        y0 = np.random.randint(0, S - h + 1)
        boxes.append((int(y0), int(y0 + h), int(x0), int(x0 + w)))
    fx = dict(data=data, target=target.to(torch.uint8), logits=logits, seed=np.int64(seed), boxes=np.array(boxes),
              cutout_data=nd, cutout_target=nt.to(torch.uint8), cutout_logits=nl)
    torch.manual_seed(seed)
    nd, nt, nl = ns.augmentation.generate_unsup_data(data, target.clone(), logits.clone(), mode="classmix")
    torch.manual_seed(seed)
    sel = np.zeros((B, 32), np.int64) - 1
    for i in range(B):
        labels = torch.unique(target[i])
        ch = labels[torch.randperm(len(labels))][: len(labels) // 2].numpy()
        sel[i, : len(ch)] = ch
    fx.update(classmix_selected=sel, classmix_data=nd, classmix_target=nt.to(torch.uint8), classmix_logits=nl)
    save("strong_aug", **fx)


def gen_pseudo(seed, S, s, C):
    gen = torch.Generator().manual_seed(seed)
    low = torch.randn(2, C, s, s, generator=gen) * 3
    large = F.interpolate(low, (S, S), mode="bilinear", align_corners=True)
    prob = F.softmax(large, dim=1)
    conf, label = torch.max(prob, dim=1)
    top2 = torch.sort(large, 1, True)[0][:, :2]
    save("pseudo_65", low=low, large=large, conf=conf, label=label, gap=(top2[:, 0] - top2[:, 1]))


def gen_sgd_ema(ns, seed):
    gen = torch.Generator().manual_seed(seed)
    p0 = torch.randn(1000, generator=gen)
    p1 = torch.randn(500, generator=gen)
    s0, s1 = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p1.clone())
    t0, t1 = p0.clone() * 0.5, p1.clone() * 0.5
    params = [dict(params=[s0], lr=0.01), dict(params=[s1], lr=0.1)]
    opt = ns.lr_helper.get_optimizer(params, dict(type="SGD", kwargs=dict(lr=0.01, momentum=0.9, weight_decay=0.0005)))
    sched = ns.lr_helper.get_scheduler(dict(epochs=2, lr_scheduler=dict(mode="poly", kwargs=dict(power=0.9))), 5, opt, 0)
    fx = dict(p0=p0, p1=p1, t0=t0.clone(), t1=t1.clone())
    for it in range(6):
        sched.step()
        g0 = torch.randn(1000, generator=gen)
        g1 = torch.randn(500, generator=gen)
        s0.grad, s1.grad = g0.clone(), g1.clone()
        opt.step()
        d = min(1 - 1 / (it - 5 * 0 + 1), 0.99)  # train_semi.py:533-542 with sup_only_epoch 0
        t0 = d * t0 + (1 - d) * s0.data
        t1 = d * t1 + (1 - d) * s1.data
        fx.update({f"g0_{it}": g0, f"g1_{it}": g1, f"s0_{it}": s0.data.clone(), f"s1_{it}": s1.data.clone(),
                   f"t0_{it}": t0.clone(), f"t1_{it}": t1.clone(),
                   f"lr_{it}": np.array([g["lr"] for g in opt.param_groups]), f"ema_{it}": np.float64(d)})
    save("sgd_ema", **fx)


# ----------------------------------------------------------------------------
def formula_state_dict(model, seed=1234):
    """Deterministic, machine-independent weights: every tensor is filled from a
    closed-form function of its name and flat index, so both the reference model
    (here) and the MI355X model (on the GPU box) can build identical weights
    without shipping a checkpoint."""
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        if v.dtype == torch.long:
            out[k] = v.clone()
            continue
        h = (sum((i + 1) * ord(c) for i, c in enumerate(k)) * 2654435761 + seed) % (2 ** 31)
        n = v.numel()
        idx = torch.arange(n, dtype=torch.float64)
        u = torch.frac(torch.sin(idx * 12.9898 + (h % 10007) * 0.618) * 43758.5453).abs()  # [0,1)
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


DROPOUT_SEED_MODEL = 2024


def gen_model(ns, tag, arch, S, B, C, aux, seed=5):
    import copy

    net = dict(
        num_classes=C, sync_bn=False, ema_decay=0.99,
        encoder=dict(type=f"u2pl.models.resnet.{arch}",
                     kwargs=dict(multi_grid=True, zero_init_residual=True, fpn=True,
                                 replace_stride_with_dilation=[False, True, True], pretrained=False)),
        decoder=dict(type="u2pl.models.decoder.dec_deeplabv3_plus", kwargs=dict(inner_planes=256, dilations=[12, 24, 36])),
    )
    if aux:
        net["aux_loss"] = dict(aux_plane=1024, loss_weight=0.4)
    model = ns.model_helper.ModelBuilder(copy.deepcopy(net))
    model.load_state_dict(formula_state_dict(model))
    # dropout ON (p = 0.1) with explicit keyed keep-masks (parity_dropout): CPU mt19937 vs device RNG cannot match
    import parity_dropout as PD
    PD.tag_model(model, "student")
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, S, S, generator=gen)
    model.train()
    with PD.patched_torch_dropout2d(PD.KeyedMasks(DROPOUT_SEED_MODEL)):
        out = model(x)
    gp = torch.randn(out["pred"].shape, generator=gen)
    gr = torch.randn(out["rep"].shape, generator=gen)
    loss = (out["pred"] * gp).sum() + (out["rep"] * gr).sum()
    if aux:
        ga = torch.randn(out["aux"].shape, generator=gen)
        loss = loss + (out["aux"] * ga).sum()
    loss.backward()
    names = ["encoder.conv1.0.weight", "encoder.layer1.0.conv2.weight", "encoder.layer2.0.downsample.0.weight",
             "encoder.layer3.1.conv2.weight", "encoder.layer4.2.conv3.weight", "encoder.layer4.2.bn3.weight",
             "decoder.aspp.conv5.0.weight", "decoder.aspp.conv1.1.weight", "decoder.low_conv.0.bias",
             "decoder.classifier.8.weight", "decoder.representation.4.bias", "decoder.head.1.bias"]
    if aux:
        names.append("auxor.aux.0.weight")
    params = dict(model.named_parameters())
    fx = dict(x=x, pred=out["pred"], rep=out["rep"], gp=gp, gr=gr)
    if aux:
        fx.update(aux=out["aux"], ga=ga)
    for n in names:
        g = params[n].grad.flatten()
        fx["grad__" + n] = g[:: max(1, g.numel() // 4096)][:4096]   # strided sample (logical OIHW order)
        fx["gabs__" + n] = g.double().abs().sum()
    bufs = dict(model.named_buffers())
    for n in ["encoder.bn1.running_mean", "encoder.bn1.running_var", "decoder.aspp.conv1.2.running_var",
              "encoder.layer4.2.bn3.running_mean"]:
        fx["buf__" + n] = bufs[n]
    model.eval()
    with torch.no_grad():
        oe = model(x)
    fx.update(pred_eval=oe["pred"], rep_eval=oe["rep"])
    fx["grad_names"] = np.array(names)
    # float64 ground truth of the same model: lets tests bound the HIP path's error by the
    # reference's OWN fp32 rounding error (|ref32 - ref64|) instead of an arbitrary tolerance
    m64 = ns.model_helper.ModelBuilder(copy.deepcopy(net))
    m64.load_state_dict(formula_state_dict(m64))
    PD.tag_model(m64, "student")
    m64 = m64.double().train()
    with PD.patched_torch_dropout2d(PD.KeyedMasks(DROPOUT_SEED_MODEL)):
        o64 = m64(x.double())
    l64 = (o64["pred"] * gp.double()).sum() + (o64["rep"] * gr.double()).sum()
    if aux:
        l64 = l64 + (o64["aux"] * ga.double()).sum()
    l64.backward()
    p64 = dict(m64.named_parameters())
    fx.update(pred64=o64["pred"].float(), rep64=o64["rep"].float())
    if aux:
        fx["aux64"] = o64["aux"].float()
    for n in names:
        g = p64[n].grad.flatten()
        fx["grad64__" + n] = g[:: max(1, g.numel() // 4096)][:4096].float()
    m64.eval()
    with torch.no_grad():
        oe64 = m64(x.double())
    fx.update(pred_eval64=oe64["pred"].float(), rep_eval64=oe64["rep"].float())
    fx["dropout_seed"] = np.int64(DROPOUT_SEED_MODEL)
    save(f"model_{tag}", **fx)


AUG_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], ignore_label=255, flip=True,
               rand_resize=[0.5, 2.0], crop=dict(type="rand", size=[49, 57]))


def gen_augment(ns):
    """reference transform chain exactly as cityscapes.build_transfrom composes it (cityscapes.py:47-77 ->
    augmentation.py ToTensor / Normalize / RandResize / RandomHorizontalFlip / Crop) on one synthetic sample, for
    several python-`random` seeds (covers up- and down-scaling, both flip outcomes, padding when the resized
    image is smaller than the crop)."""
    import importlib
    import random

    from PIL import Image
    city = importlib.import_module("u2pl.dataset.cityscapes")
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (60, 84, 3), dtype=np.uint8)
    lab = rng.integers(0, 19, (60, 84), dtype=np.uint8)
    lab[:4] = 255
    fx = dict(img=img, lab=lab)
    seeds = [0, 1, 2, 3, 4, 5, 8, 13]
    for sd in seeds:
        random.seed(sd)
        tf = city.build_transfrom(AUG_CFG)
        oi, ol = tf(Image.fromarray(img), Image.fromarray(lab))
        fx[f"img_{sd}"] = oi[0]
        fx[f"lab_{sd}"] = ol[0, 0].long().to(torch.uint8)
        fx[f"next_{sd}"] = np.float64(random.random())
    fx["seeds"] = np.array(seeds)
    cfgv = dict(AUG_CFG, flip=False, rand_resize=False, crop=dict(type="center", size=[49, 57]))
    oi, ol = city.build_transfrom(cfgv)(Image.fromarray(img), Image.fromarray(lab))
    fx["img_center"], fx["lab_center"] = oi[0], ol[0, 0].long().to(torch.uint8)
    save("augment", **fx)


def _load_ref_train_semi():
    import importlib.util

    spec = importlib.util.spec_from_file_location("u2pl_ref_train_semi", os.path.join(ref_shim.REFERENCE_ROOT, "train_semi.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    return ts


def _train_cfg(voc, arch, C, min_kept, class_thr, epochs=20):
    cfg = dict(
        dataset=dict(type="cityscapes_semi", n_sup=744, ignore_label=255),
        trainer=dict(epochs=epochs, sup_only_epoch=0,
                     optimizer=dict(type="SGD", kwargs=dict(lr=0.01, momentum=0.9, weight_decay=0.0005)),
                     lr_scheduler=dict(mode="poly", kwargs=dict(power=0.9)),
                     unsupervised=dict(TTA=False, drop_percent=80, apply_aug="cutmix"),
                     contrastive=dict(negative_high_entropy=True, low_rank=3, high_rank=20, current_class_threshold=class_thr,
                                      current_class_negative_threshold=1, unsupervised_entropy_ignore=80,
                                      low_entropy_threshold=20, num_negatives=50, num_queries=256, temperature=0.5)),
        criterion=dict(type="ohem", kwargs=dict(thresh=0.7, min_kept=min_kept)),
        net=dict(num_classes=C, sync_bn=False, ema_decay=0.99,
                 encoder=dict(type=f"u2pl.models.resnet.{arch}",
                              kwargs=dict(multi_grid=True, zero_init_residual=True, fpn=True,
                                          replace_stride_with_dilation=[False, True, True], pretrained=False)),
                 decoder=dict(type="u2pl.models.decoder.dec_deeplabv3_plus", kwargs=dict(inner_planes=256, dilations=[12, 24, 36])),
                 aux_loss=dict(aux_plane=1024, loss_weight=0.4)),
    )
    if voc:   # experiments/pascal/1464/ours flavour: C=21, no aux head, plain CE, head lr x10, sup_only_epoch = 1
        cfg["dataset"]["type"] = "pascal_semi"
        cfg["trainer"].pop("sup_only_epoch")
        cfg["trainer"]["optimizer"]["kwargs"].update(lr=0.001, weight_decay=0.0001)
        cfg["criterion"] = dict(type="CELoss", kwargs=dict(use_weight=False))
        cfg["net"].pop("aux_loss")
    return cfg


def _pack_classbits(onehot):
    """(N,C,h,w) {0,1} -> (N,h,w) int32 bit c = class c (the product's lbits layout)."""
    oh = onehot.detach().cpu().numpy().astype(np.int64)
    bits = np.zeros((oh.shape[0],) + oh.shape[2:], np.int64)
    for c in range(oh.shape[1]):
        bits |= oh[:, c] << c
    return bits.astype(np.int32)


def _run_reference_train(ns, cfg, data, steps, epochs_run, voc, init_seed=0, np_seed=31, torch_seed=41, p_drop=0.0,
                         dropout_seed=None, sharpen=None, capture=None, threads=None, ddp=False, perturb=None):
    """Drive the reference's OWN train() (train_semi.py:234-594) over `data` (list of (il, ll, iu), `steps` per epoch
    in `epochs_run`) with fake loaders, plain BN and a gloo world of 1.  Dropout: p_drop = 0, or p = 0.1 with the
    keep-masks of oracle/parity_dropout.KeyedMasks(dropout_seed) (nn.Dropout2d.forward patched at run time; no
    reference file is modified).  capture: list that receives one dict per semi-supervised step with the arguments /
    side effects of compute_unsupervised_loss and compute_contra_memobank_loss (the mask tensors the product must
    reproduce)."""
    import contextlib
    import copy
    import logging

    import parity_dropout as PD

    ts = _load_ref_train_semi()
    C = cfg["net"]["num_classes"]
    ts.cfg = cfg
    torch.manual_seed(init_seed)
    model = ns.model_helper.ModelBuilder(copy.deepcopy(cfg["net"]))
    if sharpen:
        with torch.no_grad():
            model.decoder.classifier[8].weight.mul_(sharpen)
    if perturb:     # (noise-floor runs of the mIoU gate: the initial weights moved by a relative N(0, perturb) each)
        gp = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for p_ in model.parameters():
                p_.mul_(1 + perturb * torch.randn(p_.shape, generator=gp))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    teacher = ns.model_helper.ModelBuilder(copy.deepcopy(cfg["net"]))
    teacher.load_state_dict(sd)
    for m in list(model.modules()) + list(teacher.modules()):
        if isinstance(m, torch.nn.Dropout2d):
            m.p = p_drop
    masks = None
    if dropout_seed is not None:
        PD.tag_model(model, "student")
        PD.tag_model(teacher, "teacher")
        masks = PD.KeyedMasks(dropout_seed)
    for p_ in teacher.parameters():
        p_.requires_grad = False
    cfg_optim = cfg["trainer"]["optimizer"]
    times = 10 if "pascal" in cfg["dataset"]["type"] else 1          # train_semi.py:100-110
    params_list = [dict(params=model.encoder.parameters(), lr=cfg_optim["kwargs"]["lr"])]
    if not voc:
        params_list.append(dict(params=model.auxor.parameters(), lr=cfg_optim["kwargs"]["lr"] * times))
    params_list.append(dict(params=model.decoder.parameters(), lr=cfg_optim["kwargs"]["lr"] * times))
    optimizer = ts.get_optimizer(params_list, cfg_optim)
    sup_loss_fn = ts.get_criterion(cfg)
    raw_model = model
    if ddp:   # train_semi.py:114-120 (CPU / gloo flavour of the same wrapper)
        model = torch.nn.parallel.DistributedDataParallel(model)

    class _It:
        def __init__(self, items):
            self.items, self.i = items, 0

        def next(self):                     # the reference calls the py2-style iterator.next()
            self.i += 1
            return self.items[self.i - 1]

        __next__ = next

        def __iter__(self):
            return self

    class _Loader:
        class sampler:
            @staticmethod
            def set_epoch(e):
                pass

        def __init__(self, items):
            self.items = items

        def __len__(self):
            return len(self.items)

        def __iter__(self):
            return _It(self.items)

    loader_l = _Loader([(a, b) for a, b, _ in data[:steps]])
    loader_u = _Loader([(c, None) for _, _, c in data[:steps]])
    optimizer_start = ts.get_optimizer(params_list, cfg_optim)
    lr_scheduler = ts.get_scheduler(cfg["trainer"], len(loader_l), optimizer_start, start_epoch=0)
    memobank, queue_ptrlis, queue_size = [], [], []
    for i in range(C):
        memobank.append([torch.zeros(0, 256)])
        queue_size.append(30000)
        queue_ptrlis.append(torch.zeros(1, dtype=torch.long))
    queue_size[0] = 50000
    ts.prototype = torch.zeros((C, 256, 1, 256))
    rec = []

    class _Meter:
        def __init__(self, *a, **k):
            self.val = self.avg = 0.0

        def update(self, v, *a):
            self.val = self.avg = v
            rec.append(float(v))
    ts.AverageMeter = _Meter
    if capture is not None:
        f_unsup, f_contra = ts.compute_unsupervised_loss, ts.compute_contra_memobank_loss

        def unsup(predict, target, percent, pred_teacher):
            d = dict(label_u_aug=target.clone(), percent=float(percent))
            out = f_unsup(predict, target, percent, pred_teacher)
            d["target_u"] = target.clone()                      # mutated in place (loss_helper.py:41-43)
            capture.append(d)
            return out

        def contra(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, *a, **k):
            d = capture[-1]
            d.update(low_mask=low_mask.clone(), high_mask=high_mask.clone(),
                     lbits=np.concatenate([_pack_classbits(label_l), _pack_classbits(label_u)]))
            out = f_contra(rep, label_l, label_u, prob_l, prob_u, low_mask, high_mask, *a, **k)
            d["bank_len"] = np.array([m[0].shape[0] for m in memobank])
            return out
        ts.compute_unsupervised_loss, ts.compute_contra_memobank_loss = unsup, contra
    if threads:
        torch.set_num_threads(threads)
    np.random.seed(np_seed)
    torch.manual_seed(torch_seed)
    ctx = PD.patched_torch_dropout2d(masks) if masks is not None else contextlib.nullcontext()
    with ctx:
        for e in epochs_run:
            k = epochs_run.index(e)
            loader_l.items = [(a, b) for a, b, _ in data[k * steps:(k + 1) * steps]]
            loader_u.items = [(c, None) for _, _, c in data[k * steps:(k + 1) * steps]]
            ts.train(model, teacher, optimizer, lr_scheduler, sup_loss_fn, loader_l, loader_u, e, ts.SummaryWriter(),
                     logging.getLogger("gen_golden"), memobank, queue_ptrlis, queue_size)
    # meters are updated per step in the order data_time, lr, sup, uns, con, batch_time
    nst = steps * len(epochs_run)
    per = len(rec) // nst
    return dict(meters=np.array(rec).reshape(nst, per), per=per, model=raw_model, teacher=teacher, memobank=memobank,
                queue_ptrlis=queue_ptrlis, sd=sd, masks=masks)


def gen_train_steps(ns, voc=False):
    """The reference's OWN train() (train_semi.py:234-594) for three optimizer steps on fixed inputs: tiny R50 at
    65x65, batch 2+2, OHEM + aux, CutMix, contrastive bank; fake loaders, plain BN, dropout off.  Records the per-step
    losses, the CutMix coin / RNG seeds and a few parameters afterwards: pins oracle/step_ref.CpuStepRef (the
    composition of the individually pinned pieces) to the real loop."""
    S, B, C, steps = 65, 2, 19, 3
    epochs_run = [0]
    if voc:
        C, steps, epochs_run = 21, 2, [0, 1]
    cfg = _train_cfg(voc, "resnet50", C, min_kept=2000, class_thr=0.05 if voc else 0.055)
    gen = torch.Generator().manual_seed(77)
    data = []
    for _ in range(steps * len(epochs_run)):
        il, iu = torch.randn(B, 3, S, S, generator=gen), torch.randn(B, 3, S, S, generator=gen)
        data.append((il, block_labels(B, S, C, gen, ignore_rows=4), iu))
    r = _run_reference_train(ns, cfg, data, steps, epochs_run, voc)
    model, teacher, memobank, queue_ptrlis = r["model"], r["teacher"], r["memobank"], r["queue_ptrlis"]
    fx = dict(meters=r["meters"], n_meters=np.int64(r["per"]), seeds=np.array([0, 77, 31, 41]), steps=np.int64(steps),
              epochs=np.array(epochs_run),
              bank_len=np.array([m[0].shape[0] for m in memobank]), bank_ptr=np.array([int(q[0]) for q in queue_ptrlis]))
    for k in ("encoder.conv1.0.weight", "decoder.classifier.8.weight", "decoder.representation.8.bias", "auxor.aux.4.bias",
              "encoder.layer3.2.bn2.weight"):
        if voc and k.startswith("auxor"):
            continue
        fx["student__" + k] = dict(model.named_parameters())[k].detach().clone()
        fx["teacher__" + k] = dict(teacher.named_parameters())[k].detach().clone()
    fx["teacher_bn__encoder.bn1.running_mean"] = dict(teacher.named_buffers())["encoder.bn1.running_mean"].clone()
    for i, (il, ll, iu) in enumerate(data):
        fx[f"il_{i}"], fx[f"ll_{i}"], fx[f"iu_{i}"] = il, ll.to(torch.uint8), iu
    save("train_steps_voc" if voc else "train_steps", **fx)


WORLD2 = dict(S=65, B=2, C=19, steps=2, arch="resnet50", sharpen=4.0, data_seed=500, dropout_seed=4321)


def _world2_worker(rank, port, ret):
    """one rank of the reference's train() under a gloo world of 2 (DDP on CPU, plain per-rank BN): different data
    per rank, identical seeds / weights -- pins the cross-rank conventions: contrastive value = cross-rank mean with
    gradient local/world (Q5, train_semi.py:514-519), DDP gradient mean, meters = cross-rank SUMS (train_semi.py:551-561),
    rank-major bank gather (utils.py:16-24)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    ns = ref_shim.load()
    w = WORLD2
    cfg = _train_cfg(False, w["arch"], w["C"], min_kept=2000, class_thr=0.3)
    data = survey_step_inputs(w["data_seed"] + rank, w["B"], w["S"], w["C"], w["steps"])
    cap = []
    r = _run_reference_train(ns, cfg, data, w["steps"], [0], False, sharpen=w["sharpen"], capture=cap, ddp=True, threads=4,
                             p_drop=0.1, dropout_seed=w["dropout_seed"])
    out = dict(meters=r["meters"], bank_len=np.array([m[0].shape[0] for m in r["memobank"]]),
               bank_sum=np.array([float(m[0].double().sum()) for m in r["memobank"]]))
    for k in ("encoder.conv1.0.weight", "decoder.classifier.8.weight", "decoder.representation.8.bias", "auxor.aux.4.bias",
              "encoder.layer3.2.bn2.weight"):
        out["student__" + k] = dict(r["model"].named_parameters())[k].detach().numpy().copy()
        out["teacher__" + k] = dict(r["teacher"].named_parameters())[k].detach().numpy().copy()
    ret[rank] = out
    dist.destroy_process_group()


def gen_train_world2():
    import socket
    import torch.multiprocessing as mp
    sck = socket.socket()
    sck.bind(("127.0.0.1", 0))
    port = sck.getsockname()[1]
    sck.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_world2_worker, args=(port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert np.array_equal(r0["bank_len"], r1["bank_len"]) and np.array_equal(r0["bank_sum"], r1["bank_sum"])
    fx = dict(meters_rank0=r0["meters"], meters_rank1=r1["meters"], bank_len=r0["bank_len"], bank_sum=r0["bank_sum"],
              cfg=np.array([WORLD2["S"], WORLD2["B"], WORLD2["C"], WORLD2["steps"], WORLD2["data_seed"]]),
              sharpen=np.float64(WORLD2["sharpen"]), seeds=np.array([0, 31, 41, WORLD2["dropout_seed"]]))
    for k, v in r0.items():
        if k.startswith("student__") or k.startswith("teacher__"):
            assert np.array_equal(v, r1[k]), k          # DDP keeps the replicas identical
            fx[k] = v
    save("train_world2", **fx)


def survey_step_inputs(seed, B, S, C, n):
    """SURVEY 8(d) synthetic step inputs, regenerated from the seed wherever they are needed (never stored: 28 MB per
    step at 769^2): images N(0,1); labels randint(0,C) on an (S//16+1)^2 grid, nearest-up-sampled, first 8 rows 255.
    tests/full_size.py holds the same function for the GPU box (kept identical by test_oracle_golden)."""
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


FULL_SIZE = {   # tag: (voc, arch, S, B, C, steps, epochs_run, sharpen, dropout_seed, input seed)
    # BASELINE configs[2]/[3]: Cityscapes, R101, 769^2, 2+2 per GPU, OHEM(0.7, 100000) + aux 0.4, CutMix, contrastive
    "city769": (False, "resnet101", 769, 2, 19, 1, [0], 4.0, 1234, 2),
    # BASELINE configs[1]: VOC, R101, 513^2, 4+4, plain CE, no aux, sup_only_epoch 1 -> one sup-only step + one semi step
    "voc513": (True, "resnet101", 513, 4, 21, 1, [0, 1], 4.0, 1234, 2),
    # cheap variants of the same two harness paths for the CPU suite (port <-> reference with dropout ON)
    "city97": (False, "resnet50", 97, 2, 19, 2, [0], 4.0, 1234, 2),
}


def gen_train_full(ns, tag):
    """BASELINE-size step-0 golden from the reference's OWN train(): default configuration values (class threshold
    0.3, OHEM min_kept 100000), dropout ON with keyed keep-masks, classifier last layer x4 (as bench.py: random-init
    logits are near-uniform otherwise).  Inputs are regenerated from the seed; the fixture stores the meters, the
    packed reliability masks / targets and bank bookkeeping."""
    voc, arch, S, B, C, steps, epochs_run, sharpen, dseed, iseed = FULL_SIZE[tag]
    cfg = _train_cfg(voc, arch, C, min_kept=100000 if S > 200 else 4000, class_thr=0.3, epochs=200 if S > 200 else 20)
    data = survey_step_inputs(iseed, B, S, C, steps * len(epochs_run))
    cap = []
    import time
    t0 = time.time()
    r = _run_reference_train(ns, cfg, data, steps, epochs_run, voc, p_drop=0.1, dropout_seed=dseed, sharpen=sharpen,
                             capture=cap)
    print(tag, "reference train():", round(time.time() - t0, 1), "s; meters", r["meters"][:, 1:5])
    fx = dict(meters=r["meters"], seeds=np.array([0, iseed, 31, 41, dseed]), steps=np.int64(steps), epochs=np.array(epochs_run),
              sharpen=np.float64(sharpen), geom=np.array([S, B, C]),
              bank_len=np.array([m[0].shape[0] for m in r["memobank"]]),
              bank_ptr=np.array([int(q[0]) for q in r["queue_ptrlis"]]),
              dropout_log=np.array([f"{t}|{k}|{n}|{c}|{s}" for t, k, n, c, s in r["masks"].log]))
    for i, d in enumerate(cap):
        lab, tgt = d["label_u_aug"].numpy(), d["target_u"].numpy()
        fx[f"s{i}_label_u"] = lab.astype(np.uint8)
        fx[f"s{i}_dropped"] = np.packbits((tgt == 255) & (lab != 255))
        fx[f"s{i}_percent"] = np.float64(d["percent"])
        if "low_mask" in d:
            fx[f"s{i}_low"] = np.packbits(d["low_mask"].numpy() != 0)
            fx[f"s{i}_high"] = np.packbits(d["high_mask"].numpy() != 0)
            fx[f"s{i}_lbits"] = d["lbits"]
            fx[f"s{i}_bank_len"] = d["bank_len"]
    for k in ("encoder.conv1.0.weight", "decoder.classifier.8.weight", "decoder.representation.8.bias",
              "encoder.layer3.2.bn2.weight"):
        fx["student__" + k] = dict(r["model"].named_parameters())[k].detach().clone()
    save("train_full_" + tag, **fx)


def _reference_validate(ns, cfg, model, val_batches):
    """the reference's own validate() (train_semi.py:595-654) over in-memory batches -> (mIoU, per-class IoU)"""
    import logging

    ts = _load_ref_train_semi()
    ts.cfg = cfg
    got = {}

    class _Meter(ts.AverageMeter):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            got.setdefault("meters", []).append(self)

    ts.AverageMeter = _Meter

    class _Loader:
        class sampler:
            @staticmethod
            def set_epoch(e):
                pass

        def __iter__(self):
            return iter(val_batches)

    miou = ts.validate(model, _Loader(), 0, logging.getLogger("gen_golden"))
    inter, union = got["meters"][0], got["meters"][1]
    return float(miou), inter.sum / (union.sum + 1e-10)


def gen_miou_gate(ns, steps=None, tag="miou_gate", noise_floor=True):
    """north_star gate: "mIoU on a fixed 50-image val subset within +-0.3 of the CPU reference after 1 epoch".  The
    REFERENCE's own train() (train_semi.py:234-592) runs one epoch (40 steps, R101, 193x193, 2 + 2 images, OHEM + aux,
    CutMix, contrastive bank, dropout ON with keyed masks) on the learnable synthetic task of tests/miou_gate.py from its
    own seeded initialisation, then its own validate() (train_semi.py:595-654) scores the EMA teacher on the 50
    validation images.  Also stored: the mIoU of the initial weights (the gate asserts that training moved it), the
    per-step losses, and a NOISE-FLOOR run -- the same epoch from initial weights perturbed by 1e-7 relative (one fp32
    rounding): what two fp32 implementations of the same step may legitimately differ by after 40 chaotic steps."""
    import copy
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import miou_gate as MG

    G = dict(MG.GATE)
    if steps:
        G["steps"] = steps
    cfg = _train_cfg(False, G["arch"], G["C"], min_kept=G["min_kept"], class_thr=G["class_thr"], epochs=G["epochs"])
    data = MG.gate_data(G["data_seed"], G["steps"], G["B"], G["S"])
    val = MG.gate_val(G["data_seed"] + 1, G["n_val"], G["S"])
    fx = dict(digest=MG.data_digest(data, val), steps=np.int64(G["steps"]),
              seeds=np.array([G["init_seed"], G["data_seed"], G["np_seed"], G["torch_seed"], G["dropout_seed"]]))
    runs = [("", None)] + ([("noise_", 1e-7)] if noise_floor else [])
    for pre, perturb in runs:
        t0 = time.time()
        r = _run_reference_train(ns, copy.deepcopy(cfg), data, G["steps"], [0], False, init_seed=G["init_seed"],
                                 np_seed=G["np_seed"], torch_seed=G["torch_seed"], p_drop=0.1, dropout_seed=G["dropout_seed"],
                                 perturb=perturb)
        t1 = time.time()
        if not pre:
            init = ns.model_helper.ModelBuilder(copy.deepcopy(cfg["net"]))
            init.load_state_dict(r["sd"])
            fx["miou_init"], fx["iou_init"] = _reference_validate(ns, cfg, init, val)
        miou_t, iou_t = _reference_validate(ns, cfg, r["teacher"], val)
        miou_s, iou_s = _reference_validate(ns, cfg, r["model"], val)
        fx.update({pre + "miou_teacher": miou_t, pre + "iou_teacher": iou_t, pre + "miou_student": miou_s,
                   pre + "iou_student": iou_s, pre + "meters": r["meters"],
                   pre + "bank_len": np.array([m[0].shape[0] for m in r["memobank"]])})
        print(tag, pre or "reference", "train %.0f s" % (t1 - t0), "mIoU init %.2f teacher %.2f student %.2f" %
              (100 * float(fx["miou_init"]), 100 * miou_t, 100 * miou_s), "last losses", r["meters"][-1, 1:5], flush=True)
    save(tag, **fx)


def gen_resample(ns):
    """reference city_dset.__init__ (cityscapes.py:18-33): the seeded random.sample of the (tiled) list, both
    regimes (list longer / shorter than n_sup)."""
    import importlib
    import tempfile
    city = importlib.import_module("u2pl.dataset.cityscapes")
    lines = [f"leftImg8bit/train/synth/synth_{i:06d}_000019_leftImg8bit.png" for i in range(11)]
    d = tempfile.mkdtemp(prefix="cityscapes_")
    lp = os.path.join(d, "labeled.txt")
    open(lp, "w").write("\n".join(lines) + "\n")
    fx = dict(lines=np.array(lines))
    for tag, seed, n_sup in (("short", 2, 7), ("tiled", 2, 30), ("seed5", 5, 11)):
        ds = city.city_dset(d, lp, None, seed, n_sup, "train")
        fx["order_" + tag] = np.array([a[0] for a in ds.list_sample_new])
        fx["labels_" + tag] = np.array([a[1] for a in ds.list_sample_new])
        fx["cfg_" + tag] = np.array([seed, n_sup])
    save("resample", **fx)


def gen_miou_hist(ns, seed):
    """reference utils.intersectionAndUnion (utils.py:568-580) on argmax maps with ignored pixels."""
    rng = np.random.default_rng(seed)
    K = 19
    out = rng.integers(0, K, (3, 37, 41)).astype(np.uint8)
    tgt = rng.integers(0, K, (3, 37, 41)).astype(np.uint8)
    tgt[rng.random(tgt.shape) < 0.1] = 255
    tgt[0, :5] = 255
    tgt[tgt == 7] = 3                       # a class that never occurs in the ground truth
    i, u, t = ns.utils.intersectionAndUnion(out, tgt, K)
    save("miou_hist", out=out, tgt=tgt, inter=np.asarray(i), union=np.asarray(u), target=np.asarray(t))


def gen_eval_window(ns, tag, H, W, crop, seed):
    """reference eval.py:184-224 scale_crop_process on a tiny R50 (formula weights, eval mode), fp32 and fp64."""
    import copy
    import importlib.util

    spec = importlib.util.spec_from_file_location("u2pl_ref_eval", os.path.join(ref_shim.REFERENCE_ROOT, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    C = 19
    net = dict(
        num_classes=C, sync_bn=False, ema_decay=0.99,
        encoder=dict(type="u2pl.models.resnet.resnet50",
                     kwargs=dict(multi_grid=True, zero_init_residual=True, fpn=True,
                                 replace_stride_with_dilation=[False, True, True], pretrained=False)),
        decoder=dict(type="u2pl.models.decoder.dec_deeplabv3_plus", kwargs=dict(inner_planes=256, dilations=[12, 24, 36])),
        aux_loss=dict(aux_plane=1024, loss_weight=0.4),
    )
    model = ns.model_helper.ModelBuilder(copy.deepcopy(net))
    model.load_state_dict(formula_state_dict(model))
    model.eval()
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 3, H, W, generator=gen)
    out = ev.scale_crop_process(model, x, C, crop, crop, H, W)
    m64 = ns.model_helper.ModelBuilder(copy.deepcopy(net))
    m64.load_state_dict(formula_state_dict(m64))
    m64 = m64.double().eval()
    _zeros = torch.zeros

    def zeros64(*a, **k):        # the reference allocates its accumulators as torch.float
        k["dtype"] = torch.float64
        return _zeros(*a, **k)
    torch.zeros = zeros64
    try:
        out64 = ev.scale_crop_process(m64, x.double(), C, crop, crop, H, W)
    finally:
        torch.zeros = _zeros
    save(f"evalwin_{tag}", x=x, out=out, out64=out64.float(), crop=np.int64(crop))


def _pattern(key, n, scale, mul):
    """low-entropy closed-form values (period 251, exact in fp32): the ~0.5 GB checkpoint gzips to < 1 MB"""
    h = sum((i + 1) * ord(c) for i, c in enumerate(key)) % 251
    idx = torch.arange(n, dtype=torch.int64)
    return (((idx * mul + h) % 251) - 125).to(torch.float32) / scale


def gen_ref_ckpt(ns):
    """A ckpt.pth produced the way the reference produces it (train_semi.py:61-120 and 210-224), committed gzipped as
    tests/golden/ref_ckpt_r50.pth.gz together with ref_ckpt_r50.npz (which optimizer index is which parameter, by name,
    as seen from the REFERENCE's objects -- the independent statement of the group order our loader must reproduce).

    reference ModelBuilder (R50 + aux head) x 2, the reference's parameter groups [encoder | auxor, decoder]
    (train_semi.py:79-110) into the reference's get_optimizer (lr_helper.py:12-27), DistributedDataParallel wrappers
    (-> `module.` keys), ONE optimizer.step() so that momentum buffers exist, then the dict of train_semi.py:210-216 and
    torch.save.  Parameter / gradient VALUES are low-entropy patterns (not trained weights) so the file compresses."""
    import copy
    import gzip
    import io
    net = dict(
        num_classes=19, sync_bn=False, ema_decay=0.99,
        encoder=dict(type="u2pl.models.resnet.resnet50",
                     kwargs=dict(multi_grid=True, zero_init_residual=True, fpn=True,
                                 replace_stride_with_dilation=[False, True, True], pretrained=False)),
        decoder=dict(type="u2pl.models.decoder.dec_deeplabv3_plus", kwargs=dict(inner_planes=256, dilations=[12, 24, 36])),
        aux_loss=dict(aux_plane=1024, loss_weight=0.4),
    )
    cfg_optim = dict(type="SGD", kwargs=dict(lr=0.01, momentum=0.9, weight_decay=0.0005))
    model = ns.model_helper.ModelBuilder(copy.deepcopy(net))
    modules_back, modules_head = [model.encoder], [model.auxor, model.decoder]       # train_semi.py:79-84
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                v.copy_(_pattern("s:" + k, v.numel(), 256.0, 7).reshape(v.shape))
    params_list = []
    for module in modules_back:                                                      # train_semi.py:102-110
        params_list.append(dict(params=module.parameters(), lr=cfg_optim["kwargs"]["lr"]))
    for module in modules_head:
        params_list.append(dict(params=module.parameters(), lr=cfg_optim["kwargs"]["lr"] * 1))
    optimizer = ns.lr_helper.get_optimizer(params_list, cfg_optim)
    name_of = {id(p): n for n, p in model.named_parameters()}
    for n, p in model.named_parameters():
        p.grad = _pattern("g:" + n, p.numel(), 1024.0, 5).reshape(p.shape)
    optimizer.step()
    model = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=False)    # train_semi.py:114-120
    teacher = ns.model_helper.ModelBuilder(copy.deepcopy(net))
    with torch.no_grad():
        for k, v in teacher.state_dict().items():
            if v.dtype.is_floating_point:
                v.copy_(_pattern("t:" + k, v.numel(), 512.0, 11).reshape(v.shape))
    teacher = torch.nn.parallel.DistributedDataParallel(teacher, find_unused_parameters=False)
    state = {"epoch": 3, "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict(),    # train_semi.py:210-216
             "teacher_state": teacher.state_dict(), "best_miou": 0.4321}
    buf = io.BytesIO()
    torch.save(state, buf)
    raw = buf.getvalue()
    path = os.path.join(OUT, "ref_ckpt_r50.pth.gz")
    with gzip.open(path, "wb", compresslevel=6) as f:
        f.write(raw)
    print("wrote", path, len(raw) >> 20, "MiB ->", os.path.getsize(path) >> 10, "KiB")
    osd = optimizer.state_dict()
    names, group_of = [], []
    flat = [p for g in optimizer.param_groups for p in g["params"]]
    for gi, g in enumerate(osd["param_groups"]):
        for i in g["params"]:
            names.append(name_of[id(flat[i])])
            group_of.append(gi)
    probe = ["encoder.conv1.0.weight", "encoder.layer3.4.conv2.weight", "auxor.aux.0.weight", "decoder.classifier.8.bias",
             "decoder.aspp.conv3.1.weight"]
    fx = dict(opt_names=np.array(names), opt_group=np.array(group_of), probe=np.array(probe),
              group_lr=np.array([g["lr"] for g in osd["param_groups"]]), n_model_keys=np.int64(len(state["model_state"])))
    for n in probe:
        i = names.index(n)
        fx["mom_head__" + n] = osd["state"][i]["momentum_buffer"].flatten()[:16]
        fx["mom_sum__" + n] = osd["state"][i]["momentum_buffer"].double().sum()
        fx["par_head__" + n] = state["model_state"]["module." + n].flatten()[:16]
        fx["tea_head__" + n] = state["teacher_state"]["module." + n].flatten()[:16]
    save("ref_ckpt_r50", **fx)


def main():
    which = set(sys.argv[1:])
    if "world2" in which:       # two gloo ranks: each worker installs the shim inside ITS process group
        gen_train_world2()
        return
    ns = ref_shim.load()

    def want(k):
        return not which or k in which

    if want("unsup"):
        gen_unsup(ns, 11, 65, 17, 19, 80.0, "65_c19")
        gen_unsup(ns, 12, 97, 25, 21, 86.5, "97_c21")
    if want("ohem"):
        gen_ohem(ns, 21, 65, 17, 19, 3000, "65_k3000")
        gen_ohem(ns, 22, 65, 17, 19, 100000, "65_kbig")   # min_kept > num_valid: no filtering
        gen_ohem(ns, 23, 65, 17, 19, 60, "65_k60")         # k-th prob below thresh: threshold stays 0.7
    if want("relsplit"):
        gen_relsplit(ns, 31, 2, 65, 17, 19, 20.0, "65_a20")
        gen_relsplit(ns, 32, 2, 97, 25, 21, 13.7, "97_a13")
        gen_relsplit(ns, 33, 2, 65, 17, 19, 20.0, "65_cutout", cutout=True)
        gen_relsplit(ns, 34, 3, 65, 17, 19, 7.5, "65_b3")
    if want("contra"):
        gen_contra(ns, 41, 2, 65, 17, 19, 20.0, "65_empty", prefill=0, steps=2)
        gen_contra(ns, 42, 2, 65, 17, 19, 20.0, "65_prefill", prefill=2990, steps=2, D=64)
    if want("contra_t007"):     # temperature 0.07 (loss_helper.py:205-230 is max-shifted: any temperature must work)
        gen_contra(ns, 43, 2, 65, 17, 19, 20.0, "65_t007", prefill=2990, steps=1, D=64, temperature=0.07)
    if want("contra_t001"):     # temperature 0.01: 2/temp = 200, the fixed-shift softmax would underflow -> online-max kernel
        gen_contra(ns, 44, 2, 65, 17, 19, 20.0, "65_t001", prefill=2990, steps=1, D=64, temperature=0.01)
    if want("contra_wrap"):     # the REAL capacities (30000, class 0: 50000), pre-filled to just below them: step 0 wraps
        gen_contra(ns, 45, 2, 65, 17, 19, 20.0, "65_wrap", prefill=29996, steps=2, D=64, queue_size=30000,
                   queue_size0=50000, prefill0=49998)
    if want("bank"):
        gen_bank_seq(ns, 51)
    if want("cutmix"):
        gen_cutmix(ns, 61, 2, 65)
    if want("strongaug"):
        gen_strong_aug(ns, 62, 3, 65)
    if want("pseudo"):
        gen_pseudo(71, 65, 17, 19)
    if want("sgd"):
        gen_sgd_ema(ns, 81)
    if want("trainsteps"):
        gen_train_steps(ns)
    if want("trainsteps_voc"):
        gen_train_steps(ns, voc=True)
    for tag in FULL_SIZE:
        if ("full_" + tag) in which or "full" in which:     # minutes of CPU each: only on request
            gen_train_full(ns, tag)
    if "miou_gate" in which:      # ~10 minutes of CPU (two 40-step R101 epochs at 193^2): only on request
        gen_miou_gate(ns)
    if "miou_gate_try" in which:  # quick look at the task (8 steps, no noise-floor run, scratch fixture name)
        gen_miou_gate(ns, steps=int(os.environ.get("GATE_STEPS", "8")), tag="_miou_gate_try", noise_floor=False)
    if want("resample"):
        gen_resample(ns)
    if want("augment"):
        gen_augment(ns)
    if want("miou"):
        gen_miou_hist(ns, 101)
    if want("evalwin"):
        gen_eval_window(ns, "70x100", 70, 100, 65, 91)     # 2 x 2 overlapping windows, last ones pulled back
        gen_eval_window(ns, "50x90", 50, 90, 65, 92)       # image shorter than the crop: symmetric zero padding
    if "refckpt" in which:       # ~0.6 GB in memory: only on request
        gen_ref_ckpt(ns)
    if want("model"):
        gen_model(ns, "r50_65", "resnet50", 65, 2, 19, aux=True)
        gen_model(ns, "r101_33", "resnet101", 33, 2, 21, aux=False)


if __name__ == "__main__":
    main()
