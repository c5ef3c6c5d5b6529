"""TEST INFRASTRUCTURE ONLY -- CPU port of ONE U2PL training step
(reference train_semi.py:272-561), built from the oracle pieces:
  * oracle/model_ref.RefNet          (torch-CPU network, pinned bit-exact to the reference)
  * oracle/restate.py                (numpy: percentile / masks / label_onehot quirk /
                                      contrastive phase 1 / memory bank / OHEM kept set)
  * torch autograd for the differentiable tails (CE, InfoNCE, bilinear) exactly as the
    reference computes them on CPU.
Used (a) as the checker of the HIP training step in tests/test_gpu_train_step.py and
(b) as bench.py's ``cpu_baseline`` (kind "port"), timed on the GPU box's host cores.
Never imported by u2pl_amd.
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import restate as R
from .model_ref import RefNet
from .parity_dropout import patched_torch_dropout2d, tag_model

CONTRA = dict(negative_high_entropy=True, low_rank=3, high_rank=20, current_class_threshold=0.3,
              current_class_negative_threshold=1, low_entropy_threshold=20, num_negatives=50, num_queries=256,
              temperature=0.5)


def _up(x, hw):
    return F.interpolate(x, size=hw, mode="bilinear", align_corners=True)


class CpuStepRef:
    def __init__(self, arch="resnet101", num_classes=19, aux=True, epochs=200, steps_per_epoch=163, lr=0.01,
                 momentum=0.9, weight_decay=0.0005, ema_decay=0.99, sup_only_epoch=0, drop_percent=80,
                 ohem=(0.7, 100000), contra=CONTRA, p_drop=0.0, lr_times=1, queue=(50000, 30000), state_dict=None,
                 apply_aug="cutmix", dropout_masks=None, dtype=torch.float32):
        # dtype=torch.float64: the ARBITER of the multi-step parity tests -- the same step evaluated in double precision
        # (network, losses, thresholds, bank), against which both the fp32 port and the HIP path are measured
        self.dtype = dtype
        self.np_dtype = np.float64 if dtype == torch.float64 else np.float32
        self.student = RefNet(arch, num_classes, aux, p_drop).to(dtype)
        self.teacher = RefNet(arch, num_classes, aux, p_drop).to(dtype)
        # parity mode with dropout ON: both sides take their keep-masks from oracle/parity_dropout.KeyedMasks
        self.dropout_masks = dropout_masks
        if dropout_masks is not None:
            tag_model(self.student, "student")
            tag_model(self.teacher, "teacher")
        if state_dict is not None:
            self.student.load_state_dict(state_dict)      # (copy_ converts fp32 checkpoints to the arbiter's dtype)
            self.teacher.load_state_dict(state_dict)
        for p in self.teacher.parameters():
            p.requires_grad = False
        groups = [dict(params=self.student.encoder.parameters(), lr=lr)]
        if aux:
            groups.append(dict(params=self.student.auxor.parameters(), lr=lr * lr_times))
        groups.append(dict(params=self.student.decoder.parameters(), lr=lr * lr_times))
        self.opt = torch.optim.SGD(groups, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.base_lr = [g["lr"] for g in self.opt.param_groups]
        self.C, self.aux = num_classes, aux
        self.epochs, self.spe, self.ema_decay, self.sup_only_epoch = epochs, steps_per_epoch, ema_decay, sup_only_epoch
        self.drop_percent, self.ohem, self.contra, self.apply_aug = drop_percent, ohem, contra, apply_aug   # ohem=None: plain CE
        self.bank = [[np.zeros((0, 256), self.np_dtype)] for _ in range(num_classes)]
        self.ptr = [[0] for _ in range(num_classes)]
        self.qsize = [queue[0]] + [queue[1]] * (num_classes - 1)
        self.cur_iter = 0

    # ---- pieces -------------------------------------------------------------------------
    def _ohem(self, pred, target):
        if self.ohem is None:   # Criterion (CELoss, loss_helper.py:295-320), no aux weighting needed here
            return F.cross_entropy(pred, target, ignore_index=255)
        thresh, min_kept = self.ohem
        _, kept, _ = R.ohem_ce(pred.detach().numpy(), target.numpy(), thresh, min_kept)
        return F.cross_entropy(pred, torch.from_numpy(kept), ignore_index=255)

    def _sup(self, pred_l_large, aux_large, label_l):
        loss = self._ohem(pred_l_large, label_l)
        if self.aux:
            loss = loss + 0.4 * self._ohem(aux_large, label_l)
        return loss

    def _contra(self, rep, rs, prob_l, prob_u, rep_teacher, randint):
        cfg = self.contra
        ph1 = R.contra_phase1(rep_teacher.numpy(), rs["label_l_small"], rs["label_u_small"], prob_l.numpy(),
                              prob_u.numpy(), rs["low_mask_all"], rs["high_mask_all"], cfg)
        D = rep.shape[1]
        rows = rep.permute(0, 2, 3, 1).reshape(-1, D)
        rows_t = rep_teacher.permute(0, 2, 3, 1).reshape(-1, D).numpy()
        valid, new_keys = [], []
        for i in range(self.C):
            new_keys.append(R.dequeue_and_enqueue(rows_t[ph1[i]["neg_idx"]], self.bank[i], self.ptr[i], self.qsize[i]))
            if ph1[i]["n_low"] > 0:
                valid.append(i)
        info = dict(new_keys=new_keys, valid=valid, njobs=0)
        if len(valid) <= 1:
            return 0 * rep.sum(), info
        loss = torch.tensor(0.0, dtype=self.dtype)
        Q, K = cfg["num_queries"], cfg["num_negatives"]
        for i in range(len(valid)):  # Q1 index mismatch reproduced
            cand, bank = ph1[i]["anchor_idx"], self.bank[valid[i]][0]
            if not (cand.size > 0 and bank.shape[0] > 0):
                loss = loss + 0 * rep.sum()
                continue
            ia = randint(cand.size, Q)
            anchor = rows[torch.from_numpy(cand[ia])]
            inn = randint(bank.shape[0], Q * K)
            neg = torch.from_numpy(bank[inn]).reshape(Q, K, D)
            pos = torch.from_numpy(ph1[i]["proto"].astype(self.np_dtype)).reshape(1, 1, D).repeat(Q, 1, 1)
            logits = torch.cosine_similarity(anchor.unsqueeze(1), torch.cat((pos, neg), 1), dim=2)
            loss = loss + F.cross_entropy(logits / cfg["temperature"], torch.zeros(Q).long())
            info["njobs"] += 1
        return loss / len(valid), info

    # ---- the step -----------------------------------------------------------------------
    def step(self, image_l, label_l, image_u, epoch, cutmix_boxes="draw", randint=None):
        if self.dropout_masks is not None:
            with patched_torch_dropout2d(self.dropout_masks):
                return self._step(image_l, label_l, image_u, epoch, cutmix_boxes, randint)
        return self._step(image_l, label_l, image_u, epoch, cutmix_boxes, randint)

    def _step(self, image_l, label_l, image_u, epoch, cutmix_boxes="draw", randint=None):
        if randint is None:
            def randint(high, n):
                return torch.randint(high, size=(n,)).numpy()
        B, h, w = label_l.shape
        image_l, image_u = image_l.to(self.dtype), image_u.to(self.dtype)
        max_iter = self.epochs * self.spe
        for g, b in zip(self.opt.param_groups, self.base_lr):
            g["lr"] = R.poly_lr(b, self.cur_iter, max_iter)
        i_iter = self.cur_iter
        self.cur_iter += 1
        student, teacher = self.student, self.teacher
        student.train()
        out = {}
        if epoch < self.sup_only_epoch:   # train_semi.py:288-307: labeled images only, teacher BN stats still move
            outs = student(image_l)
            pred = _up(outs["pred"], (h, w))
            aux = _up(outs["aux"], (h, w)) if self.aux else None
            sup_loss = self._sup(pred, aux, label_l)
            teacher.train()
            with torch.no_grad():
                teacher(image_l)
            loss = sup_loss + 0 * outs["rep"].sum() + 0 * outs["rep"].sum()
            self.opt.zero_grad()
            loss.backward()
            self.opt.step()
            out.update(sup=float(sup_loss), unsup=0.0, contra=0.0)
            return out
        if epoch == self.sup_only_epoch:
            with torch.no_grad():
                for t, s in zip(teacher.parameters(), student.parameters()):
                    t.data = s.data  # aliasing, as upstream
        teacher.eval()
        with torch.no_grad():
            pu = F.softmax(_up(teacher(image_u)["pred"], (h, w)), dim=1)
            conf_u, label_u = torch.max(pu, dim=1)
        image_u_aug = image_u
        coin = np.random.uniform(0, 1)
        if coin < 0.5 and self.apply_aug:
            boxes = [R.cutmix_box(h, w, np.random.randint) for _ in range(B)] if cutmix_boxes == "draw" else cutmix_boxes
            a, b, c = R.cutmix_apply(image_u.numpy(), label_u.numpy(), conf_u.numpy(), boxes)
            image_u_aug, label_u, conf_u = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c)
            out["boxes"] = boxes
        outs = student(torch.cat((image_l, image_u_aug)))
        pred_all, rep_all = outs["pred"], outs["rep"]
        pred_l_large, pred_u_large = _up(pred_all[:B], (h, w)), _up(pred_all[B:], (h, w))
        aux_large = _up(outs["aux"][:B], (h, w)) if self.aux else None
        sup_loss = self._sup(pred_l_large, aux_large, label_l)
        teacher.train()
        with torch.no_grad():
            ot = teacher(torch.cat((image_l, image_u_aug)))
            prob_all_t = F.softmax(ot["pred"], dim=1)
            pred_u_large_t = _up(ot["pred"][B:], (h, w))
        drop_percent = 100 - (100 - self.drop_percent) * (1 - epoch / self.epochs)
        ent = R.entropy_from_logits(pred_u_large_t.numpy())
        _, new_target, thr = R.unsup_loss(pred_u_large.detach().numpy(), label_u.numpy(), drop_percent, None, entropy=ent)
        weight = B * h * w / float((new_target != 255).sum())
        unsup_loss = weight * F.cross_entropy(pred_u_large, torch.from_numpy(new_target), ignore_index=255)
        alpha_t = self.contra["low_entropy_threshold"] * (1 - epoch / self.epochs)
        rs = R.reliability_split(None, label_u.numpy(), label_l.numpy(), alpha_t, tuple(pred_all.shape[2:]), self.C,
                                 entropy=ent, negative_high_entropy=self.contra.get("negative_high_entropy", True))
        contra_loss, cinfo = self._contra(rep_all, rs, prob_all_t[:B], prob_all_t[B:], ot["rep"], randint)
        loss = sup_loss + unsup_loss + contra_loss
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        with torch.no_grad():
            d = R.ema_decay(i_iter, self.spe, self.sup_only_epoch, self.ema_decay)
            for t, s in zip(teacher.parameters(), student.parameters()):
                t.data = d * t.data + (1 - d) * s.data
        out.update(sup=float(sup_loss), unsup=float(unsup_loss), contra=float(contra_loss), entropy=ent,
                   new_target=new_target, low_mask=rs["low_mask_all"], high_mask=rs["high_mask_all"],
                   label_u=label_u.numpy(), contra_info=cinfo, coin=coin)
        return out


def timed_cpu_baseline(crop=769, arch="resnet101", batch=2, warmup=1, steps=2):
    """bench.py cpu_baseline leg (SURVEY 8d "CPU reference timing"): `warmup` un-timed + `steps`
    timed reference-equivalent CPU training steps at the FULL per-GPU batch (batch labeled + batch unlabeled crops),
    dropout on.  The port is pinned to the reference's own train() (tests/test_oracle_golden.py, incl. a 769^2 step);
    the reference itself cannot travel to the GPU box, so `kind` is "port" and profiles/r02_cpu_reference_timing.json
    holds the reference's own timing next to the port's in the build container."""
    # torch-CPU conv scaling collapses when oversubscribed (256 threads: 650 s per step on the MI355X host; 32: ~20 s)
    ncores = min(os.cpu_count() or 1, int(os.environ.get("U2PL_CPU_BASELINE_THREADS", "32")))
    torch.set_num_threads(ncores)
    torch.manual_seed(2)
    np.random.seed(2)
    ref = CpuStepRef(arch=arch, p_drop=0.1)
    gen = torch.Generator().manual_seed(2)
    data = []
    for _ in range(warmup + steps):
        il, iu = torch.randn(batch, 3, crop, crop, generator=gen), torch.randn(batch, 3, crop, crop, generator=gen)
        gsz = crop // 16 + 1
        coarse = torch.randint(0, 19, (batch, gsz, gsz), generator=gen)
        iy = (torch.arange(crop) * gsz // crop).clamp(max=gsz - 1)
        ll = coarse[:, iy][:, :, iy].contiguous()
        ll[:, :8] = 255
        data.append((il, ll, iu))
    times = []
    for i, (il, ll, iu) in enumerate(data):
        t0 = time.perf_counter()
        ref.step(il, ll, iu, epoch=0)
        times.append(time.perf_counter() - t0)
    dt = float(np.mean(times[warmup:]))
    return {"value": round(2 * batch / dt, 5), "unit": "images/s", "cores": ncores, "kind": "port",
            "s_per_step": [round(t, 2) for t in times],
            "sample": f"{warmup} warm-up + {steps} timed full U2PL steps (teacher eval fwd, student fwd+bwd, teacher train fwd, "
                      f"OHEM+unsup+contrastive losses, SGD, EMA) of oracle/step_ref.py (pinned to the reference's train()) on "
                      f"{batch} labeled + {batch} unlabeled {crop}x{crop} crops ({arch}), torch-CPU fp32 with {ncores} threads "
                      f"+ numpy; {dt:.1f} s per timed step"}


def validate_ref(model, batches, num_classes, ignore=255):
    """CPU restatement of validate() (train_semi.py:595-654, utils.py:568-580): bilinear up, argmax,
    intersection / union histograms, mIoU = mean(I / (U + 1e-10))."""
    model.eval()
    inter = np.zeros(num_classes)
    union = np.zeros(num_classes)
    with torch.no_grad():
        for images, labels in batches:
            out = _up(model(images)["pred"], labels.shape[1:]).argmax(1).numpy()
            ai, au, _ = R.intersection_and_union(out, labels.numpy(), num_classes, ignore)
            inter += ai
            union += au
    iou = inter / (union + 1e-10)
    return float(iou.mean()), iou


def sliding_window_ref(model, image, classes, crop_h, crop_w, h, w, stride_rate=2 / 3):
    """CPU restatement of scale_crop_process (reference eval.py:184-224) for one (1,3,H,W) image."""
    ori_h, ori_w = image.shape[-2:]
    pad_h, pad_w = max(crop_h - ori_h, 0), max(crop_w - ori_w, 0)
    ph, pw = int(pad_h / 2), int(pad_w / 2)
    if pad_h > 0 or pad_w > 0:
        image = F.pad(image, (pw, pad_w - pw, ph, pad_h - ph), mode="constant", value=0.0)
    new_h, new_w = image.shape[-2:]
    stride_h, stride_w = int(np.ceil(crop_h * stride_rate)), int(np.ceil(crop_w * stride_rate))
    grid_h = int(np.ceil(float(new_h - crop_h) / stride_h) + 1)
    grid_w = int(np.ceil(float(new_w - crop_w) / stride_w) + 1)
    pred = torch.zeros((1, classes, new_h, new_w))
    cnt = torch.zeros((new_h, new_w))
    model.eval()
    with torch.no_grad():
        for ih in range(grid_h):
            for iw in range(grid_w):
                e_h = min(ih * stride_h + crop_h, new_h)
                e_w = min(iw * stride_w + crop_w, new_w)
                s_h, s_w = e_h - crop_h, e_w - crop_w
                cropped = image[:, :, s_h:e_h, s_w:e_w].contiguous()
                cnt[s_h:e_h, s_w:e_w] += 1
                pred[:, :, s_h:e_h, s_w:e_w] += _up(model(cropped)["pred"], (crop_h, crop_w))
    pred /= cnt
    pred = pred[:, :, ph:ph + ori_h, pw:pw + ori_w]
    return _up(pred, (h, w))[0]
