"""The U2PL semi-supervised training step on one MI355X per rank.

This is the reference's hot loop ``train_semi.py:train()`` (lines 272-561)
re-expressed around device-resident state: every tensor op is a HIP kernel
(u2pl_amd.nn / u2pl_amd.hipops), the memory bank lives in HBM, entropy and the
three percentile thresholds are computed once and selected exactly on device,
parameters / gradients / momentum live in flat arenas (one SGD launch, one EMA
launch, one gradient all-reduce).  The only host synchronisation per step is the
read-back of ~60 list lengths that bound the reference's CPU ``torch.randint``
anchor / negative sampling (kept for bit-identical sampling).

Reference behaviours reproduced on purpose: label_onehot
batch-slot-0 quirk (Q0), class-index mismatch (Q1), bank update before
sampling (Q3), contrastive gradient scaled by an extra 1/world (Q5), teacher
aliasing in the first semi epoch (Q6), per-rank thresholds (Q7), LR set before
the forward (Q9), zero (not None) grads for unused heads (Q13).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import graphs as G
from . import hipops as H
from . import nn as K
from .utils import loss_helper as LH
from .utils.lr_helper import (adam_state_dict, check_adam_kwargs, check_sgd_kwargs, load_adam_state_dict, load_sgd_state_dict,
                              poly_lr, sgd_state_dict)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def generate_cutmix_boxes(B, im_h, im_w, ratio=2):
    """Host-side rectangle draws, same np.random call order as the reference
    (augmentation.py:471-485): w, x_start, y_start per sample."""
    boxes = []
    for _ in range(B):
        area = im_h * im_w / ratio
        w = np.random.randint(im_w / ratio + 1, im_w)
        h = np.round(area / w)
        x0 = np.random.randint(0, im_w - w + 1)
        y0 = np.random.randint(0, im_h - h + 1)
        boxes.append((int(y0), int(y0 + h), int(x0), int(x0 + w)))
    return boxes


def cutmix(image, label, conf, boxes):
    """generate_unsup_data(mode='cutmix') (augmentation.py:498-541) in one launch."""
    B, C, Hh, Ww = image.shape
    image = image.contiguous()
    bx = H.h2d(torch.tensor(boxes, dtype=torch.int32), image.device)
    oi, ol, oc = torch.empty_like(image), torch.empty_like(label), torch.empty_like(conf)
    K.call("u2pl_cutmix_f32", image, label, conf, bx, B, C, Hh, Ww, oi, ol, oc)
    return oi, ol, oc


def cutout(image, label, conf, boxes):
    """generate_unsup_data(mode='cutout') (augmentation.py:506-513): zero the box in image / confidence, label 255
    inside -- the only producer of ignored unlabeled pixels."""
    B, C, Hh, Ww = image.shape
    image = image.contiguous()
    bx = H.h2d(torch.tensor(boxes, dtype=torch.int32), image.device)
    oi, ol, oc = torch.empty_like(image), torch.empty_like(label), torch.empty_like(conf)
    K.call("u2pl_strong_aug_f32", image, label, conf, bx, None, 1, B, C, Hh, Ww, oi, ol, oc)
    return oi, ol, oc


def classmix_select(label, randperm=None):
    """generate_class_mask (augmentation.py:487-495) per image, in the reference's order: the sorted unique labels
    come back from the device as one presence bitmask per image (the one host sync of this mode), the random half
    is drawn with torch.randperm on the global CPU generator like upstream.  -> uint64 selection bitmasks (host)."""
    B = label.shape[0]
    bits = torch.zeros(B, dtype=torch.int64, device=label.device)
    K.call("u2pl_label_presence_i64", label.contiguous(), B, label[0].numel(), bits)
    present = bits.cpu().numpy().view(np.uint64)
    if randperm is None:
        randperm = torch.randperm
    sel = np.zeros(B, dtype=np.uint64)
    for i in range(B):
        labels = np.array([c for c in range(64) if (int(present[i]) >> c) & 1], dtype=np.int64)
        chosen = labels[randperm(len(labels)).numpy()][: len(labels) // 2]
        for c in chosen:
            sel[i] |= np.uint64(1) << np.uint64(c)
    return sel


def classmix(image, label, conf, sel):
    """generate_unsup_data(mode='classmix') (augmentation.py:517-535) given the selection bitmasks."""
    B, C, Hh, Ww = image.shape
    image = image.contiguous()
    sd = H.h2d(torch.from_numpy(sel.view(np.int64).copy()), image.device)
    oi, ol, oc = torch.empty_like(image), torch.empty_like(label), torch.empty_like(conf)
    K.call("u2pl_strong_aug_f32", image, label, conf, None, sd, 2, B, C, Hh, Ww, oi, ol, oc)
    return oi, ol, oc


class SemiTrainer:
    def __init__(self, cfg, model, model_teacher, sup_loss_fn, steps_per_epoch, memobank=None):
        self.cfg = cfg
        self.model, self.teacher = model, model_teacher
        self.sup_loss_fn = sup_loss_fn
        self.steps_per_epoch = steps_per_epoch
        tr = cfg["trainer"]
        self.epochs = tr["epochs"]
        self.sup_only_epoch = tr.get("sup_only_epoch", 1)
        self._init_schedule(tr)
        times = 10 if cfg["dataset"]["type"].startswith("pascal") else 1  # train_semi.py:100-110
        groups = [list(model.encoder.parameters()), list(model.decoder.parameters())]
        tgroups = [list(model_teacher.encoder.parameters()), list(model_teacher.decoder.parameters())]
        self.lr_mult = [1, times]
        if hasattr(model, "auxor"):
            groups.append(list(model.auxor.parameters()))
            tgroups.append(list(model_teacher.auxor.parameters()))
            self.lr_mult.append(times)
        self.arena = K.ParamArena(groups)
        self.t_arena = K.ParamArena(tgroups, with_grad=False)
        for p in model_teacher.parameters():
            p.requires_grad = False
        if K.dist_active() and os.environ.get("U2PL_TEACHER_COMM", "0") == "1":
            # OPT-IN (U2PL_TEACHER_COMM=1): the teacher's train-mode forward runs on the side HIP stream next to the student's
            # forward; its ~105 SyncBN all-reduces then get their own communicator so that they do not queue in front of the
            # student's.  Two communicators driven concurrently from one process are a documented NCCL / RCCL deadlock hazard
            # and this build has never run on RCCL: the DEFAULT is one communicator -- every collective of the step in one
            # host-issued order, identical on all ranks (U2PL_COMM_DEBUG=1 checks it), which cannot deadlock.
            K.use_process_group(model_teacher, dist.new_group())
        C = cfg["net"]["num_classes"]
        self.num_classes = C
        dev = next(model.parameters()).device
        if memobank is None:
            qs = [30000] * C
            qs[0] = 50000  # train_semi.py:161-169
            memobank = H.DeviceMemoryBank(C, qs, 256, dev)
        self.memobank = memobank
        self.cur_iter = 0
        self.use_aux = "aux_loss" in cfg["net"].keys()

    def _init_schedule(self, tr):
        ok = tr["optimizer"]
        self.opt_type = ok["type"]
        if ok["type"] == "SGD":            # lr_helper.py:18-19
            check_sgd_kwargs(ok["kwargs"])
            self.momentum = ok["kwargs"].get("momentum", 0.0)
        elif ok["type"] == "adam":         # lr_helper.py:20-21
            check_adam_kwargs(ok["kwargs"])
            self.adam = dict(betas=tuple(ok["kwargs"].get("betas", (0.9, 0.999))), eps=ok["kwargs"].get("eps", 1e-8),
                             weight_decay=ok["kwargs"].get("weight_decay", 0.0))
            self.momentum = 0.0
        else:
            raise AssertionError("optimizer type is not supported by LightSeg")      # the reference's own message (lr_helper.py:25)
        self.base_lr = ok["kwargs"].get("lr", 1e-3)
        self.weight_decay = ok["kwargs"].get("weight_decay", 0.0)
        sch = tr["lr_scheduler"]
        self.lr_mode = sch.get("mode", "poly")
        kw = sch.get("kwargs", {}) or {}
        if self.lr_mode == "poly":
            self.power = kw.get("power", 0.9) or 0.9
        elif self.lr_mode == "cosine":
            self.targetlr = kw["targetlr"]
        else:   # the reference accepts "multistep" but only implements "step" (lr_helper.py:47,84; Q9): neither is wired here
            raise NotImplementedError(f"lr_scheduler.mode {self.lr_mode!r}: poly and cosine are implemented")

    # -- LRScheduler.step (lr_helper.py:78-113): lr for this step is set before the forward
    def _lrs(self):
        max_iter = self.epochs * self.steps_per_epoch
        if self.lr_mode == "poly":
            lrs = [poly_lr(self.base_lr * m, self.cur_iter, max_iter, self.power) for m in self.lr_mult]
        else:
            from math import cos, pi
            lrs = [self.targetlr + (self.base_lr * m - self.targetlr) * (1 + cos(pi * self.cur_iter / max_iter)) / 2
                   for m in self.lr_mult]
        self.cur_iter += 1
        self.last_lr = lrs[0]
        self.last_lrs = list(lrs)      # per group (encoder, decoder[, aux]): the cosine schedule is affine, not linear, in base_lr * mult
        return lrs

    # -- torch.optim.SGD.state_dict() layout in the reference's group order (train_semi.py:100-110,214; utils.py:622-625)
    def _ref_groups(self):
        enc, dec = list(self.model.encoder.parameters()), list(self.model.decoder.parameters())
        aux = [list(self.model.auxor.parameters())] if hasattr(self.model, "auxor") else []
        return [enc] + aux + [dec]

    def optimizer_state_dict(self):
        m = self.lr_mult
        lrs = getattr(self, "last_lrs", None) or [self.base_lr * k for k in m]    # before the first step: the base lrs
        ref_lrs = [lrs[0]] + ([lrs[2]] if len(m) > 2 else []) + [lrs[1]]            # reference group order: encoder, aux, decoder
        if self.opt_type == "adam":
            return adam_state_dict(self._ref_groups(), ref_lrs, self.adam, self.arena.adam_views, self.arena.steps)
        return sgd_state_dict(self._ref_groups(), ref_lrs, self.momentum, self.weight_decay,
                              self.arena.momentum_view, self.arena.steps > 0)

    def load_optimizer_state_dict(self, sd):
        if self.opt_type == "adam":
            self.arena.steps = max(self.arena.steps, load_adam_state_dict(sd, self._ref_groups(), self.arena.adam_views))
        elif load_sgd_state_dict(sd, self._ref_groups(), self.arena.momentum_view):
            self.arena.steps = max(self.arena.steps, 1)

    def _optimizer_step(self, lrs, grad_scale):
        if self.opt_type == "adam":
            self.arena.adam_step(lrs, self.adam["betas"], self.adam["eps"], self.adam["weight_decay"], grad_scale=grad_scale)
        else:
            self.arena.sgd_step(lrs, self.momentum, self.weight_decay, grad_scale=grad_scale)

    # per-phase timing (bench.py `phase_ms`): when `phase_log` is a list, train_step appends (name, event) at every
    # phase boundary, recorded on the CURRENT stream.  Meaningful when the step is serialised on one stream (bench.py's extra
    # diagnostic step sets _side to the main stream and turns the weight-gradient side stream off); a no-op otherwise.
    phase_log = None

    def _mark(self, name):
        if self.phase_log is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phase_log.append((name, ev))

    # ---- the step's static segments, eager for the first calls and as HIP graphs from then on (u2pl_amd.graphs) ----
    def _teacher_eval_pass(self, image_u, hw):
        """pseudo labels (train_semi.py:317-324): eval-mode teacher, bilinear up, softmax max / arg-max"""
        with K.eval_invstd(self.teacher):
            pred_u_t = self.teacher(image_u, need_aux=False, need_rep=False)["pred"]
        return H.pseudo_label(H.bilinear_up(pred_u_t, hw))

    def _teacher_train_pass(self, image_all):
        """train-mode teacher forward (train_semi.py:360-374) + the class probabilities of its logits"""
        out_t = self.teacher(image_all, need_aux=False)
        pred_all_t, rep_all_t = out_t["pred"], out_t["rep"]
        prob_all_t = K.new_act(*pred_all_t.shape, pred_all_t.device)
        pt, ldp = K.as_rows(pred_all_t)
        Cn = pred_all_t.shape[1]
        K.call("u2pl_softmax_rows_f32", pt, ldp, prob_all_t, Cn, pt.shape[0] * pt.shape[2] * pt.shape[3], Cn)
        return pred_all_t, rep_all_t, prob_all_t

    def _graphed(self, which, arg):
        cache = self.__dict__.setdefault("_graph_cache", {})
        fn = cache.get((which, arg))
        if fn is None:
            if which == "teacher_eval":
                fn = G.GraphedNoGrad(lambda x, hw=arg: self._teacher_eval_pass(x, hw), [self.teacher], which)
            elif which == "teacher_train":
                fn = G.GraphedNoGrad(self._teacher_train_pass, [self.teacher], which,
                                     uniforms=lambda xs: K.dropout_uniforms_needed(self.teacher, xs[0].shape[0]))
            else:
                fn = G.GraphedTrain(self.model, which)
            cache[(which, arg)] = fn
        return fn

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream() if os.environ.get("U2PL_NO_SIDE_STREAM") is None else torch.cuda.current_stream()
        return self._side

    def _reduce_grads_and_step(self, lrs):
        K.wgrad_stream_sync()   # weight gradients are produced on a side stream
        W = _world()
        # bucketed all-reduce: most buckets were launched from the backward hooks and overlapped with it; this
        # launches the rest and joins them (DDP's mean is folded into the SGD launch)
        self.arena.finish_allreduce()
        self._optimizer_step(lrs, 1.0 / W)

    def train_step(self, image_l, label_l, image_u, epoch, cutmix_boxes=None, randint=None, debug=None):
        cfg = self.cfg
        model, teacher = self.model, self.teacher
        if K._lib.SIDE_WORK:
            # an exception that escaped the previous step between a ledger entry and its removal (a caller may catch it and
            # go on): join what the entries stand for and clear them, instead of leaving every later split on the
            # multi-launch path
            main_ = torch.cuda.current_stream()
            side_ = self._side_stream()
            if side_ is not main_:
                main_.wait_stream(side_)
            K.wgrad_stream_sync()
            if "buckets" in K._lib.SIDE_WORK:      # in-flight gradient all-reduces: wait for them before forgetting them
                for w_ in getattr(self.arena, "_works", ()):
                    if w_ is not None:
                        w_.wait()
            K._lib.SIDE_WORK.clear()
        B, h, w = label_l.shape
        lrs = self._lrs()
        i_iter = self.cur_iter - 1
        model.train()
        self.arena.zero_grad()
        label_l = label_l.long().contiguous()
        self._mark("start")
        if epoch < self.sup_only_epoch:  # train_semi.py:288-307
            outs = model(image_l)
            pred = H.bilinear_up(outs["pred"], (h, w))
            if self.use_aux:
                aux = H.bilinear_up(outs["aux"], (h, w))
                sup_loss = self.sup_loss_fn([pred, aux], label_l)
            else:
                sup_loss = self.sup_loss_fn(pred, label_l)
            teacher.train()
            with torch.no_grad():
                teacher(image_l)
            unsup_loss = H.zero_times_sum(outs["rep"])
            contra_loss = H.zero_times_sum(outs["rep"])
        else:
            if epoch == self.sup_only_epoch:  # Q6: teacher params alias the student's
                self.t_arena.copy_from(self.arena)
            unsup_cfg = cfg["trainer"]["unsupervised"]
            main = torch.cuda.current_stream()
            side = self._side_stream()
            side.wait_stream(main)
            if side is not main:
                K._lib.SIDE_WORK.add("teacher")
            # Both teacher passes run on a SIDE HIP stream, concurrently with the student forward (they do not
            # depend on it): their memory-bound BN passes and kernel tails fill the bubbles of the student's
            # MFMA-bound convs.  Results and RNG draw order are unchanged.
            # (1) pseudo labels (train_semi.py:317-324), eval mode
            teacher.eval()
            with torch.cuda.stream(side), torch.no_grad():
                conf_u, label_u_aug = self._graphed("teacher_eval", (h, w))(image_u)
            # strong augmentation (train_semi.py:326-337): host coin flip + host rectangle draws.  The IMAGE mix
            # needs only the boxes, so it is issued on the main stream right away; labels are mixed on the side.
            image_u_aug = image_u
            aug = unsup_cfg.get("apply_aug", False)
            mixed_on_main = None
            if np.random.uniform(0, 1) < 0.5 and aug:
                if aug not in ("cutmix", "cutout", "classmix"):
                    raise ValueError(f"trainer.unsupervised.apply_aug: {aug!r} (cutout | cutmix | classmix)")
                if aug == "classmix":
                    # the mask depends on the pseudo labels: the student's input has to wait for the side stream
                    with torch.cuda.stream(side):
                        sel = classmix_select(label_u_aug)
                        image_u_aug, label_u_aug, conf_u = classmix(image_u, label_u_aug, conf_u, sel)
                        image_all = torch.cat((image_l, image_u_aug))
                    main.wait_stream(side)
                    if side is not main:
                        image_u_aug.record_stream(main)
                    mixed_on_main = image_u_aug
                else:
                    fn = cutmix if aug == "cutmix" else cutout
                    boxes = cutmix_boxes if cutmix_boxes is not None else generate_cutmix_boxes(B, h, w)
                    with torch.cuda.stream(side):
                        image_u_aug, label_u_aug, conf_u = fn(image_u, label_u_aug, conf_u, boxes)
                        image_all = torch.cat((image_l, image_u_aug))
                    # the student's input is rebuilt on the main stream from the same boxes (no cross-stream wait)
                    mixed_on_main = fn(image_u, label_u_aug.new_zeros(label_u_aug.shape), conf_u.new_zeros(conf_u.shape),
                                       boxes)[0]
            else:
                with torch.cuda.stream(side):
                    image_all = torch.cat((image_l, image_u_aug))
            self._mark("teacher_eval")       # pseudo-label pass + strong augmentation
            # (2) teacher train-mode forward (train_semi.py:360-374)
            teacher.train()
            with torch.cuda.stream(side), torch.no_grad():
                pred_all_t, rep_all_t, prob_all_t = self._graphed("teacher_train", None)(image_all)
            self._mark("teacher_train")
            image_all_s = torch.cat((image_l, mixed_on_main if mixed_on_main is not None else image_u))
            # student forward (train_semi.py:339-358)
            outs = self._graphed("student", None)(image_all_s)
            pred_all, rep_all = outs["pred"], outs["rep"]
            pred_l_large = H.bilinear_up(pred_all[:B], (h, w))
            pred_u_large = H.bilinear_up(pred_all[B:], (h, w))
            if self.use_aux:
                aux = H.bilinear_up(outs["aux"][:B], (h, w))
                sup_loss = self.sup_loss_fn([pred_l_large, aux], label_l.clone())
            else:
                sup_loss = self.sup_loss_fn(pred_l_large, label_l.clone())
            self._mark("student_fwd")        # incl. the supervised loss heads (bilinear up + OHEM / CE)
            main.wait_stream(side)
            K._lib.SIDE_WORK.discard("teacher")
            if side is not main:
                for t_ in (pred_all_t, rep_all_t, prob_all_t, label_u_aug, conf_u):
                    t_.record_stream(main)   # allocated on the side stream, consumed on the main stream
                for t_ in (image_u, image_l):
                    t_.record_stream(side)   # allocated on the main stream, read on the side stream
            with torch.no_grad():
                # one fused pass: bilinear up-sampling + entropy + valid count + select histogram, then ONE
                # exact selection for all three percentiles (drop_percent, alpha_t, 100 - alpha_t)
                drop_percent = unsup_cfg.get("drop_percent", 100)
                percent_unreliable = (100 - drop_percent) * (1 - epoch / self.epochs)
                drop_percent = 100 - percent_unreliable
                ccfg = cfg["trainer"].get("contrastive", False)
                percents = [float(drop_percent)]
                if ccfg:
                    alpha_t = ccfg["low_entropy_threshold"] * (1 - epoch / self.epochs)
                    percents += [float(alpha_t), float(100 - alpha_t)]
                neg_high = bool(ccfg.get("negative_high_entropy", True)) if ccfg else True
                if H.REPLAY is not None:
                    H.REPLAY["rel"] = (pred_all_t[B:], (h, w), label_l, label_u_aug, tuple(pred_all.shape[2:]), percents, neg_high)
                # ONE persistent launch: fused bilinear up-sampling + entropy, exact selection of all three
                # percentiles (drop_percent, alpha_t, 100 - alpha_t), target overwrite, masks, class bits
                rs = H.reliability_split(pred_all_t[B:], (h, w), label_l, label_u_aug, tuple(pred_all.shape[2:]), percents,
                                         negative_high_entropy=neg_high)
                ent, thr, target_u = rs["entropy"], rs["thr"], rs["target_u"]
                low_mask, high_mask, lbits = rs["low_mask"], rs["high_mask"], rs["lbits"]
            self._mark("reliability")
            unsup_loss = H.cross_entropy(pred_u_large, target_u, 255, unsup_weight=True,
                                         scale=float(unsup_cfg.get("loss_weight", 1)))
            if ccfg:
                _, contra_local = LH.contra_memobank_core(
                    rep_all, lbits, B, prob_all_t[:B], prob_all_t[B:], low_mask, high_mask, ccfg, self.memobank,
                    rep_all_t, randint=randint)
                H.check_split(rs)     # (the contrastive block just synchronised with the host: this read is free)
                # Q5: value = cross-rank mean, gradient = local / world
                contra_loss = contra_local * (float(ccfg.get("loss_weight", 1)) / _world())
            else:
                H.watch_split(rs)     # no host sync on this path: the error word is examined one step later (non-blocking)
                contra_loss = H.zero_times_sum(rep_all)
        if debug is not None and epoch >= self.sup_only_epoch:
            debug.update(label_u=label_u_aug, target_u=target_u, entropy=ent, thr=thr, pred_u_large=pred_u_large.detach())
            if ccfg:
                debug.update(low_mask=low_mask, high_mask=high_mask, lbits=lbits)
        self._mark("contrastive")            # unsupervised CE + memory-bank contrastive loss (forward)
        loss = sup_loss + unsup_loss + contra_loss
        loss.backward()
        if self.phase_log is not None:
            K.wgrad_stream_sync()            # (the weight-gradient side stream, if on, belongs to the backward phase)
        self._mark("bwd")
        self._reduce_grads_and_step(lrs)
        # teacher EMA (train_semi.py:531-548)
        if epoch >= self.sup_only_epoch:
            d = min(1 - 1 / (i_iter - self.steps_per_epoch * self.sup_only_epoch + 1), cfg["net"]["ema_decay"])
            if epoch == self.sup_only_epoch:
                self.t_arena.copy_from(self.arena)  # aliasing: t == s_new before the EMA line
            self.t_arena.ema_from(self.arena, d)
        self._mark("opt_ema")
        meters = torch.stack((sup_loss.detach(), unsup_loss.detach(), contra_loss.detach()))
        if K.dist_active():
            cv = contra_loss.detach().clone()
            K._all_reduce(cv, "loss_allreduce")        # contra value = cross-rank mean (train_semi.py:514-519)
            meters[2] = cv
            K._all_reduce(meters, "meter_allreduce")    # logged meters are cross-rank SUMS (train_semi.py:551-561)
            K.check_comm_sequence()     # (U2PL_COMM_DEBUG=1: every rank issued the same collectives in the same order)
        return meters


class SupTrainer:
    """train_sup.py:177-251: supervised-only step (student only, no teacher / bank)."""

    def __init__(self, cfg, model, sup_loss_fn, steps_per_epoch):
        self.cfg, self.model, self.sup_loss_fn, self.steps_per_epoch = cfg, model, sup_loss_fn, steps_per_epoch
        tr = cfg["trainer"]
        self.epochs = tr["epochs"]
        self._init_schedule(tr)
        times = 10 if cfg["dataset"]["type"].startswith("pascal") else 1
        groups = [list(model.encoder.parameters()), list(model.decoder.parameters())]
        self.lr_mult = [1, times]
        if hasattr(model, "auxor"):
            groups.append(list(model.auxor.parameters()))
            self.lr_mult.append(times)
        self.arena = K.ParamArena(groups)
        self.cur_iter, self.last_lr, self.last_lrs = 0, self.base_lr, None
        self.use_aux = "aux_loss" in cfg["net"].keys()

    _init_schedule = SemiTrainer._init_schedule
    _lrs = SemiTrainer._lrs
    _ref_groups = SemiTrainer._ref_groups
    optimizer_state_dict = SemiTrainer.optimizer_state_dict
    load_optimizer_state_dict = SemiTrainer.load_optimizer_state_dict
    _optimizer_step = SemiTrainer._optimizer_step

    def train_step(self, image, label, epoch=0):
        lrs = self._lrs()
        self.model.train()
        self.arena.zero_grad()
        h, w = label.shape[1:]
        label = label.long().contiguous()
        outs = self.model(image, need_rep=False)
        pred = H.bilinear_up(outs["pred"], (h, w))
        if self.use_aux:
            loss = self.sup_loss_fn([pred, H.bilinear_up(outs["aux"], (h, w))], label)
        else:
            loss = self.sup_loss_fn(pred, label)
        loss.backward()
        K.wgrad_stream_sync()
        W = _world()
        self.arena.finish_allreduce()
        self._optimizer_step(lrs, 1.0 / W)
        z = torch.zeros((), device=loss.device)
        meters = torch.stack((loss.detach(), z, z))
        if K.dist_active():
            K._all_reduce(meters, "meter_allreduce")
            K.check_comm_sequence()
        return meters
