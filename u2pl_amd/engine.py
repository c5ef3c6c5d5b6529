"""Epoch-level driver shared by train_semi.py / train_sup.py: process-group bootstrap, model +
trainer construction from a reference YAML, validate() on device, checkpoint wire format
(reference: train_semi.py:47-231,595-654; train_sup.py:42-174,254-311; dist_helper.py:13-46)."""
import logging
import os
import os.path as osp
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from . import hipops as H
from ._lib import call
from .models.model_helper import ModelBuilder
from .trainer import SemiTrainer, SupTrainer
from .utils.loss_helper import get_criterion


def setup_distributed(port=None):
    """env:// bootstrap (torchrun); backend "nccl" == RCCL on ROCm.  Single process works too."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", rank % max(torch.cuda.device_count(), 1)))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if port is not None:
            os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    return rank, world


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def get_logger(name="global"):
    lg = logging.getLogger(name)
    if not lg.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("[%(asctime)s %(levelname)s] %(message)s"))
        lg.addHandler(h)
        lg.setLevel(logging.INFO)
    return lg


@torch.no_grad()
def validate(model, loader, cfg, device):
    """mIoU = mean(I / (U + 1e-10)) with the reference's accumulation (train_semi.py:595-654)."""
    model.eval()
    C, ign = cfg["net"]["num_classes"], cfg["dataset"]["ignore_label"]
    hist = torch.zeros(3 * C, dtype=torch.int64, device=device)
    from . import nn as K
    for images, labels in loader:
        images, labels = images.to(device, non_blocking=True), labels.to(device, non_blocking=True).long().contiguous()
        with K.eval_invstd(model):
            out = model(images, need_aux=False, need_rep=False)["pred"]
        large = H.bilinear_up(out, labels.shape[1:])
        N, _, Hh, Ww = large.shape
        call("u2pl_confusion_hist_f32", large, labels, ign, N, C, Hh, Ww, hist)
    from .nn import dist_active
    if dist_active():
        from .nn import _all_reduce
        _all_reduce(hist, "validate_allreduce")      # one all-reduce at the end instead of three per batch
    hist = hist.cpu().double().reshape(3, C)
    inter, union = hist[0], hist[1] + hist[2] - hist[0]
    iou = (inter / (union + 1e-10)).numpy()
    return float(np.mean(iou)), iou


def state_dict_ddp(model):
    """reference checkpoints carry DDP's 'module.' prefix (train_semi.py:210-224)."""
    return {"module." + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}


def load_state(path, model, optimizer=None, key="model_state"):
    """utils.py:583-636 semantics: strips 'module.', drops size-mismatched keys, strict=False; with `optimizer`
    (anything exposing load_state_dict / load_optimizer_state_dict) also restores `optimizer_state` and returns
    (best_miou, epoch) like the reference."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt[key] if key in ckpt else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    own = model.state_dict()
    sd = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
    model.load_state_dict(sd, strict=False)
    if optimizer is not None:
        load = getattr(optimizer, "load_optimizer_state_dict", None) or optimizer.load_state_dict
        load(ckpt["optimizer_state"])
        return ckpt["best_miou"], ckpt["epoch"]
    return ckpt


def _rng_state():
    """the four RNG streams of a step as tensors / plain scalars only (torch.load(weights_only=True) accepts the file)"""
    kind, keys, pos, has_gauss, cached = np.random.get_state()
    ver, pk, gauss = random.getstate()
    return {"torch": torch.get_rng_state(), "cuda": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
            "numpy_keys": torch.from_numpy(keys.astype(np.int64)), "numpy_pos": int(pos), "numpy_has_gauss": int(has_gauss),
            "numpy_cached": float(cached), "python_version": int(ver), "python_keys": torch.tensor(pk, dtype=torch.int64),
            "python_gauss": gauss}


def checkpoint_state(epoch, best, model, teacher, trainer):
    """train_semi.py:210-224 wire format (`module.`-prefixed model_state / teacher_state, torch-SGD optimizer_state,
    best_miou, epoch) + what upstream forgets and a bit-faithful resume needs: the memory bank, the iteration counter
    and the RNG streams (extra keys, ignored by the reference's load_state)."""
    state = {"epoch": epoch, "model_state": state_dict_ddp(model), "optimizer_state": trainer.optimizer_state_dict(),
             "best_miou": best, "cur_iter": trainer.cur_iter,
             "rng_state": _rng_state()}
    if teacher is not None:
        state["teacher_state"] = state_dict_ddp(teacher)
        bank = trainer.memobank
        state["memobank"] = {"rows": [bank.logical(c).cpu() for c in range(len(bank))], "ptr": list(bank.ptr)}
    return state


def restore_extras(ckpt, trainer, steps_per_epoch):
    trainer.cur_iter = ckpt.get("cur_iter", ckpt.get("epoch", 0) * steps_per_epoch)
    if "optimizer_state" in ckpt:
        trainer.load_optimizer_state_dict(ckpt["optimizer_state"])
    mb = ckpt.get("memobank")
    if mb is not None and getattr(trainer, "memobank", None) is not None:
        dev = trainer.memobank.buf[0].device
        for c, rows in enumerate(mb["rows"]):
            if rows.shape[0]:
                trainer.memobank.load_logical(c, rows.to(dev))
            trainer.memobank.ptr[c] = mb["ptr"][c]
    rs = ckpt.get("rng_state")
    if rs is not None:
        torch.set_rng_state(rs["torch"])
        if rs.get("cuda") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(rs["cuda"])
        np.random.set_state(("MT19937", rs["numpy_keys"].numpy().astype(np.uint32), rs["numpy_pos"], rs["numpy_has_gauss"],
                             rs["numpy_cached"]))
        random.setstate((rs["python_version"], tuple(int(x) for x in rs["python_keys"]), rs["python_gauss"]))


def absolutize_paths(cfg, exp_path):
    """reference configs hold paths relative to the experiment directory (train.sh cd's there); resolve them ONCE so
    that datasets opened lazily in DataLoader workers and the pretrain checkpoint do not depend on the cwd"""
    def fix(d, key):
        if isinstance(d.get(key), str) and d[key] and not osp.isabs(d[key]):
            d[key] = osp.normpath(osp.join(exp_path, d[key]))
    ds = cfg["dataset"]
    for sub in (ds, ds.get("train", {}), ds.get("val", {})):
        for key in ("data_root", "data_list"):
            fix(sub, key)
    if isinstance(cfg.get("saver", {}).get("pretrain"), str):
        fix(cfg["saver"], "pretrain")


def build(cfg, device, semi=True, steps_per_epoch=1):
    model = ModelBuilder(cfg["net"]).to(device)
    crit = get_criterion(cfg)
    if not semi:
        return model, None, SupTrainer(cfg, model, crit, steps_per_epoch)
    teacher = ModelBuilder(cfg["net"]).to(device)
    return model, teacher, SemiTrainer(cfg, model, teacher, crit, steps_per_epoch)


def run(cfg, args, semi):
    from .dataset import get_loader

    logger = get_logger()
    rank, world = setup_distributed(args.port)
    device = torch.device("cuda", torch.cuda.current_device())
    if args.seed is not None:
        set_random_seed(args.seed)
    cfg["exp_path"] = osp.dirname(osp.abspath(args.config))
    cfg["save_path"] = osp.join(cfg["exp_path"], cfg["saver"]["snapshot_dir"])
    if rank == 0:
        os.makedirs(cfg["save_path"], exist_ok=True)
    absolutize_paths(cfg, cfg["exp_path"])
    loaders = get_loader(cfg, seed=args.seed or 0)
    loader_l, loader_u, loader_val = (loaders if semi else (loaders[0], None, loaders[1]))
    model, teacher, trainer = build(cfg, device, semi, steps_per_epoch=len(loader_l))
    best, start_epoch = 0.0, 0
    ck = osp.join(cfg["save_path"], "ckpt.pth")
    if cfg["saver"].get("auto_resume", False) and osp.exists(ck):
        c = load_state(ck, model)      # parameters are views of the arenas: loaded in place
        if teacher is not None and "teacher_state" in c:
            load_state(ck, teacher, key="teacher_state")
        best, start_epoch = c.get("best_miou", 0.0), c.get("epoch", 0)
        restore_extras(c, trainer, len(loader_l))      # momentum (optimizer_state), bank, iteration counter, RNG
    elif cfg["saver"].get("pretrain", False):          # train_semi.py:152-154: student AND teacher
        c = load_state(cfg["saver"]["pretrain"], model)
        if teacher is not None and isinstance(c, dict) and "teacher_state" in c:
            load_state(cfg["saver"]["pretrain"], teacher, key="teacher_state")
    for epoch in range(start_epoch, cfg["trainer"]["epochs"]):
        for ld in (loader_l, loader_u):
            if ld is not None and hasattr(ld.sampler, "set_epoch"):
                ld.sampler.set_epoch(epoch)
        it_u = iter(loader_u) if loader_u is not None else None
        t0 = time.time()
        plan = getattr(loader_l, "device_plan", None)    # dataset.device_aug: raw uint8 batches, GPU transform chain
        for step, batch_l in enumerate(loader_l):
            if plan is not None:
                from .dataset.device_aug import augment_batch
                image_l, label_l = augment_batch(plan, batch_l[0].to(device, non_blocking=True),
                                                 batch_l[1].to(device, non_blocking=True), batch_l[2])
            else:
                image_l, label_l = batch_l[0].to(device, non_blocking=True), batch_l[1].to(device, non_blocking=True)
            if semi:
                batch_u = next(it_u)
                if plan is not None:
                    image_u, _ = augment_batch(plan, batch_u[0].to(device, non_blocking=True),
                                               batch_u[1].to(device, non_blocking=True), batch_u[2])
                else:
                    image_u = batch_u[0].to(device, non_blocking=True)
                meters = trainer.train_step(image_l, label_l, image_u, epoch)
            else:
                meters = trainer.train_step(image_l, label_l, epoch)
            i_iter = epoch * len(loader_l) + step
            if i_iter % 10 == 0 and rank == 0:
                m = [float(x) for x in meters.cpu()]
                logger.info("Iter [{}/{}] Time {:.2f}s/it Sup {:.3f} Uns {:.3f} Con {:.3f} LR {:.5f}".format(
                    i_iter, cfg["trainer"]["epochs"] * len(loader_l), (time.time() - t0) / (step + 1), m[0], m[1], m[2],
                    trainer.last_lr))
        if cfg["trainer"].get("eval_on", True):
            use_teacher = semi and epoch >= cfg["trainer"].get("sup_only_epoch", 1)
            miou, _ = validate(teacher if use_teacher else model, loader_val, cfg, device)
            if rank == 0:
                state = checkpoint_state(epoch + 1, max(best, miou), model, teacher, trainer)
                if miou > best:
                    best = miou
                    torch.save(state, osp.join(cfg["save_path"], "ckpt_best.pth"))
                torch.save(state, ck)
                logger.info("Epoch {} mIoU {:.2f} (best {:.2f})".format(epoch, miou * 100, best * 100))
    if dist.is_initialized():
        dist.destroy_process_group()
    return best
