"""Learning-rate schedule with the reference's semantics (u2pl/utils/lr_helper.py:42-113):
poly / cosine, stepped per iteration, LR for step k set before that step's forward."""
from math import cos, pi


def poly_lr(base_lr, cur_iter, max_iter, power=0.9):
    return base_lr * ((1 - float(cur_iter) / max_iter) ** power)


class LRScheduler(object):
    """Drives any object exposing ``param_groups`` (list of dicts with 'lr')."""

    def __init__(self, mode, lr_args, data_size, optimizer, num_epochs, start_epochs):
        assert mode in ["multistep", "poly", "cosine"]
        self.mode, self.optimizer, self.data_size = mode, optimizer, data_size
        self.cur_iter = start_epochs * data_size
        self.max_iter = num_epochs * data_size
        self.base_lr = [g["lr"] for g in optimizer.param_groups]
        self.cur_lr = list(self.base_lr)
        if mode == "poly":
            self.power = lr_args["power"] if lr_args.get("power", False) else 0.9
        if mode == "cosine":
            self.targetlr = lr_args["targetlr"]

    def step(self):
        self._step()
        for g, lr in zip(self.optimizer.param_groups, self.cur_lr):
            g["lr"] = lr
        self.cur_iter += 1

    def _step(self):
        if self.mode == "poly":
            self.cur_lr = [poly_lr(lr, self.cur_iter, self.max_iter, self.power) for lr in self.base_lr]
        elif self.mode == "cosine":
            self.cur_lr = [self.targetlr + (lr - self.targetlr) * (1 + cos(pi * self.cur_iter / self.max_iter)) / 2
                           for lr in self.base_lr]
        else:
            raise NotImplementedError  # "multistep" is accepted but unimplemented upstream too (Q9)

    def get_lr(self):
        return self.cur_lr


def get_scheduler(cfg_trainer, len_data, optimizer, start_epoch=0, use_iteration=False):
    epochs = cfg_trainer["epochs"] if not use_iteration else 1
    return LRScheduler(cfg_trainer["lr_scheduler"]["mode"], cfg_trainer["lr_scheduler"]["kwargs"], len_data,
                       optimizer, epochs, start_epoch)
