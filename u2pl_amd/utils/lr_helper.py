"""Learning-rate schedule with the reference's semantics (u2pl/utils/lr_helper.py:42-113):
poly / cosine, stepped per iteration, LR for step k set before that step's forward."""
from math import cos, pi

import torch


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


# ------------------------------------------------------------------ optimizer seam (lr_helper.py:12-27)
_SGD_DEFAULTS = dict(momentum=0, dampening=0, weight_decay=0, nesterov=False, maximize=False, foreach=None,
                     differentiable=False, fused=None)


def check_sgd_kwargs(kw):
    """the flat-arena step implements plain torch.optim.SGD(lr, momentum, weight_decay); anything that would change the
    update rule is rejected instead of being silently ignored"""
    known = {"lr", "momentum", "weight_decay", "dampening", "nesterov"}
    extra = set(kw) - known
    if extra:
        raise ValueError(f"optimizer kwargs not supported by the HIP SGD step: {sorted(extra)}")
    if kw.get("nesterov", False) or kw.get("dampening", 0) not in (0, 0.0):
        raise NotImplementedError("nesterov / dampening are not implemented by u2pl_sgd_step_f32 (never set by the reference configs)")


def sgd_state_dict(ref_groups, lrs, momentum, weight_decay, momentum_of, stepped):
    """torch.optim.SGD.state_dict() layout (what train_semi.py:210-224 saves and utils.py:622-625 reloads):
    ref_groups: parameter lists in the REFERENCE's group order (encoder, [aux head], decoder: train_semi.py:100-110);
    momentum_of(p) -> that parameter's momentum buffer (a view of the momentum arena)."""
    state, groups, idx = {}, [], 0
    for g, lr in zip(ref_groups, lrs):
        ids = []
        for p in g:
            if stepped:
                state[idx] = {"momentum_buffer": momentum_of(p).detach().cpu().contiguous()}
            ids.append(idx)
            idx += 1
        groups.append(dict(_SGD_DEFAULTS, lr=lr, momentum=momentum, weight_decay=weight_decay, params=ids))
    return {"state": state, "param_groups": groups}


def load_sgd_state_dict(sd, ref_groups, momentum_of):
    """copies the momentum buffers back into the arena; returns True when at least one step had been taken"""
    params = [p for g in ref_groups for p in g]
    if sum(len(g["params"]) for g in sd["param_groups"]) != len(params):
        raise ValueError("optimizer_state has a different number of parameters than this model")
    order = [i for g in sd["param_groups"] for i in g["params"]]
    stepped = False
    for p, i in zip(params, order):
        st = sd["state"].get(i, sd["state"].get(str(i)))
        if st is not None and st.get("momentum_buffer") is not None:
            momentum_of(p).copy_(st["momentum_buffer"].to(momentum_of(p).device))
            stepped = True
    return stepped


class ArenaSGD:
    """`get_optimizer(params_list, cfg_optim)` of the reference (lr_helper.py:12-27) for `type: SGD`: the parameters
    of all groups move into ONE flat arena (u2pl_amd.nn.ParamArena; layer kernels accumulate their weight gradients
    straight into it) and step() is one launch.  torch.optim.SGD surface used by train_semi.py: param_groups (the
    LRScheduler writes g['lr']), zero_grad(), step(), state_dict() / load_state_dict() in torch's layout.  Under a
    process group step() also all-reduces the gradient arena (the layers bypass autograd's .grad accumulation, so a
    DistributedDataParallel wrapper would never see them)."""

    def __init__(self, params_list, lr, momentum=0.0, weight_decay=0.0, **kw):
        from .. import nn as K
        check_sgd_kwargs(dict(kw, lr=lr, momentum=momentum, weight_decay=weight_decay))
        params_list = list(params_list)
        if len(params_list) > 3:
            raise ValueError("u2pl_sgd_step_f32 has three learning-rate segments (encoder / aux head / decoder)")
        self.momentum, self.weight_decay = float(momentum), float(weight_decay)
        # like torch.optim: the caller's group dicts are mutated (generators -> lists, defaults filled in) and KEPT, so a
        # second optimizer built from the same dicts shares them -- the reference relies on that (the
        # scheduler drives `optimizer_start`, whose groups are `optimizer`'s groups)
        for g in params_list:
            g["params"] = list(g["params"])
            for k, v in dict(_SGD_DEFAULTS, lr=lr, momentum=momentum, weight_decay=weight_decay).items():
                g.setdefault(k, v)
        self.param_groups = params_list
        self.groups = [g["params"] for g in params_list]
        owner = getattr(self.groups[0][0], "_u2pl_owner", None)
        if owner is not None:            # second get_optimizer() over the same parameters: same arena, same momentum
            self.arena = owner.arena
        else:
            self.arena = K.ParamArena(self.groups)
            for g in self.groups:
                for p in g:
                    p._u2pl_owner = self

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def step(self):
        import torch.distributed as dist
        from .. import nn as K
        K.wgrad_stream_sync()
        W = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.arena.finish_allreduce()
        self.arena.sgd_step([g["lr"] for g in self.param_groups], self.momentum, self.weight_decay, grad_scale=1.0 / W)

    def _mom(self, p):
        return self.arena.momentum_view(p)

    def state_dict(self):
        return sgd_state_dict(self.groups, [g["lr"] for g in self.param_groups], self.momentum, self.weight_decay,
                              self._mom, self.arena.steps > 0)

    def load_state_dict(self, sd):
        if load_sgd_state_dict(sd, self.groups, self._mom):
            self.arena.steps = max(self.arena.steps, 1)
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g["lr"] = src["lr"]


_ADAM_DEFAULTS = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)


def check_adam_kwargs(kw):
    extra = set(kw) - {"lr", "betas", "eps", "weight_decay", "amsgrad"}
    if extra:
        raise ValueError(f"optimizer kwargs not supported by the HIP Adam step: {sorted(extra)}")
    if kw.get("amsgrad", False):
        raise NotImplementedError("amsgrad is not implemented by u2pl_adam_step_f32")


def adam_state_dict(ref_groups, lrs, hyper, views_of, steps):
    """torch.optim.Adam.state_dict() layout: state[i] = {step, exp_avg, exp_avg_sq}; groups in the reference's order"""
    state, groups, idx = {}, [], 0
    for g, lr in zip(ref_groups, lrs):
        ids = []
        for p in g:
            if steps > 0:
                m, v = views_of(p)
                state[idx] = {"step": torch.tensor(float(steps)), "exp_avg": m.detach().cpu().contiguous(),
                              "exp_avg_sq": v.detach().cpu().contiguous()}
            ids.append(idx)
            idx += 1
        groups.append(dict(_ADAM_DEFAULTS, lr=lr, betas=tuple(hyper["betas"]), eps=hyper["eps"], weight_decay=hyper["weight_decay"],
                           params=ids))
    return {"state": state, "param_groups": groups}


def load_adam_state_dict(sd, ref_groups, views_of):
    """copies exp_avg / exp_avg_sq back into the arenas; returns the step count stored in the file (0: never stepped)"""
    params = [p for g in ref_groups for p in g]
    if sum(len(g["params"]) for g in sd["param_groups"]) != len(params):
        raise ValueError("optimizer_state has a different number of parameters than this model")
    order = [i for g in sd["param_groups"] for i in g["params"]]
    steps = 0
    for p, i in zip(params, order):
        st = sd["state"].get(i, sd["state"].get(str(i)))
        if st is not None and st.get("exp_avg") is not None:
            m, v = views_of(p)
            m.copy_(st["exp_avg"].to(m.device))
            v.copy_(st["exp_avg_sq"].to(v.device))
            steps = max(steps, int(float(st["step"])))
    return steps


class ArenaAdam(ArenaSGD):
    """`get_optimizer` for `type: adam` (lr_helper.py:20-21): torch.optim.Adam's surface on the flat arena"""

    def __init__(self, params_list, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **kw):
        from .. import nn as K
        check_adam_kwargs(dict(kw, lr=lr))
        params_list = list(params_list)
        if len(params_list) > 3:
            raise ValueError("u2pl_adam_step_f32 has three learning-rate segments (encoder / aux head / decoder)")
        self.hyper = dict(betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay))
        for g in params_list:
            g["params"] = list(g["params"])
            for k, v in dict(_ADAM_DEFAULTS, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay).items():
                g.setdefault(k, v)
        self.param_groups = params_list
        self.groups = [g["params"] for g in params_list]
        owner = getattr(self.groups[0][0], "_u2pl_owner", None)
        if owner is not None:
            self.arena = owner.arena
        else:
            self.arena = K.ParamArena(self.groups)
            for g in self.groups:
                for p in g:
                    p._u2pl_owner = self

    def step(self):
        import torch.distributed as dist
        from .. import nn as K
        K.wgrad_stream_sync()
        W = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.arena.finish_allreduce()
        self.arena.adam_step([g["lr"] for g in self.param_groups], self.hyper["betas"], self.hyper["eps"],
                             self.hyper["weight_decay"], grad_scale=1.0 / W)

    def state_dict(self):
        return adam_state_dict(self.groups, [g["lr"] for g in self.param_groups], self.hyper, self.arena.adam_views, self.arena.steps)

    def load_state_dict(self, sd):
        self.arena.steps = max(self.arena.steps, load_adam_state_dict(sd, self.groups, self.arena.adam_views))
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g["lr"] = src["lr"]


def get_optimizer(parms, cfg_optim):
    """lr_helper.py:12-27: `SGD` and `adam` (anything else is the reference's "optimizer type is not supported" assert)"""
    if cfg_optim["type"] == "SGD":
        return ArenaSGD(parms, **cfg_optim["kwargs"])
    if cfg_optim["type"] == "adam":
        return ArenaAdam(parms, **cfg_optim["kwargs"])
    raise AssertionError("optimizer type is not supported by LightSeg")
