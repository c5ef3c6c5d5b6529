"""ModelBuilder -- the reference's plug-in API (u2pl/models/model_helper.py:9-66):
encoder / decoder classes are resolved from dotted paths in the YAML and receive
the same injected kwargs; ``forward(x)`` returns ``{"pred", "rep"[, "aux"]}``.
Dotted paths that start with ``u2pl.`` resolve to this package (``u2pl`` is a thin
alias package of ``u2pl_amd``), so reference configs work unchanged."""
import importlib

import torch
import torch.nn as nn

from .decoder import Aux_Module


class ModelBuilder(nn.Module):
    def __init__(self, net_cfg):
        super().__init__()
        self._sync_bn = net_cfg["sync_bn"]
        self._num_classes = net_cfg["num_classes"]
        self.encoder = self._build_encoder(net_cfg["encoder"])
        self.decoder = self._build_decoder(net_cfg["decoder"])
        self._use_auxloss = True if net_cfg.get("aux_loss", False) else False
        self.fpn = True if net_cfg["encoder"]["kwargs"].get("fpn", False) else False
        if self._use_auxloss:
            cfg_aux = net_cfg["aux_loss"]
            self.loss_weight = cfg_aux["loss_weight"]
            self.auxor = Aux_Module(cfg_aux["aux_plane"], self._num_classes, self._sync_bn)

    def _build_encoder(self, enc_cfg):
        enc_cfg["kwargs"].update({"sync_bn": self._sync_bn})  # mutates the cfg like the reference (Q11)
        return self._build_module(enc_cfg["type"], enc_cfg["kwargs"])

    def _build_decoder(self, dec_cfg):
        dec_cfg["kwargs"].update({"in_planes": self.encoder.get_outplanes(), "sync_bn": self._sync_bn,
                                  "num_classes": self._num_classes})
        return self._build_module(dec_cfg["type"], dec_cfg["kwargs"])

    def _build_module(self, mtype, kwargs):
        module_name, class_name = mtype.rsplit(".", 1)
        if module_name == "u2pl" or module_name.startswith("u2pl."):
            module_name = "u2pl_amd" + module_name[4:]
        module = importlib.import_module(module_name)
        return getattr(module, class_name)(**kwargs)

    def forward(self, x, need_aux=True, need_rep=True):
        """need_aux / need_rep let the training step skip heads whose outputs are never read
        (teacher passes, train_semi.py:319,363-364); defaults reproduce the reference."""
        if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if self._use_auxloss:
            if self.fpn:
                f1, f2, feat1, feat2 = self.encoder(x)
                outs = self._decode([f1, f2, feat1, feat2], need_rep)
            else:
                feat1, feat2 = self.encoder(x)
                outs = self.decoder(feat2)
            if need_aux:
                outs.update({"aux": self.auxor(feat1)})
            return outs
        feat = self.encoder(x)
        return self._decode(feat, need_rep) if self.fpn else self.decoder(feat)

    def _decode(self, feats, need_rep):
        try:
            return self.decoder(feats, need_rep=need_rep)
        except TypeError:
            return self.decoder(feats)
