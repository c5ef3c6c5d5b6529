"""DeepLabv3 / DeepLabv3+ decoders and the auxiliary head (reference:
u2pl/models/decoder.py).  nn.Sequential containers keep the reference's
indices (``classifier.8.weight``); ReLU / Dropout2d entries are markers fused
into the BatchNorm apply kernel by u2pl_amd.nn.run_seq."""
import torch.nn as nn

from .. import nn as K
from .base import ASPP, norm_layer_for


class dec_deeplabv3(nn.Module):
    def __init__(self, in_planes, num_classes=19, inner_planes=256, sync_bn=False, dilations=(12, 24, 36)):
        super().__init__()
        norm_layer = norm_layer_for(sync_bn)
        self.aspp = ASPP(in_planes, inner_planes=inner_planes, sync_bn=sync_bn, dilations=dilations)
        self.head = nn.Sequential(
            K.Conv2d(self.aspp.get_outplanes(), 256, kernel_size=3, padding=1, dilation=1, bias=False),
            norm_layer(256), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
            K.Conv2d(256, num_classes, kernel_size=1, stride=1, padding=0, bias=True),
        )

    def forward(self, x):
        return K.run_seq(self.head, self.aspp(x))


class dec_deeplabv3_plus(nn.Module):
    def __init__(self, in_planes, num_classes=19, inner_planes=256, sync_bn=False, dilations=(12, 24, 36),
                 rep_head=True):
        super().__init__()
        norm_layer = norm_layer_for(sync_bn)
        self.rep_head = rep_head
        self.low_conv = nn.Sequential(K.Conv2d(256, 256, kernel_size=1), norm_layer(256), nn.ReLU(inplace=True))
        self.aspp = ASPP(in_planes, inner_planes=inner_planes, sync_bn=sync_bn, dilations=dilations)
        self.head = nn.Sequential(
            K.Conv2d(self.aspp.get_outplanes(), 256, kernel_size=3, padding=1, dilation=1, bias=False),
            norm_layer(256), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
        )

        def tower(out_planes):
            return nn.Sequential(
                K.Conv2d(512, 256, kernel_size=3, stride=1, padding=1, bias=True),
                norm_layer(256), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
                K.Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=True),
                norm_layer(256), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
                K.Conv2d(256, out_planes, kernel_size=1, stride=1, padding=0, bias=True),
            )

        self.classifier = tower(num_classes)
        if self.rep_head:
            self.representation = tower(256)

    def forward(self, x, need_rep=True):
        x1, x2, x3, x4 = x
        aspp_out = self.aspp(x4)
        low_feat = K.run_seq(self.low_conv, x1)
        aspp_out = K.run_seq(self.head, aspp_out)
        h, w = low_feat.size()[-2:]
        aspp_out = K.upsample_bilinear(aspp_out, (h, w))
        aspp_out = K.cat_channels((low_feat, aspp_out))
        # (no nn.GradJoin over the two towers: a caller may back-propagate through "pred" only -- the reference's supervised loop
        # does -- and a join needs every wired consumer's backward to run)
        res = {"pred": K.run_seq(self.classifier, aspp_out)}
        if self.rep_head and need_rep:
            res["rep"] = K.run_seq(self.representation, aspp_out)
        return res


class Aux_Module(nn.Module):
    def __init__(self, in_planes, num_classes=19, sync_bn=False):
        super().__init__()
        norm_layer = norm_layer_for(sync_bn)
        self.aux = nn.Sequential(
            K.Conv2d(in_planes, 256, kernel_size=3, stride=1, padding=1),
            norm_layer(256), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
            K.Conv2d(256, num_classes, kernel_size=1, stride=1, padding=0, bias=True),
        )

    def forward(self, x):
        return K.run_seq(self.aux, x)
