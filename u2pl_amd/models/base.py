"""ASPP head and the norm-layer factory (reference: u2pl/models/base.py)."""
import torch.nn as nn

from .. import nn as K


def get_syncbn():
    """base.py:6-8 -- the reference returns nn.SyncBatchNorm; ours exchanges the
    per-channel sums with one RCCL all-reduce per layer (u2pl_amd.nn.SyncBatchNorm)."""
    return K.SyncBatchNorm


def norm_layer_for(sync_bn):
    return get_syncbn() if sync_bn else K.BatchNorm2d


class ASPP(nn.Module):
    """base.py:11-100: image pooling + 1x1 + three dilated 3x3 branches, concatenated."""

    def __init__(self, in_planes, inner_planes=256, sync_bn=False, dilations=(12, 24, 36)):
        super().__init__()
        norm_layer = norm_layer_for(sync_bn)
        self.conv1 = nn.Sequential(
            nn.AdaptiveAvgPool2d((1, 1)),  # marker: executed by K.global_avg_pool
            K.Conv2d(in_planes, inner_planes, kernel_size=1, padding=0, dilation=1, bias=False),
            norm_layer(inner_planes),
            nn.ReLU(inplace=True),
        )
        self.conv2 = nn.Sequential(
            K.Conv2d(in_planes, inner_planes, kernel_size=1, padding=0, dilation=1, bias=False),
            norm_layer(inner_planes),
            nn.ReLU(inplace=True),
        )

        def branch(d):
            return nn.Sequential(
                K.Conv2d(in_planes, inner_planes, kernel_size=3, padding=d, dilation=d, bias=False),
                norm_layer(inner_planes),
                nn.ReLU(inplace=True),
            )

        self.conv3 = branch(dilations[0])
        self.conv4 = branch(dilations[1])
        self.conv5 = branch(dilations[2])
        self.out_planes = (len(dilations) + 2) * inner_planes

    def get_outplanes(self):
        return self.out_planes

    def forward(self, x):
        _, _, h, w = x.size()
        pooled = K.global_avg_pool(x)
        # five independent conv -> BN -> ReLU branches: under a process group their SyncBatchNorm statistics are exchanged
        # in ONE packed all-reduce (forward and backward); otherwise exactly run_seq's fused conv_bn per branch
        # x has five consumers; the four convolutions share a gradient join (nn.GradJoin: their data-gradient launches add up the
        # contributions instead of autograd's elementwise adds over the 2048-channel tensor).  The branches whose data gradient
        # has no fused accumulate form (the direct 3x3, d = 24 at 97^2) are issued LAST, so that their backward runs first and
        # finds nothing to add yet.
        branches = [self.conv2, self.conv3, self.conv4, self.conv5]
        order = sorted(range(4), key=lambda i: 0 if K.dgrad_fusable(branches[i][0], x) else 1)
        join = K.grad_join(x, 4)
        units = [(self.conv1[1], self.conv1[2], pooled, True, None)]
        units += [(branches[i][0], branches[i][1], x, True, None, join) for i in order]
        f = K.conv_bn_group(units)
        outs = [None] * 4
        for k, i in enumerate(order):
            outs[i] = f[1 + k]
        return K.cat_channels((K.upsample_bilinear(f[0], (h, w)), outs[0], outs[1], outs[2], outs[3]))
