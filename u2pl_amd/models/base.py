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
        feat1 = K.upsample_bilinear(K.run_seq(nn.Sequential(*list(self.conv1)[1:]), pooled), (h, w))
        feat2 = K.run_seq(self.conv2, x)
        feat3 = K.run_seq(self.conv3, x)
        feat4 = K.run_seq(self.conv4, x)
        feat5 = K.run_seq(self.conv5, x)
        return K.cat_channels((feat1, feat2, feat3, feat4, feat5))
