"""TEST INFRASTRUCTURE ONLY -- plain torch.nn (CPU, fp32) restatement of the
reference network: deep-stem dilated ResNet encoder (u2pl/models/resnet.py:93-292),
ASPP (u2pl/models/base.py:11-100), DeepLabv3+ decoder with representation head and
auxiliary head (u2pl/models/decoder.py:45-142), assembled like ModelBuilder
(u2pl/models/model_helper.py:9-66).

Written table-driven (not a transcription): modules are created through small
factories so that parameter NAMES equal the reference's state_dict keys; pinned in
tests/test_oracle_golden.py against tests/golden/model_*.npz (outputs of the real
reference model) -- "parity pinned".  Used as the checker for the HIP model and as
the CPU baseline ("port") in bench.py.  Never imported by u2pl_amd.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def _c3(i, o, stride=1, d=1):
    return nn.Conv2d(i, o, 3, stride, d, d, bias=False)


def _c1(i, o, stride=1):
    return nn.Conv2d(i, o, 1, stride, bias=False)


class Neck(nn.Module):  # bottleneck residual unit
    def __init__(self, cin, planes, stride, dil, project):
        super().__init__()
        self.conv1, self.bn1 = _c1(cin, planes), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _c3(planes, planes, stride, dil), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _c1(planes, planes * 4), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = nn.Sequential(_c1(cin, planes * 4, stride), nn.BatchNorm2d(planes * 4)) if project else None

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class Encoder(nn.Module):
    def __init__(self, arch, dilate=(False, True, True), multi_grid=True):
        super().__init__()
        self.conv1 = nn.Sequential(_c3(3, 64, 2), nn.BatchNorm2d(64), nn.ReLU(True), _c3(64, 64), nn.BatchNorm2d(64),
                                   nn.ReLU(True), _c3(64, 128))
        self.bn1, self.relu = nn.BatchNorm2d(128), nn.ReLU(True)
        self.maxpool = nn.MaxPool2d(3, 2, 1, ceil_mode=True)
        cin, dil = 128, 1
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), DEPTHS[arch])):
            stride = 1 if li == 0 else 2
            prev = dil
            if li > 0 and dilate[li - 1]:
                dil, stride = dil * stride, 1
            grids = (2, 2, 4) if (li == 3 and multi_grid) else (1,) * n
            units = [Neck(cin, planes, stride, prev * grids[0], stride != 1 or cin != planes * 4)]
            cin = planes * 4
            units += [Neck(cin, planes, 1, dil * grids[i], False) for i in range(1, n)]
            setattr(self, f"layer{li + 1}", nn.Sequential(*units))

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        return [x1, x2, x3, self.layer4(x3)]


def _cbr(conv):
    return nn.Sequential(conv, nn.BatchNorm2d(conv.out_channels), nn.ReLU(True))


class Aspp(nn.Module):
    def __init__(self, cin, inner=256, rates=(12, 24, 36)):
        super().__init__()
        self.conv1 = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(cin, inner, 1, bias=False), nn.BatchNorm2d(inner),
                                   nn.ReLU(True))
        self.conv2 = _cbr(nn.Conv2d(cin, inner, 1, bias=False))
        for i, r in enumerate(rates):
            setattr(self, f"conv{i + 3}", _cbr(nn.Conv2d(cin, inner, 3, padding=r, dilation=r, bias=False)))

    def forward(self, x):
        h, w = x.shape[-2:]
        pooled = F.interpolate(self.conv1(x), size=(h, w), mode="bilinear", align_corners=True)
        return torch.cat([pooled, self.conv2(x), self.conv3(x), self.conv4(x), self.conv5(x)], 1)


def _tower(out, p_drop):
    return nn.Sequential(nn.Conv2d(512, 256, 3, 1, 1), nn.BatchNorm2d(256), nn.ReLU(True), nn.Dropout2d(p_drop),
                         nn.Conv2d(256, 256, 3, 1, 1), nn.BatchNorm2d(256), nn.ReLU(True), nn.Dropout2d(p_drop),
                         nn.Conv2d(256, out, 1))


class Decoder(nn.Module):
    def __init__(self, cin, num_classes, p_drop=0.1):
        super().__init__()
        self.low_conv = _cbr(nn.Conv2d(256, 256, 1))
        self.aspp = Aspp(cin)
        self.head = nn.Sequential(nn.Conv2d(1280, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(True),
                                  nn.Dropout2d(p_drop))
        self.classifier = _tower(num_classes, p_drop)
        self.representation = _tower(256, p_drop)

    def forward(self, feats):
        x1, _, _, x4 = feats
        low = self.low_conv(x1)
        a = F.interpolate(self.head(self.aspp(x4)), size=low.shape[-2:], mode="bilinear", align_corners=True)
        z = torch.cat([low, a], 1)
        return {"pred": self.classifier(z), "rep": self.representation(z)}


class AuxHead(nn.Module):
    def __init__(self, cin, num_classes, p_drop=0.1):
        super().__init__()
        self.aux = nn.Sequential(nn.Conv2d(cin, 256, 3, 1, 1), nn.BatchNorm2d(256), nn.ReLU(True), nn.Dropout2d(p_drop),
                                 nn.Conv2d(256, num_classes, 1))

    def forward(self, x):
        return self.aux(x)


class RefNet(nn.Module):
    """ModelBuilder-equivalent (fpn=True, rep_head=True).  p_drop=0 gives the parity mode
    (CPU mt19937 vs device RNG streams cannot match; SURVEY hard part 10)."""

    def __init__(self, arch="resnet101", num_classes=19, aux=True, p_drop=0.1):
        super().__init__()
        self.encoder = Encoder(arch)
        self.decoder = Decoder(2048, num_classes, p_drop)
        if aux:
            self.auxor = AuxHead(1024, num_classes, p_drop)

    def forward(self, x):
        feats = self.encoder(x)
        out = self.decoder(feats)
        if hasattr(self, "auxor"):
            out["aux"] = self.auxor(feats[2])
        return out
