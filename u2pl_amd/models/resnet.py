"""Dilated deep-stem ResNet encoder (reference: u2pl/models/resnet.py).  Same
constructor surface, module names and initialisation order as the reference so
configs (``net.encoder.type: u2pl.models.resnet.resnet101``) and checkpoints
drop in; the compute is HIP (u2pl_amd.nn)."""
import torch
import torch.nn as nn

from .. import nn as K
from .base import norm_layer_for

__all__ = ["ResNet", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152"]

model_urls = {k: f"/path/to/{k}.pth" for k in ["resnet18", "resnet34", "resnet50", "resnet101", "resnet152"]}


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    assert groups == 1
    return K.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return K.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or K.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = K.conv_bn(self.conv1, self.bn1, x, relu=True)
        identity = x if self.downsample is None else K.run_seq(self.downsample, x)
        return K.conv_bn(self.conv2, self.bn2, out, res=identity, relu=True)


class Bottleneck(nn.Module):
    """resnet.py:93-140: 1x1 -> 3x3(stride, dilation) -> 1x1, BN after each, residual add + ReLU
    (the add and both ReLUs are fused into the BatchNorm apply kernels)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=K.BatchNorm2d):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # (the block input has two consumers, conv1 and the residual add in bn3: conv1's data-gradient launch adds the residual
        # branch's gradient instead of autograd's elementwise add -- nn.residual_grad_link)
        if self.downsample is None:
            link = K.residual_grad_link(x)
        else:   # conv1 and the downsample conv both consume x; joined when the downsample's data gradient has a fused form
            link = K.grad_join(x, 2) if K.dgrad_fusable(self.downsample[0], x) and K.dgrad_fusable(self.conv1, x) else None
        out = K.conv_bn(self.conv1, self.bn1, x, relu=True, grad_link=link)
        out = K.conv_bn(self.conv2, self.bn2, out, relu=True)
        if self.downsample is None:
            return K.conv_bn(self.conv3, self.bn3, out, res=x, relu=True, res_link=link)
        # bn3 and the downsample BatchNorm are independent: under a process group their statistics share one all-reduce
        return K.conv_bn_res_pair(self.conv3, self.bn3, out, self.downsample[0], self.downsample[1], x, grad_link_b=link)


class ResNet(nn.Module):
    def __init__(self, block, layers, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=[False, False, False], sync_bn=False, multi_grid=False, fpn=False):
        super().__init__()
        norm_layer = norm_layer_for(sync_bn)
        self._norm_layer = norm_layer
        self.inplanes = 128
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(
                replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        self.fpn = fpn
        self.conv1 = nn.Sequential(
            conv3x3(3, 64, stride=2), norm_layer(64), nn.ReLU(inplace=True),
            conv3x3(64, 64), norm_layer(64), nn.ReLU(inplace=True),
            conv3x3(64, self.inplanes),
        )
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = K.MaxPool3x3s2Ceil()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2],
                                       multi_grid=multi_grid)
        # same order of RNG draws as the reference (resnet.py:209-224)
        for m in self.modules():
            if isinstance(m, K.Conv2d):
                w = torch.empty(m.weight.shape)
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                with torch.no_grad():
                    m.weight.copy_(w)
            elif isinstance(m, K.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def get_outplanes(self):
        return self.inplanes

    def get_auxplanes(self):
        return self.inplanes // 2

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False, multi_grid=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        grids = [1] * blocks
        if multi_grid:
            grids = [2, 2, 4]
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width,
                        previous_dilation * grids[0], norm_layer)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation * grids[i], norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        stem = list(self.conv1)
        x = K.run_seq(nn.Sequential(*stem[:-1]), x)
        x = K.conv_bn(stem[-1], self.bn1, x, relu=True)
        x = self.maxpool(x)
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        return [x1, x2, x3, x4] if self.fpn else [x3, x4]


def _build(name, block, layers, pretrained, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        state_dict = torch.load(model_urls[name])
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        print(f"[Info] Load ImageNet pretrain from '{model_urls[name]}'", "\nmissing_keys: ", missing,
              "\nunexpected_keys: ", unexpected)
    return model


def resnet18(pretrained=False, **kwargs):
    return _build("resnet18", BasicBlock, [2, 2, 2, 2], pretrained, **kwargs)


def resnet34(pretrained=False, **kwargs):
    return _build("resnet34", BasicBlock, [3, 4, 6, 3], pretrained, **kwargs)


def resnet50(pretrained=True, **kwargs):
    return _build("resnet50", Bottleneck, [3, 4, 6, 3], pretrained, **kwargs)


def resnet101(pretrained=True, **kwargs):
    return _build("resnet101", Bottleneck, [3, 4, 23, 3], pretrained, **kwargs)


def resnet152(pretrained=True, **kwargs):
    return _build("resnet152", Bottleneck, [3, 8, 36, 3], pretrained, **kwargs)
