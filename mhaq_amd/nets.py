"""Network definitions that give the fake-quant path the tensor shapes of the BASELINE
configs.  torchvision / pytorchcv are not available, so
the layouts are written out here with the SAME module names and child order as the nets the
reference wraps, because its wrapping rule depends on both (excluded_layers by dotted name;
signedness by the module preceding the conv in named_modules(), gdnsq_quant.py:123-141):

  resnet18        torchvision layout: conv1 bn1 relu maxpool layer{1..4}.{0,1}.{conv1 bn1 relu
                  conv2 bn2 [downsample.{0,1}]} avgpool fc      (configs exclude conv1, fc)
  resnet20_cifar  pytorchcv layout: features.init_block.conv, features.stage{1,2,3}.unit{1,2,3}
                  .body.conv{1,2}.conv, .identity_conv.conv, output
                  (configs exclude features.init_block.conv, output)
  rfdn            /root/reference/src/models/sr/rfdn/{rfdn,block}.py layout: fea_conv B1..B4 c
                  LR_conv upsampler   (configs exclude fea_conv, upsampler.0)

These are plain nn.Modules: no quantization logic lives here.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------- ResNet-18
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 2, 1)
        self.layer2 = self._make_layer(128, 2, 2)
        self.layer3 = self._make_layer(256, 2, 2)
        self.layer4 = self._make_layer(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(num_classes=1000):
    return ResNet18(num_classes)


# ----------------------------------------------------------------------------- ResNet-20 (CIFAR)
class ConvBlock(nn.Module):
    def __init__(self, cin, cout, k, stride, pad, activate=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, pad, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        if activate:
            self.activ = nn.ReLU(inplace=True)
        self.activate = activate

    def forward(self, x):
        x = self.bn(self.conv(x))
        return self.activ(x) if self.activate else x


class ResBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = ConvBlock(cin, cout, 3, stride, 1)
        self.conv2 = ConvBlock(cout, cout, 3, 1, 1, activate=False)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class ResUnit(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.resize_identity = (cin != cout) or (stride != 1)
        self.body = ResBlock(cin, cout, stride)
        if self.resize_identity:
            self.identity_conv = ConvBlock(cin, cout, 1, stride, 0, activate=False)
        self.activ = nn.ReLU(inplace=True)

    def forward(self, x):
        identity = self.identity_conv(x) if self.resize_identity else x
        return self.activ(self.body(x) + identity)


class CIFARResNet20(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        self.features = nn.Sequential()
        self.features.add_module("init_block", ConvBlock(3, 16, 3, 1, 1))
        cin = 16
        for i, cout in enumerate((16, 32, 64)):
            stage = nn.Sequential()
            for j in range(3):
                stride = 2 if (j == 0 and i != 0) else 1
                stage.add_module(f"unit{j + 1}", ResUnit(cin, cout, stride))
                cin = cout
            self.features.add_module(f"stage{i + 1}", stage)
        self.features.add_module("final_pool", nn.AvgPool2d(8, 1))
        self.output = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = self.features(x)
        return self.output(x.view(x.size(0), -1))


def resnet20_cifar(num_classes=10):
    return CIFARResNet20(num_classes)


# ----------------------------------------------------------------------------- RFDN
def _conv(cin, cout, k, stride=1, dilation=1, groups=1):
    pad = int((k - 1) / 2) * dilation
    return nn.Conv2d(cin, cout, k, stride, padding=pad, bias=True, dilation=dilation, groups=groups)


class ESA(nn.Module):
    def __init__(self, n_feats):
        super().__init__()
        f = n_feats // 4
        self.conv1 = nn.Conv2d(n_feats, f, 1)
        self.conv_f = nn.Conv2d(f, f, 1)
        self.conv_max = nn.Conv2d(f, f, 3, padding=1)
        self.conv2 = nn.Conv2d(f, f, 3, stride=2, padding=0)
        self.conv3 = nn.Conv2d(f, f, 3, padding=1)
        self.conv3_ = nn.Conv2d(f, f, 3, padding=1)
        self.conv4 = nn.Conv2d(f, n_feats, 1)
        self.sigmoid = nn.Sigmoid()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        squeezed = self.conv1(x)                                   # 1x1 channel squeeze
        pooled = F.max_pool2d(self.conv2(squeezed), kernel_size=7, stride=3)
        att = self.relu(self.conv_max(pooled))
        att = self.conv3_(self.relu(self.conv3(att)))
        att = F.interpolate(att, x.shape[2:], mode="bilinear", align_corners=False)
        gate = self.sigmoid(self.conv4(att + self.conv_f(squeezed)))
        return x * gate


class RFDB(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.dc = self.distilled_channels = in_channels // 2
        self.rc = self.remaining_channels = in_channels
        self.c1_d = _conv(in_channels, self.dc, 1)
        self.c1_r = _conv(in_channels, self.rc, 3)
        self.c2_d = _conv(self.remaining_channels, self.dc, 1)
        self.c2_r = _conv(self.remaining_channels, self.rc, 3)
        self.c3_d = _conv(self.remaining_channels, self.dc, 1)
        self.c3_r = _conv(self.remaining_channels, self.rc, 3)
        self.c4 = _conv(self.remaining_channels, self.dc, 3)
        self.act = nn.LeakyReLU(0.05, inplace=True)
        self.c5 = _conv(self.dc * 4, in_channels, 1)
        self.esa = ESA(in_channels)

    def forward(self, x):
        distilled, cur = [], x
        for d_conv, r_conv in ((self.c1_d, self.c1_r), (self.c2_d, self.c2_r), (self.c3_d, self.c3_r)):
            distilled.append(self.act(d_conv(cur)))          # 1x1 distillation branch (never wrapped)
            cur = self.act(r_conv(cur) + cur)                # 3x3 residual refinement (wrapped)
        distilled.append(self.act(self.c4(cur)))
        return self.esa(self.c5(torch.cat(distilled, dim=1)))


class RFDN(nn.Module):
    def __init__(self, in_nc=3, nf=50, num_modules=4, out_nc=3, upscale=4):
        super().__init__()
        self.fea_conv = _conv(in_nc, nf, 3)
        self.B1, self.B2, self.B3, self.B4 = (RFDB(nf) for _ in range(4))
        self.c = nn.Sequential(_conv(nf * num_modules, nf, 1), nn.LeakyReLU(0.05, inplace=True))
        self.LR_conv = _conv(nf, nf, 3)
        self.upsampler = nn.Sequential(_conv(nf, out_nc * upscale ** 2, 3), nn.PixelShuffle(upscale))

    def forward(self, x):
        fea = self.fea_conv(x)
        b1 = self.B1(fea)
        b2 = self.B2(b1)
        b3 = self.B3(b2)
        b4 = self.B4(b3)
        out_b = self.c(torch.cat([b1, b2, b3, b4], dim=1))
        return self.upsampler(self.LR_conv(out_b) + fea)


def rfdn(upscale=4):
    return RFDN(upscale=upscale)
