"""3D-ResNet (full 3x3x3 / 7x7x7 convolutions) on the HIP engine.

Architecture / names / init as the reference's lib/modeling/backbone/backbone_3d/resnet.py
(BasicBlock :39-67, Bottleneck :70-106, ResNet :109-191, resnet10..200 :208-261; shortcut 'B').
This is the only place a full 3x3x3 window occurs: the implicit-GEMM kernel gathers its 27 taps
through the same row table as the factorised convs.
"""
import math

import torch.nn as nn

from .....engine import layers as L
from .....engine.layers import HipBatchNorm3d, HipConv3d, HipLinear, HipMaxPool3d


def conv3x3x3(cin, cout, stride=1):
    return HipConv3d(cin, cout, 3, stride, 1)


class _Block(nn.Module):
    def _shortcut(self, tape, xv):
        if self.downsample is None:
            return xv
        return L.f_conv_bn_act(tape, self.downsample[0], self.downsample[1], xv, relu=False)


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3x3(inplanes, planes, stride)
        self.bn1 = HipBatchNorm3d(planes)
        self.relu = L.HipReLU()
        self.conv2 = conv3x3x3(planes, planes)
        self.bn2 = HipBatchNorm3d(planes)
        self.downsample = downsample
        self.stride = stride

    def fwd(self, tape, xv):
        o = L.f_conv_bn_act(tape, self.conv1, self.bn1, xv)
        return L.f_conv_bn_act(tape, self.conv2, self.bn2, o, relu=True, residual=self._shortcut(tape, xv))


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = HipConv3d(inplanes, planes, 1)
        self.bn1 = HipBatchNorm3d(planes)
        self.conv2 = HipConv3d(planes, planes, 3, stride, 1)
        self.bn2 = HipBatchNorm3d(planes)
        self.conv3 = HipConv3d(planes, planes * 4, 1)
        self.bn3 = HipBatchNorm3d(planes * 4)
        self.relu = L.HipReLU()
        self.downsample = downsample
        self.stride = stride

    def fwd(self, tape, xv):
        o = L.f_conv_bn_act(tape, self.conv1, self.bn1, xv)
        o = L.f_conv_bn_act(tape, self.conv2, self.bn2, o)
        return L.f_conv_bn_act(tape, self.conv3, self.bn3, o, relu=True, residual=self._shortcut(tape, xv))


class ResNet(nn.Module):
    def __init__(self, block, layers, sample_size, sample_duration, shortcut_type='B', num_classes=400):
        super().__init__()
        if shortcut_type != 'B':
            raise NotImplementedError('only shortcut type B is on the pre-training path')
        self.inplanes = 64
        self.conv1 = HipConv3d(3, 64, 7, (1, 2, 2), (3, 3, 3))
        self.bn1 = HipBatchNorm3d(64)
        self.relu = L.HipReLU()
        self.maxpool = HipMaxPool3d((3, 3, 3), 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        # AvgPool3d((ceil(T/16), ceil(S/32), ceil(S/32)), stride=1): sized to cover the final map (:137-140)
        self.pool_window = (int(math.ceil(sample_duration / 16)), int(math.ceil(sample_size / 32)),
                            int(math.ceil(sample_size / 32)))
        self.avgpool = L.HipIdentity()
        self.fc = HipLinear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, HipConv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')     # leaky_relu gain with a=0 == sqrt(2)
            elif isinstance(m, HipBatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(HipConv3d(self.inplanes, planes * block.expansion, 1, stride),
                               HipBatchNorm3d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def fwd(self, tape, xv):
        x = L.f_conv_bn_relu_maxpool(tape, self.conv1, self.bn1, self.maxpool, xv)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = L.f_seq(tape, stage, x)
        if tuple(x.t.shape[2:]) != self.pool_window:
            raise RuntimeError('final feature map %r != AvgPool3d window %r: construct the backbone with the '
                               'sample_size / sample_duration of the clips' % (tuple(x.t.shape[2:]), self.pool_window))
        x = L.f_wavgpool(tape, x)
        return L.f_head_fc(tape, self.fc, x)


def _make(block, layers):
    def ctor(**kwargs):
        return ResNet(block, layers, **kwargs)
    return ctor


resnet10 = _make(BasicBlock, [1, 1, 1, 1])
resnet18 = _make(BasicBlock, [2, 2, 2, 2])
resnet34 = _make(BasicBlock, [3, 4, 6, 3])
resnet50 = _make(Bottleneck, [3, 4, 6, 3])
resnet101 = _make(Bottleneck, [3, 4, 23, 3])
resnet152 = _make(Bottleneck, [3, 8, 36, 3])
resnet200 = _make(Bottleneck, [3, 24, 36, 3])
