"""Oracle clip encoders (torch CPU, fp32).  Test infrastructure -- see oracle/__init__.py.

Table-driven restatements of the three conv stacks on the hot path.  Attribute
names follow the reference so its ``state_dict()`` loads by key:

* ``R2Plus1D``  -- lib/modeling/backbone/backbone_3d/resnet2p1d.py:39-87 (BasicBlock),
  :90-136 (Bottleneck), :139-265 (ResNet), :268-285 (generate_model).
* ``S3D``       -- lib/modeling/backbone/backbone_3d/s3d_1.py:5-35 (net), :37-69
  (BasicConv3d / SepConv3d), :71-329 (Mixed_3b..5c).
* ``R3D``       -- lib/modeling/backbone/backbone_3d/resnet.py:39-106 (blocks), :109-191.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _c3(cin, cout, k, s, p):
    return nn.Conv3d(cin, cout, kernel_size=k, stride=s, padding=p, bias=False)


def _mid(cin, cout, kt=3, ks=3):
    """(2+1)D mid-plane count: resnet2p1d.py:45-47 (same formula :156-158 for the stem)."""
    return (cin * cout * kt * ks * ks) // (cin * ks * ks + kt * cout)


# --------------------------------------------------------------------------- R(2+1)D
class R2BasicBlock(nn.Module):
    """resnet2p1d.py:39-87.  Note conv1_t gets the SAME `stride` as conv1_s (:48,50):
    space is strided by conv1_s, time by conv1_t."""
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        m1 = _mid(cin, planes)
        self.conv1_s = _c3(cin, m1, (1, 3, 3), (1, stride, stride), (0, 1, 1))
        self.bn1_s = nn.BatchNorm3d(m1)
        self.conv1_t = _c3(m1, planes, (3, 1, 1), (stride, 1, 1), (1, 0, 0))
        self.bn1_t = nn.BatchNorm3d(planes)
        m2 = _mid(planes, planes)
        self.conv2_s = _c3(planes, m2, (1, 3, 3), 1, (0, 1, 1))
        self.bn2_s = nn.BatchNorm3d(m2)
        self.conv2_t = _c3(m2, planes, (3, 1, 1), 1, (1, 0, 0))
        self.bn2_t = nn.BatchNorm3d(planes)
        self.downsample = downsample

    def forward(self, x):
        o = F.relu(self.bn1_s(self.conv1_s(x)))
        o = F.relu(self.bn1_t(self.conv1_t(o)))
        o = F.relu(self.bn2_s(self.conv2_s(o)))
        o = self.bn2_t(self.conv2_t(o))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(o + r)


class R2Bottleneck(nn.Module):
    """resnet2p1d.py:90-136."""
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(cin, planes, 1, 1, 0)
        self.bn1 = nn.BatchNorm3d(planes)
        m = _mid(planes, planes)
        self.conv2_s = _c3(planes, m, (1, 3, 3), (1, stride, stride), (0, 1, 1))
        self.bn2_s = nn.BatchNorm3d(m)
        self.conv2_t = _c3(m, planes, (3, 1, 1), (stride, 1, 1), (1, 0, 0))
        self.bn2_t = nn.BatchNorm3d(planes)
        self.conv3 = _c3(planes, planes * 4, 1, 1, 0)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        o = F.relu(self.bn1(self.conv1(x)))
        o = F.relu(self.bn2_s(self.conv2_s(o)))
        o = F.relu(self.bn2_t(self.conv2_t(o)))
        o = self.bn3(self.conv3(o))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(o + r)


_R2_DEPTHS = {10: (R2BasicBlock, (1, 1, 1, 1)), 18: (R2BasicBlock, (2, 2, 2, 2)),
              34: (R2BasicBlock, (3, 4, 6, 3)), 50: (R2Bottleneck, (3, 4, 6, 3)),
              101: (R2Bottleneck, (3, 4, 23, 3)), 152: (R2Bottleneck, (3, 8, 36, 3)),
              200: (R2Bottleneck, (3, 24, 36, 3))}


def _kaiming_fan_out_(mod):
    """resnet2p1d.py:200-207 / resnet.py:142-147."""
    for m in mod.modules():
        if isinstance(m, nn.Conv3d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm3d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class R2Plus1D(nn.Module):
    """resnet2p1d.py:139-265 with shortcut type B only (the default, :147)."""

    def __init__(self, depth=18, widen_factor=1.0, n_classes=400, conv1_t_size=7):
        super().__init__()
        block, reps = _R2_DEPTHS[depth]
        widths = [int(w * widen_factor) for w in (64, 128, 256, 512)]
        self.in_planes = widths[0]
        m = (3 * self.in_planes * conv1_t_size * 49) // (3 * 49 + conv1_t_size * self.in_planes)
        self.conv1_s = _c3(3, m, (1, 7, 7), (1, 2, 2), (0, 3, 3))
        self.bn1_s = nn.BatchNorm3d(m)
        self.conv1_t = _c3(m, self.in_planes, (conv1_t_size, 1, 1), 1, (conv1_t_size // 2, 0, 0))
        self.bn1_t = nn.BatchNorm3d(self.in_planes)
        self.maxpool = nn.MaxPool3d(kernel_size=3, stride=2, padding=1)
        for i, (w, r) in enumerate(zip(widths, reps)):
            setattr(self, 'layer%d' % (i + 1), self._stage(block, w, r, 1 if i == 0 else 2))
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Linear(widths[3] * block.expansion, n_classes)
        _kaiming_fan_out_(self)

    def _stage(self, block, planes, reps, stride):
        ds = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            ds = nn.Sequential(_c3(self.in_planes, planes * block.expansion, 1, stride, 0),
                               nn.BatchNorm3d(planes * block.expansion))
        mods = [block(self.in_planes, planes, stride, ds)]
        self.in_planes = planes * block.expansion
        mods += [block(self.in_planes, planes) for _ in range(1, reps)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = F.relu(self.bn1_s(self.conv1_s(x)))
        x = F.relu(self.bn1_t(self.conv1_t(x)))
        x = self.maxpool(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x).flatten(1)
        return self.fc(x)


# --------------------------------------------------------------------------- S3D
class S3DUnit(nn.Module):
    """BasicConv3d, s3d_1.py:37-48: conv(no bias) -> BN(eps 1e-3, momentum 1e-3) -> ReLU."""

    def __init__(self, cin, cout, k, s, p=0):
        super().__init__()
        self.conv = _c3(cin, cout, k, s, p)
        self.bn = nn.BatchNorm3d(cout, eps=1e-3, momentum=0.001, affine=True)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class S3DSep(nn.Module):
    """SepConv3d, s3d_1.py:50-69: (1,k,k) then (k,1,1), each conv -> BN -> ReLU."""

    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        self.conv_s = _c3(cin, cout, (1, k, k), (1, s, s), (0, p, p))
        self.bn_s = nn.BatchNorm3d(cout, eps=1e-3, momentum=0.001, affine=True)
        self.conv_t = _c3(cout, cout, (k, 1, 1), (s, 1, 1), (p, 0, 0))
        self.bn_t = nn.BatchNorm3d(cout, eps=1e-3, momentum=0.001, affine=True)

    def forward(self, x):
        x = F.relu(self.bn_s(self.conv_s(x)))
        return F.relu(self.bn_t(self.conv_t(x)))


class S3DMixed(nn.Module):
    """One Inception block, s3d_1.py:71-99 (same shape for all nine)."""

    def __init__(self, cin, b0, b1a, b1b, b2a, b2b, b3):
        super().__init__()
        self.branch0 = nn.Sequential(S3DUnit(cin, b0, 1, 1))
        self.branch1 = nn.Sequential(S3DUnit(cin, b1a, 1, 1), S3DSep(b1a, b1b, 3, 1, 1))
        self.branch2 = nn.Sequential(S3DUnit(cin, b2a, 1, 1), S3DSep(b2a, b2b, 3, 1, 1))
        self.branch3 = nn.Sequential(nn.MaxPool3d(kernel_size=(3, 3, 3), stride=1, padding=1),
                                     S3DUnit(cin, b3, 1, 1))

    def forward(self, x):
        return torch.cat((self.branch0(x), self.branch1(x), self.branch2(x), self.branch3(x)), 1)


# (cin, b0, b1a, b1b, b2a, b2b, b3) for Mixed_3b .. Mixed_5c -- s3d_1.py:71-329
S3D_MIXED = {
    '3b': (192, 64, 96, 128, 16, 32, 32), '3c': (256, 128, 128, 192, 32, 96, 64),
    '4b': (480, 192, 96, 208, 16, 48, 64), '4c': (512, 160, 112, 224, 24, 64, 64),
    '4d': (512, 128, 128, 256, 24, 64, 64), '4e': (512, 112, 144, 288, 32, 64, 64),
    '4f': (528, 256, 160, 320, 32, 128, 128), '5b': (832, 256, 160, 320, 32, 128, 128),
    '5c': (832, 384, 192, 384, 48, 128, 128)}


class S3D(nn.Module):
    """s3d_1.py:5-35."""

    def __init__(self, num_class=400):
        super().__init__()
        mp = nn.MaxPool3d
        M = lambda k: S3DMixed(*S3D_MIXED[k])
        self.base = nn.Sequential(
            S3DSep(3, 64, 7, 2, 3),
            mp((1, 3, 3), (1, 2, 2), (0, 1, 1)),
            S3DUnit(64, 64, 1, 1),
            S3DSep(64, 192, 3, 1, 1),
            mp((1, 3, 3), (1, 2, 2), (0, 1, 1)),
            M('3b'), M('3c'),
            mp((3, 3, 3), (2, 2, 2), (1, 1, 1)),
            M('4b'), M('4c'), M('4d'), M('4e'), M('4f'),
            mp((2, 2, 2), (2, 2, 2), (0, 0, 0)),
            M('5b'), M('5c'))
        self.fc = nn.Sequential(nn.Conv3d(1024, num_class, kernel_size=1, stride=1, bias=True))

    def forward(self, x):
        y = self.base(x)
        y = F.avg_pool3d(y, (2, y.size(3), y.size(4)), stride=1)   # s3d_1.py:30
        y = self.fc(y)
        y = y.view(y.size(0), y.size(1), y.size(2))
        return y.mean(2)                                            # s3d_1.py:33


# --------------------------------------------------------------------------- 3D-ResNet
class R3BasicBlock(nn.Module):
    """resnet.py:39-67."""
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(cin, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = _c3(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm3d(planes)
        self.downsample = downsample

    def forward(self, x):
        o = F.relu(self.bn1(self.conv1(x)))
        o = self.bn2(self.conv2(o))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(o + r)


class R3Bottleneck(nn.Module):
    """resnet.py:70-106."""
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(cin, planes, 1, 1, 0)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = _c3(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = _c3(planes, planes * 4, 1, 1, 0)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        o = F.relu(self.bn1(self.conv1(x)))
        o = F.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(o + r)


_R3_DEPTHS = {10: (R3BasicBlock, (1, 1, 1, 1)), 18: (R3BasicBlock, (2, 2, 2, 2)),
              34: (R3BasicBlock, (3, 4, 6, 3)), 50: (R3Bottleneck, (3, 4, 6, 3)),
              101: (R3Bottleneck, (3, 4, 23, 3))}


class R3D(nn.Module):
    """resnet.py:109-191 (shortcut B).  The fixed AvgPool3d window is sized from
    sample_size/sample_duration (:137-140)."""

    def __init__(self, depth=50, sample_size=224, sample_duration=32, num_classes=400, width=64):
        super().__init__()
        block, reps = _R3_DEPTHS[depth]
        self.inplanes = width
        self.conv1 = _c3(3, width, 7, (1, 2, 2), (3, 3, 3))
        self.bn1 = nn.BatchNorm3d(width)
        self.maxpool = nn.MaxPool3d(kernel_size=(3, 3, 3), stride=2, padding=1)
        for i, r in enumerate(reps):
            setattr(self, 'layer%d' % (i + 1),
                    self._stage(block, width * 2 ** i, r, 1 if i == 0 else 2))
        ld, ls = int(math.ceil(sample_duration / 16)), int(math.ceil(sample_size / 32))
        self.avgpool = nn.AvgPool3d((ld, ls, ls), stride=1)
        self.fc = nn.Linear(width * 8 * block.expansion, num_classes)
        _kaiming_fan_out_(self)   # kaiming_normal(mode='fan_out'), gain sqrt(2) (:144)

    def _stage(self, block, planes, reps, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(_c3(self.inplanes, planes * block.expansion, 1, stride, 0),
                               nn.BatchNorm3d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, reps)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.fc(x.view(x.size(0), -1))


# Registry used by oracle.wrappers (the reference resolves names with getattr on the
# backbone_3d package, visual_wrappers.py:130-135; only 'S3D' is registered there --
# R2P1D*/R3D* are the unregistered-but-present files).
BACKBONES = {
    'S3D': lambda: S3D(),
    'R2P1D10': lambda: R2Plus1D(10),
    'R2P1D18': lambda: R2Plus1D(18),
    'R2P1D34': lambda: R2Plus1D(34),
    'R3D18': lambda: R3D(18, 112, 16),
    'R3D50': lambda: R3D(50, 224, 32),
}
