"""R(2+1)D clip encoder on the HIP engine.

Same architecture, parameter names and initialisation as the reference's
lib/modeling/backbone/backbone_3d/resnet2p1d.py (BasicBlock :39-87, Bottleneck :90-136,
ResNet :139-265, generate_model :268-285; shortcut type 'B'), so its checkpoints load by key
and the same seed gives the same weights.  The forward is written against the engine's fused
units: every conv -> BN -> ReLU (-> += residual) triple is one f_conv_bn_act, i.e. an MFMA
implicit-GEMM conv whose epilogue emits the BN batch statistics, plus one fused
normalise/ReLU/residual pass.
"""
import torch.nn as nn

from .....engine import layers as L
from .....engine.layers import HipBatchNorm3d, HipConv3d, HipLinear, HipMaxPool3d


def get_inplanes():
    return [64, 128, 256, 512]


def mid_planes(cin, cout, kt=3, ks=3):
    """Channel count of the (2+1)D factorisation that matches the full 3D conv's parameter budget."""
    return (cin * cout * kt * ks * ks) // (cin * ks * ks + kt * cout)


def conv1x3x3(cin, cout, stride=1):
    return HipConv3d(cin, cout, (1, 3, 3), (1, stride, stride), (0, 1, 1))


def conv3x1x1(cin, cout, stride=1):
    return HipConv3d(cin, cout, (3, 1, 1), (stride, 1, 1), (1, 0, 0))


def conv1x1x1(cin, cout, stride=1):
    return HipConv3d(cin, cout, 1, stride, 0)


class _Block(nn.Module):
    def _shortcut(self, tape, xv):
        if self.downsample is None:
            return xv
        return L.f_conv_bn_act(tape, self.downsample[0], self.downsample[1], xv, relu=False)


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        m1 = mid_planes(in_planes, planes)
        self.conv1_s = conv1x3x3(in_planes, m1, stride)
        self.bn1_s = HipBatchNorm3d(m1)
        self.conv1_t = conv3x1x1(m1, planes, stride)     # time is strided here, space above
        self.bn1_t = HipBatchNorm3d(planes)
        m2 = mid_planes(planes, planes)
        self.conv2_s = conv1x3x3(planes, m2)
        self.bn2_s = HipBatchNorm3d(m2)
        self.conv2_t = conv3x1x1(m2, planes)
        self.bn2_t = HipBatchNorm3d(planes)
        self.downsample = downsample
        self.stride = stride

    def fwd(self, tape, xv):
        o = L.f_conv_bn_act(tape, self.conv1_s, self.bn1_s, xv)
        o = L.f_conv_bn_act(tape, self.conv1_t, self.bn1_t, o)
        o = L.f_conv_bn_act(tape, self.conv2_s, self.bn2_s, o)
        r = self._shortcut(tape, xv)
        return L.f_conv_bn_act(tape, self.conv2_t, self.bn2_t, o, relu=True, residual=r)


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1x1(in_planes, planes)
        self.bn1 = HipBatchNorm3d(planes)
        m = mid_planes(planes, planes)
        self.conv2_s = conv1x3x3(planes, m, stride)
        self.bn2_s = HipBatchNorm3d(m)
        self.conv2_t = conv3x1x1(m, planes, stride)
        self.bn2_t = HipBatchNorm3d(planes)
        self.conv3 = conv1x1x1(planes, planes * 4)
        self.bn3 = HipBatchNorm3d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def fwd(self, tape, xv):
        o = L.f_conv_bn_act(tape, self.conv1, self.bn1, xv)
        o = L.f_conv_bn_act(tape, self.conv2_s, self.bn2_s, o)
        o = L.f_conv_bn_act(tape, self.conv2_t, self.bn2_t, o)
        r = self._shortcut(tape, xv)
        return L.f_conv_bn_act(tape, self.conv3, self.bn3, o, relu=True, residual=r)


class ResNet(nn.Module):
    def __init__(self, block, layers, block_inplanes, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1,
                 no_max_pool=False, shortcut_type='B', widen_factor=1.0, n_classes=400):
        super().__init__()
        if shortcut_type != 'B':
            raise NotImplementedError('only shortcut type B (1x1x1 conv + BN) is on the pre-training path')
        widths = [int(w * widen_factor) for w in block_inplanes]
        self.in_planes = widths[0]
        self.no_max_pool = no_max_pool
        m = (3 * self.in_planes * conv1_t_size * 49) // (3 * 49 + conv1_t_size * self.in_planes)
        self.conv1_s = HipConv3d(n_input_channels, m, (1, 7, 7), (1, 2, 2), (0, 3, 3))
        self.bn1_s = HipBatchNorm3d(m)
        self.conv1_t = HipConv3d(m, self.in_planes, (conv1_t_size, 1, 1), (conv1_t_stride, 1, 1),
                                 (conv1_t_size // 2, 0, 0))
        self.bn1_t = HipBatchNorm3d(self.in_planes)
        self.maxpool = HipMaxPool3d(3, 2, 1)
        self.layer1 = self._make_layer(block, widths[0], layers[0])
        self.layer2 = self._make_layer(block, widths[1], layers[1], 2)
        self.layer3 = self._make_layer(block, widths[2], layers[2], 2)
        self.layer4 = self._make_layer(block, widths[3], layers[3], 2)
        self.avgpool = L.HipIdentity()         # AdaptiveAvgPool3d((1,1,1)) has no parameters
        self.fc = HipLinear(widths[3] * block.expansion, n_classes)
        for mod in self.modules():             # same traversal order / RNG use as the reference (:200-207)
            if isinstance(mod, HipConv3d):
                nn.init.kaiming_normal_(mod.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(mod, HipBatchNorm3d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            ds = nn.Sequential(conv1x1x1(self.in_planes, planes * block.expansion, stride),
                               HipBatchNorm3d(planes * block.expansion))
        mods = [block(self.in_planes, planes, stride, ds)]
        self.in_planes = planes * block.expansion
        mods += [block(self.in_planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def fwd(self, tape, xv):
        x = L.f_conv_bn_act(tape, self.conv1_s, self.bn1_s, xv)
        if not self.no_max_pool:          # conv1_t -> bn1_t -> relu -> maxpool: BN+ReLU evaluated inside the pool kernel
            x = L.f_conv_bn_relu_maxpool(tape, self.conv1_t, self.bn1_t, self.maxpool, x)
        else:
            x = L.f_conv_bn_act(tape, self.conv1_t, self.bn1_t, x)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = L.f_seq(tape, stage, x)
        x = L.f_wavgpool(tape, x)              # global mean == AdaptiveAvgPool3d(1) + flatten (:260-262)
        return L.f_head_fc(tape, self.fc, x)


def generate_model(model_depth, **kwargs):
    cfg = {10: (BasicBlock, [1, 1, 1, 1]), 18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]),
           50: (Bottleneck, [3, 4, 6, 3]), 101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3]),
           200: (Bottleneck, [3, 24, 36, 3])}
    if model_depth not in cfg:
        raise ValueError('unsupported R(2+1)D depth %r' % (model_depth,))
    block, layers = cfg[model_depth]
    return ResNet(block, layers, get_inplanes(), **kwargs)
