"""S3D (separable Inception-3D) clip encoder on the HIP engine.

Architecture / parameter names / default initialisation as the reference's
lib/modeling/backbone/backbone_3d/s3d_1.py (S3D :5-35, BasicConv3d :37-48, SepConv3d :50-69,
Mixed_3b..5c :71-329), written table-driven.  Engine-specific choices:
  * each conv+BN(eps 1e-3, momentum 1e-3)+ReLU is one fused unit (stats in the conv epilogue);
  * the four Inception branches write their result straight into channel slices of the block
    output (no torch.cat copy);
  * the tail avg_pool3d((2,H,W), stride 1) -> mean over T' (:30-33) is ONE frame-weighted
    global mean (frames 0 and T-1 weigh 1, inner frames 2).
"""
import torch
import torch.nn as nn

from .....engine import layers as L
from .....engine.layers import HipBatchNorm3d, HipConv3d, HipMaxPool3d

_BN = dict(eps=1e-3, momentum=0.001, affine=True)


class BasicConv3d(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = HipConv3d(in_planes, out_planes, kernel_size, stride, padding)
        self.bn = HipBatchNorm3d(out_planes, **_BN)
        self.relu = L.HipReLU()

    def fwd(self, tape, xv, out=None):
        return L.f_conv_bn_act(tape, self.conv, self.bn, xv, out=out)


class SepConv3d(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        k, s, p = kernel_size, stride, padding
        self.conv_s = HipConv3d(in_planes, out_planes, (1, k, k), (1, s, s), (0, p, p))
        self.bn_s = HipBatchNorm3d(out_planes, **_BN)
        self.relu_s = L.HipReLU()
        self.conv_t = HipConv3d(out_planes, out_planes, (k, 1, 1), (s, 1, 1), (p, 0, 0))
        self.bn_t = HipBatchNorm3d(out_planes, **_BN)
        self.relu_t = L.HipReLU()

    def fwd(self, tape, xv, out=None):
        x = L.f_conv_bn_act(tape, self.conv_s, self.bn_s, xv)
        return L.f_conv_bn_act(tape, self.conv_t, self.bn_t, x, out=out)


class _Mixed(nn.Module):
    """branch0: 1x1x1 | branch1: 1x1x1 -> sep3 | branch2: 1x1x1 -> sep3 | branch3: maxpool3 s1 -> 1x1x1."""
    SPEC = None      # (cin, b0, b1a, b1b, b2a, b2b, b3)

    def __init__(self):
        super().__init__()
        cin, b0, b1a, b1b, b2a, b2b, b3 = self.SPEC
        self.in_channels = cin
        self.widths = (b0, b1b, b2b, b3)
        self.branch0 = nn.Sequential(BasicConv3d(cin, b0, 1, 1))
        self.branch1 = nn.Sequential(BasicConv3d(cin, b1a, 1, 1), SepConv3d(b1a, b1b, 3, 1, 1))
        self.branch2 = nn.Sequential(BasicConv3d(cin, b2a, 1, 1), SepConv3d(b2a, b2b, 3, 1, 1))
        self.branch3 = nn.Sequential(HipMaxPool3d((3, 3, 3), 1, 1), BasicConv3d(cin, b3, 1, 1))

    def fwd(self, tape, xv):
        x = xv.t
        N, _, D, H, W = x.shape
        buf = torch.empty((N, sum(self.widths), D, H, W), dtype=x.dtype, device=x.device)
        offs, o = [], 0
        for w in self.widths:
            offs.append(o)
            o += w
        sl = [buf[:, a:a + w] for a, w in zip(offs, self.widths)]
        v0 = self.branch0[0].fwd(tape, xv, out=sl[0])
        v1 = self.branch1[1].fwd(tape, self.branch1[0].fwd(tape, xv), out=sl[1])
        v2 = self.branch2[1].fwd(tape, self.branch2[0].fwd(tape, xv), out=sl[2])
        v3 = self.branch3[1].fwd(tape, L.f_maxpool(tape, self.branch3[0], xv), out=sl[3])
        return L.f_concat(tape, buf, [v0, v1, v2, v3], offs)


def _mixed(name, spec):
    return type(name, (_Mixed,), {'SPEC': spec})


Mixed_3b = _mixed('Mixed_3b', (192, 64, 96, 128, 16, 32, 32))
Mixed_3c = _mixed('Mixed_3c', (256, 128, 128, 192, 32, 96, 64))
Mixed_4b = _mixed('Mixed_4b', (480, 192, 96, 208, 16, 48, 64))
Mixed_4c = _mixed('Mixed_4c', (512, 160, 112, 224, 24, 64, 64))
Mixed_4d = _mixed('Mixed_4d', (512, 128, 128, 256, 24, 64, 64))
Mixed_4e = _mixed('Mixed_4e', (512, 112, 144, 288, 32, 64, 64))
Mixed_4f = _mixed('Mixed_4f', (528, 256, 160, 320, 32, 128, 128))
Mixed_5b = _mixed('Mixed_5b', (832, 256, 160, 320, 32, 128, 128))
Mixed_5c = _mixed('Mixed_5c', (832, 384, 192, 384, 48, 128, 128))


class S3D(nn.Module):
    def __init__(self, num_class=400):
        super().__init__()
        self.base = nn.Sequential(
            SepConv3d(3, 64, kernel_size=7, stride=2, padding=3),
            HipMaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)),
            BasicConv3d(64, 64, kernel_size=1, stride=1),
            SepConv3d(64, 192, kernel_size=3, stride=1, padding=1),
            HipMaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)),
            Mixed_3b(), Mixed_3c(),
            HipMaxPool3d((3, 3, 3), (2, 2, 2), (1, 1, 1)),
            Mixed_4b(), Mixed_4c(), Mixed_4d(), Mixed_4e(), Mixed_4f(),
            HipMaxPool3d((2, 2, 2), (2, 2, 2), (0, 0, 0)),
            Mixed_5b(), Mixed_5c())
        self.fc = nn.Sequential(HipConv3d(1024, num_class, 1, 1, 0, bias=True))
        self._wt = {}

    def _tail_weights(self, T, HW, device):
        key = (T, HW, device)
        if key not in self._wt:
            if T < 2:
                raise RuntimeError('S3D needs a final temporal extent >= 2 (avg_pool3d((2,H,W)), s3d_1.py:30)')
            w = torch.full((T,), 2.0)
            w[0] = w[-1] = 1.0
            self._wt[key] = (w.to(device), 1.0 / (2.0 * HW * (T - 1)))
        return self._wt[key]

    def fwd(self, tape, xv):
        x = xv
        for m in self.base:                        # incl. graph blocks inserted by lib/ops/build.py
            x = L.f_seq(tape, m, x)
        T, HW = x.t.shape[2], x.t.shape[3] * x.t.shape[4]
        if isinstance(self.fc, nn.Dropout) and self.fc.p > 0 and self.fc.training:
            return L.f_s3d_tail_with_dropout(tape, self.fc, x)       # dropout sits between the two averages (:31)
        wt, norm = self._tail_weights(T, HW, x.t.device)
        x = L.f_wavgpool(tape, x, wt, norm)
        if isinstance(self.fc, nn.Sequential):
            raise NotImplementedError('S3D classifier head (1x1x1 conv fc) is stripped for pre-training '
                                      '(visual_wrappers.py:107-110); build it through VisualModelWrapper')
        return L.f_head_fc(tape, self.fc, x)
