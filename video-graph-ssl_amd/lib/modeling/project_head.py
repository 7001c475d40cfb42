"""Projection / prediction heads on the HIP engine (reference: lib/modeling/project_head.py).
Same module tree as the reference -- ``head.{0,2}`` Linear, ``l{1,2,3}.{0,1}`` Linear/BatchNorm1d --
so state dicts interchange."""
import torch.nn as nn

from ...engine import layers as L
from ...engine.layers import HipBatchNorm1d, HipLinear, HipNormalize, HipReLU


class Normalize(HipNormalize):
    """Row L2 normalisation (project_head.py:4-10)."""


class ProjectHead(nn.Module):
    def __init__(self, in_dim, feat_dim=128, head_type='mlp'):
        super().__init__()
        if head_type == 'linear':
            self.head = nn.Sequential(HipLinear(in_dim, feat_dim), Normalize(2))
        elif head_type == 'mlp':
            self.head = nn.Sequential(HipLinear(in_dim, in_dim), HipReLU(inplace=True),
                                      HipLinear(in_dim, feat_dim), Normalize(2))
        else:
            raise NotImplementedError('head not supported: {}'.format(head_type))

    def fwd(self, tape, xv):
        for m in self.head:
            if isinstance(m, HipLinear):
                xv = L.f_linear(tape, m, xv)
            elif isinstance(m, HipReLU):
                xv = L.f_relu(tape, xv)
            else:
                xv = L.f_l2norm(tape, xv)
        return xv


def _lin_bn(tape, seq, xv):
    """Sequential(Linear, BatchNorm1d[, ReLU]) -> linear GEMM + fused BN1d(+ReLU)."""
    xv = L.f_linear(tape, seq[0], xv)
    return L.f_bn1d_act(tape, seq[1], xv, relu=len(seq) > 2)


class ProjectionMLP(nn.Module):
    def __init__(self, in_dim, hid_dim, out_dim):
        super().__init__()
        self.l1 = nn.Sequential(HipLinear(in_dim, hid_dim), HipBatchNorm1d(hid_dim), HipReLU(inplace=True))
        self.l2 = nn.Sequential(HipLinear(hid_dim, hid_dim), HipBatchNorm1d(hid_dim), HipReLU(inplace=True))
        self.l3 = nn.Sequential(HipLinear(hid_dim, out_dim), HipBatchNorm1d(out_dim))

    def fwd(self, tape, xv):
        return _lin_bn(tape, self.l3, _lin_bn(tape, self.l2, _lin_bn(tape, self.l1, xv)))


class PredictionMLP(nn.Module):
    def __init__(self, in_dim, hid_dim, out_dim):
        super().__init__()
        self.l1 = nn.Sequential(HipLinear(in_dim, hid_dim), HipBatchNorm1d(hid_dim), HipReLU(inplace=True))
        self.l2 = HipLinear(hid_dim, out_dim)

    def fwd(self, tape, xv):
        return L.f_linear(tape, self.l2, _lin_bn(tape, self.l1, xv))
