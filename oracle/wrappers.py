"""Oracle model assembly (heads + wrappers).  Test infrastructure -- see oracle/__init__.py.

Follows lib/modeling/{project_head,visual_wrappers,graph_wrappers,build}.py.  State-dict
keys match the reference: ``model.encoder.base_model.*``, ``model.proj_head.head.{0,2}.*``,
``model.projection.l{1,2,3}.{0,1}.*``, ``model.prediction.l1.{0,1}.*``, ``model.prediction.l2.*``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoders import BACKBONES
from .graph import build_aug_block


class Normalize(nn.Module):
    """project_head.py:4-10."""

    def forward(self, x):
        return F.normalize(x, p=2, dim=1)


class ProjectHead(nn.Module):
    """project_head.py:12-34: 'mlp' = Linear(in,in) ReLU Linear(in,feat) L2-norm."""

    def __init__(self, in_dim, feat_dim=128, head_type='mlp'):
        super().__init__()
        if head_type == 'linear':
            self.head = nn.Sequential(nn.Linear(in_dim, feat_dim), Normalize())
        elif head_type == 'mlp':
            self.head = nn.Sequential(nn.Linear(in_dim, in_dim), nn.ReLU(inplace=True),
                                      nn.Linear(in_dim, feat_dim), Normalize())
        else:
            raise NotImplementedError('head not supported: {}'.format(head_type))

    def forward(self, x):
        return self.head(x)


class ProjectionMLP(nn.Module):
    """project_head.py:36-58."""

    def __init__(self, in_dim, hid, out):
        super().__init__()
        self.l1 = nn.Sequential(nn.Linear(in_dim, hid), nn.BatchNorm1d(hid), nn.ReLU(inplace=True))
        self.l2 = nn.Sequential(nn.Linear(hid, hid), nn.BatchNorm1d(hid), nn.ReLU(inplace=True))
        self.l3 = nn.Sequential(nn.Linear(hid, out), nn.BatchNorm1d(out))

    def forward(self, x):
        return self.l3(self.l2(self.l1(x)))


class PredictionMLP(nn.Module):
    """project_head.py:61-76."""

    def __init__(self, in_dim, hid, out):
        super().__init__()
        self.l1 = nn.Sequential(nn.Linear(in_dim, hid), nn.BatchNorm1d(hid), nn.ReLU(inplace=True))
        self.l2 = nn.Linear(hid, out)

    def forward(self, x):
        return self.l2(self.l1(x))


def neg_cosine(p, z):
    """graph_wrappers.py:93-108, fun_type 'v2': -cos(p, stopgrad(z)).mean()."""
    return -F.cosine_similarity(p, z.detach(), dim=-1).mean()


class Identity(nn.Module):
    def forward(self, x):
        return x


# default insertion points of the graph block, visual_wrappers.py:113-124
AUG_SITES = {'S3D': ['base.5', 'base.9', 'base.14']}


class VisualModelWrapper(nn.Module):
    """visual_wrappers.py:9-99, 3D/RGB path only: build backbone by name, read feature_dim
    off `.fc`, replace `.fc` by Identity (DROPOUT==0) or Dropout, optionally insert graph
    blocks; forward = backbone(x).view(-1, feature_dim)."""

    def __init__(self, clip_length, backbone_name='S3D', dropout=0.0, aug_flag=False,
                 module_name_list=None):
        super().__init__()
        self.clip_length = clip_length
        self.base_model = BACKBONES[backbone_name]()
        fc = self.base_model.fc
        self.feature_dim = fc[0].in_channels if backbone_name == 'S3D' else fc.in_features   # :103-106
        self.base_model.fc = Identity() if dropout == 0 else nn.Dropout(p=dropout)           # :107-110
        if aug_flag:
            names = module_name_list or AUG_SITES.get(backbone_name, ['layer2', 'layer3', 'layer4'])
            build_aug_block(self.base_model, names, n_segments=clip_length)

    def forward(self, x):
        return self.base_model(x).view(-1, self.feature_dim)


class ContrastWrapper(nn.Module):
    """graph_wrappers.py:8-26."""

    def __init__(self, encoder, hid_dim=128, head_type='mlp'):
        super().__init__()
        self.encoder = encoder
        self.proj_head = ProjectHead(encoder.feature_dim, hid_dim, head_type)

    def forward(self, x):
        return self.proj_head(self.encoder(x))


class SimSiam(nn.Module):
    """graph_wrappers.py:30-71: both views through encoder/projection/prediction with grad;
    L = D(p1,z2)/2 + D(p2,z1)/2."""

    def __init__(self, encoder, hid_dim=1024):
        super().__init__()
        self.encoder = encoder
        self.feature_dim = encoder.feature_dim
        self.projection = ProjectionMLP(self.feature_dim, hid_dim, hid_dim)
        self.prediction = PredictionMLP(hid_dim, hid_dim // 2, hid_dim)

    def forward(self, x):
        x1, x2 = torch.chunk(x, 2, dim=1)
        z1 = self.projection(self.encoder(x1))
        p1 = self.prediction(z1)
        z2 = self.projection(self.encoder(x2))
        p2 = self.prediction(z2)
        return neg_cosine(p1, z2) / 2 + neg_cosine(p2, z1) / 2


class GraphWrapper(nn.Module):
    """graph_wrappers.py:110-120."""

    def __init__(self, encoder, hid_dim=1024, head_type='mlp', mem_type='simsiam'):
        super().__init__()
        self.model = SimSiam(encoder, hid_dim) if mem_type == 'simsiam' else \
            ContrastWrapper(encoder, hid_dim, head_type)

    def forward(self, x):
        return self.model(x)


def create_visual_model(backbone='S3D', clip_length=16, feat_dim=128, head_type='mlp',
                        mem_type='moco', dropout=0.0, aug_flag=False, module_name_list=None):
    """lib/modeling/build.py:16-32 -> (model, model_ema | None)."""
    mk = lambda: GraphWrapper(VisualModelWrapper(clip_length, backbone, dropout, aug_flag, module_name_list),
                              feat_dim, head_type, mem_type)
    model = mk()
    return model, (mk() if mem_type == 'moco' else None)
