"""VisualModelWrapper for the 3D / RGB pre-training path (reference: lib/modeling/visual_wrappers.py).

Keeps the constructor signature and attributes the reference's factories use (:10-25), resolves
the backbone by name from the registry (:130-135), reads ``feature_dim`` off the backbone's ``fc``
and replaces it (:102-110), optionally inserts temporal-graph blocks (:112-126).  The 2D / optical
flow branches of the reference are downstream-only and out of scope."""
import torch.nn as nn

from . import backbone
from ..ops import build_aug_block, get_agg
from ...engine.layers import HipIdentity


class Identity(HipIdentity):
    pass


class VisualModelWrapper(nn.Module):
    def __init__(self, clip_length, modality, backbone_name='S3D', backbone_type='3D', new_length=None,
                 agg_fun='avg', before_softmax=True, dropout=0.8, crop_num=1, partial_bn=True, fc_sche=False,
                 custom_flag=False, pretrained=False, module_name_list=None, pretrain_path=None, aug_flag=False):
        super().__init__()
        if backbone_type != '3D':
            raise ValueError('Only the 3D backbones are on the pre-training hot path (got %r)' % (backbone_type,))
        if modality != 'RGB':
            raise ValueError('Only RGB clips are on the pre-training hot path (got %r)' % (modality,))
        if not before_softmax and agg_fun != 'avg':
            raise ValueError('Only avg aggregattion can be used after Softmax')
        self.modality, self.backbone_name, self.backbone_type = modality, backbone_name, backbone_type
        self.clip_length, self.dropout, self.agg_fun = clip_length, dropout, agg_fun
        self.module_name_list, self.aug_flag = module_name_list, aug_flag
        self.pretrained, self.pretrain_path = pretrained, pretrain_path
        self.new_length = 1 if new_length is None else new_length
        self._prepare_base_model(backbone_name)
        self.feature_dim = self._prepare_video_model()
        self.aggregation = get_agg(agg_fun=agg_fun, model_type=backbone_type)
        self._enable_pbn = partial_bn

    def _prepare_base_model(self, backbone_name):
        ctor = getattr(backbone.backbone_3d, backbone_name, None)
        if ctor is None:
            raise ValueError('unknown 3D backbone %r' % (backbone_name,))
        self.base_model = ctor()
        if self.pretrained and self.pretrain_path != 'none':
            import torch
            self.base_model.load_state_dict(torch.load(self.pretrain_path))
        self.base_model.last_layer_name = 'fc'

    def _prepare_video_model(self):
        last = getattr(self.base_model, self.base_model.last_layer_name)
        feature_dim = last[0].in_channels if self.backbone_name == 'S3D' else last.in_features
        setattr(self.base_model, self.base_model.last_layer_name,
                Identity() if self.dropout == 0 else nn.Dropout(p=self.dropout))
        if self.aug_flag:
            if self.module_name_list is None:
                if self.backbone_name == 'S3D':
                    self.module_name_list = ['base.5', 'base.9', 'base.14']
                else:
                    self.module_name_list = ['layer2', 'layer3', 'layer4']
            self.base_model = build_aug_block(self.base_model, self.module_name_list, n_segments=self.clip_length)
        return feature_dim

    def fwd(self, tape, xv):
        """(B,3,T,H,W) NCDHW fp32 -> (B, feature_dim)   (reference forward :76-97, 3D branch)."""
        out = self.base_model.fwd(tape, xv)
        if out.t.dim() != 2 or out.t.shape[1] != self.feature_dim:
            raise RuntimeError('backbone returned %r, expected (B, %d)' % (tuple(out.t.shape), self.feature_dim))
        return out
