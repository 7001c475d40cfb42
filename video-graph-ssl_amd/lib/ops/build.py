"""Graph-block insertion (reference: lib/ops/build.py).  The reference version raises for every
name (uses ``module.in_channels`` which Inception blocks lack, and would only wrap the last name);
this is the intended behaviour: EVERY named sub-module m becomes Sequential(TemporalGraphAug(C_in(m)), m)."""
import torch.nn as nn

from .module_wrappers.temporal_graph import AugThen, TemporalGraphAug
from ...engine.layers import HipConv3d


class TemporalAggreModel(nn.Module):
    """Placeholder of the reference's 2D-path temporal aggregation (pooling_opts/basic_ops_wrap.py);
    the 3D path returns before it is used (visual_wrappers.py:96-97)."""

    def __init__(self, pooling='avg', model_type='2D'):
        super().__init__()
        self.pooling, self.model_type = pooling, model_type


def get_agg(agg_fun='avg', model_type='2D'):
    return TemporalAggreModel(pooling=agg_fun, model_type='2D')


def _first_conv_in_channels(module):
    if hasattr(module, 'in_channels'):
        return module.in_channels
    for m in module.modules():
        if isinstance(m, HipConv3d):
            return m.in_channels
    raise ValueError('cannot infer the input channels of %r' % type(module))


def build_aug_block(base_model, module_name_list, n_segments=None):
    for name in module_name_list:
        parts = name.split('.')
        parent = base_model
        for p in parts[:-1]:
            parent = getattr(parent, p)
        mod = getattr(parent, parts[-1])
        setattr(parent, parts[-1], AugThen(TemporalGraphAug(in_channels=_first_conv_in_channels(mod)), mod))
    return base_model
