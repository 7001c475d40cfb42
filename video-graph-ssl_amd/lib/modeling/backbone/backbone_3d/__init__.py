"""Backbone registry.  VisualModelWrapper resolves ``cfg.MODEL.BACKBONE`` with
``getattr(backbone.backbone_3d, name)()`` (reference visual_wrappers.py:130-135), i.e. every entry
is a zero-argument callable returning an nn.Module with an ``fc`` attribute.  The reference only
exports S3D; R(2+1)D and 3D-ResNet exist there as unregistered files and are
registered here under the names BASELINE.json's configs use."""
from .resnet2p1d import generate_model as _r2p1d
from .s3d_1 import S3D
from .resnet import resnet18 as _r3d18, resnet50 as _r3d50


def R2P1D10():
    return _r2p1d(10)


def R2P1D18():
    return _r2p1d(18)


def R2P1D34():
    return _r2p1d(34)


def R2P1D50():
    return _r2p1d(50)


def R3D18():
    return _r3d18(sample_size=112, sample_duration=16)


def R3D50():
    return _r3d50(sample_size=224, sample_duration=32)


def register(name, ctor):
    """Plug an extra zero-arg backbone constructor into the registry (tests use tiny ones)."""
    globals()[name] = ctor
