from . import backbone_3d  # noqa: F401
