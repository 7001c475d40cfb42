from .defaults import CfgNode, get_defaults  # noqa: F401

cfg = get_defaults()
