from .build import build_aug_block, get_agg  # noqa: F401
from .module_wrappers import TemporalGraphAug  # noqa: F401
