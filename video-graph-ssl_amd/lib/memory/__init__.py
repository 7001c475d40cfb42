from .build import create_contrast, create_criterion  # noqa: F401
from .criterion import NCESoftmaxLoss  # noqa: F401
from .mem_moco import RGBMoCo  # noqa: F401
