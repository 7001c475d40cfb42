from .build import create_visual_model  # noqa: F401
from .graph_wrappers import GraphWrapper  # noqa: F401
from .visual_wrappers import VisualModelWrapper  # noqa: F401
