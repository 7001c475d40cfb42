"""gca-mi355x: MI355X-native (gfx950) implementation of GCA's contrastive pre-training hot path.

    lib/       host-side mirror of the reference's interface (lib.modeling / lib.ops / lib.memory /
               lib.solver / lib.config): same factories, module trees and state-dict keys
    engine/    fused-op tape, flat parameter arenas, hipGraph-captured train step
    csrc/      hand-written HIP kernels + the C ABI (include/gca_hip.h) -> libgca_hip.so
    parallel   RCCL exchange steps (ShuffleBN row all-to-all, negatives all-gather, grad all-reduce)

The directory name contains a hyphen (it follows the reference repo's name), so import it with
``importlib.import_module('video-graph-ssl_amd')``.  Importing the package loads libgca_hip.so and
raises if it has not been built -- there is no CPU fallback.
"""
from . import _hip  # noqa: F401  (fails loudly when the HIP library is missing)
from . import engine, lib, parallel  # noqa: F401
from .engine.trainer import MoCoTrainer, SimSiamTrainer  # noqa: F401
from .lib import evaluation  # noqa: F401  (accuracy / AverageMeter, lib/evaluation/metric.py)
from .lib.config import CfgNode, get_defaults  # noqa: F401
from .lib.memory import create_contrast, create_criterion  # noqa: F401
from .lib.modeling import create_visual_model  # noqa: F401
from .lib.solver import make_lr_scheduler, make_optimizer  # noqa: F401

__version__ = '0.1.0'
