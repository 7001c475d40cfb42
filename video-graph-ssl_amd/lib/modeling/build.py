"""Model factory (reference: lib/modeling/build.py:16-32).  Same cfg keys, same return value."""
from .graph_wrappers import GraphWrapper
from .visual_wrappers import VisualModelWrapper


def _encoder(cfg):
    return VisualModelWrapper(cfg.INPUT.VIDEO_LENGTH, cfg.INPUT.MODALITY, backbone_name=cfg.MODEL.BACKBONE,
                              backbone_type=cfg.MODEL.BACKBONE_TYPE, agg_fun=cfg.MODEL.POOLING_TYPE,
                              dropout=cfg.MODEL.DROPOUT, partial_bn=not cfg.SOLVER.NO_PARTIALBN,
                              pretrained=cfg.MODEL.PRETRAINED, pretrain_path=cfg.MODEL.PRETRAIN_PATH,
                              aug_flag=bool(getattr(cfg.MODEL, 'AUG_FLAG', False)))


def create_visual_model(cfg):
    """-> (model, model_ema | None).  NOTE: the reference never forwards MODEL.AUG_FLAG (dead code,
    SURVEY.md fact 5); here the flag is honoured so the graph block can actually be switched on."""
    model = GraphWrapper(_encoder(cfg), cfg.CROSS.FEAT_DIM, cfg.CROSS.HEAD_TYPE, cfg.CONTRAST.MEM_TYPE)
    model_ema = None
    if cfg.CONTRAST.MEM_TYPE == 'moco':
        model_ema = GraphWrapper(_encoder(cfg), cfg.CROSS.FEAT_DIM, cfg.CROSS.HEAD_TYPE, cfg.CONTRAST.MEM_TYPE)
    return model, model_ema
