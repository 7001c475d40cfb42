"""Factories (reference: lib/memory/build.py).  Only what the pre-training configs select is on the
HIP path: MEM_TYPE 'moco' (visual) / 'simsiam', CRITERION 'crossentropy'."""
from .criterion import NCESoftmaxLoss
from .mem_moco import RGBMoCo


def create_contrast(cfg, n_data):
    if cfg.CONTRAST.MEM_TYPE == 'moco':
        if cfg.CROSS.MODALITY != 'visual':
            raise NotImplementedError('two-modality CMCMoCo is outside the hot path (CROSS.MODALITY is always visual)')
        return RGBMoCo(cfg.CROSS.FEAT_DIM, cfg.CONTRAST.NCE_K, cfg.CONTRAST.NCE_T)
    if cfg.CONTRAST.MEM_TYPE == 'simsiam':
        return None
    raise NotImplementedError('mem not suported: {}'.format(cfg.CONTRAST.MEM_TYPE))


def create_criterion(cfg, n_data):
    if cfg.CROSS.CRITERION == 'crossentropy':
        return NCESoftmaxLoss()
    raise NotImplementedError('criterion not suported: {}'.format(cfg.CROSS.CRITERION))
