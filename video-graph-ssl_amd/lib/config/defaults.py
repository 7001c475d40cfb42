"""Config defaults (reference: lib/config/defaults.py, a yacs CfgNode tree).  yacs is not a dependency
here: CfgNode below is a minimal attribute tree with the three methods the reference's entry point
uses (merge_from_file, merge_from_list, freeze -- tools/train_video_contrast_dis.py:548-551), so
configs/visual_moco.yaml and configs/visual_simsiam.yaml drop in unchanged.  Only keys read on the
pre-training path carry defaults; unknown keys in a YAML are accepted and kept."""
import ast

import yaml


class CfgNode(object):
    def __init__(self, d=None):
        object.__setattr__(self, '_frozen', False)
        for k, v in (d or {}).items():
            setattr(self, k, CfgNode(v) if isinstance(v, dict) else v)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError('config is frozen')
        object.__setattr__(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def to_dict(self):
        return {k: (getattr(self, k).to_dict() if isinstance(getattr(self, k), CfgNode) else getattr(self, k))
                for k in self.keys()}

    def _merge(self, d):
        for k, v in d.items():
            cur = getattr(self, k, None)
            if isinstance(v, dict):
                if not isinstance(cur, CfgNode):
                    cur = CfgNode()
                    setattr(self, k, cur)
                cur._merge(v)
            else:
                setattr(self, k, v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError('override list must be KEY VALUE pairs')
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = getattr(node, p)
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            setattr(node, parts[-1], val)

    def freeze(self):
        for k in self.keys():
            v = getattr(self, k)
            if isinstance(v, CfgNode):
                v.freeze()
        object.__setattr__(self, '_frozen', True)

    def clone(self):
        return CfgNode(self.to_dict())


def get_defaults():
    return CfgNode({
        'MODEL': dict(DEVICE='cuda', DEVICE_IDS='0, 1, 2, 3', SEED=1, BACKBONE='S3D', BACKBONE_TYPE='3D',
                      PRETRAINED=False, PRETRAIN_PATH='none', POOLING_TYPE='avg', DROPOUT=0.5,
                      NO_PARTIALBN=False, DISTRIBUTED=True, AUG_FLAG=False),
        'INPUT': dict(BASE_SIZE=[224, 224], CROP_SIZE=[224, 224], MEAN=[0.485, 0.456, 0.406],
                      STD=[0.229, 0.224, 0.225], MODALITY='RGB', SAMPLE_TYPE='uniform', VIDEO_LENGTH=16,
                      FLIP=True, TEMPORAL_JITTER=False),
        'DATASET': dict(NAME='kinetics', NUM_CLASS=101),
        'DATALOADER': dict(NUM_WORKERS=8, BATCH_SIZE=128),
        'SOLVER': dict(OPTIMIZER_NAME='SGD', LR_SCHEDULER='poly', MAX_EPOCHS=50, START_EPOCH=0, BASE_LR=0.001,
                       BIAS_LR_FACTOR=2, MOMENTUM=0.9, WEIGHT_DECAY=5e-4, WEIGHT_DECAY_BIAS=0, NESTEROV=False,
                       USE_TRICK=False, LR_STEP=20, CLIP_GRADIENT='none', NO_PARTIALBN=True, GAMMA=0.1,
                       STEPS=(30, 60), WARMUP_FACTOR=1.0 / 3, WARMUP_ITERS=5, WARMUP_METHOD='linear'),
        'APEX': dict(FLAG=False, OPT_LEVEL='O1', LOCAL_RANK=-1),
        'CHECKPOINT': dict(RESUME='none', CHECKNAME='video_model', CHECKPOINT_INTERVAL=20, NO_VAL=False,
                           FINETUNE=False, PRINT_FREQ=20),
        'CONTRAST': dict(MEM_TYPE='bank', NCE_K=65536, NCE_T=0.07, NCE_M=0.5, ALPHA=0.999, JIGSAW=False),
        'CROSS': dict(FEAT_DIM=128, HEAD_TYPE='mlp', MEM=None, BETA=0.5, MODALITY='visual',
                      CRITERION='crossentropy'),
    })
