#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE's own classes.

Run only in the build container (needs /root/reference, CPU torch):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing of the reference is copied: the reference modules are imported from where they
lie, fed seeded inputs, and only inputs / weights / outputs are written as .npz data.
Harness shims applied at import time (reference files untouched):
  1. torch.Tensor.cuda = identity          (mem_moco.py:25,78 / criterion.py:43 hard-code .cuda())
  2. TemporalGraphAug.reset_parameter = TemporalGraphAug.reset_parameters   (ctor typo, temporal_graph.py:117,124)
  3. register the unexported backbones on the imported backbone_3d package
  4. cfg objects are types.SimpleNamespace (yacs absent)
  5. collections.Iterable = collections.abc.Iterable  (lr_scheduler.py:54 on Python >= 3.10)
  6. (gen_input only) inert placeholder modules for `cv2` and `albumentations.augmentations.functional`, which
     lib/data/transform/consistency_transforms.py imports at its top and which are not installed: any attribute reads as 0
     (the file uses cv2.INTER_LINEAR etc. as default arguments), nothing of them is ever CALLED -- the two classes exercised,
     VideoNormalize and VideoToTensor, are pure numpy / torch
"""
import collections
import collections.abc
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
REF = os.environ.get('GCA_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

torch.Tensor.cuda = lambda self, *a, **k: self                    # shim 1
collections.Iterable = collections.abc.Iterable                   # shim 5

from lib.modeling.backbone import backbone_3d                      # noqa: E402
from lib.modeling.backbone.backbone_3d import resnet2p1d, resnet, s3d_1   # noqa: E402
from lib.modeling import build as ref_build                        # noqa: E402
from lib.modeling import project_head as ref_head                  # noqa: E402
from lib.modeling import graph_wrappers as ref_gw                  # noqa: E402
from lib.memory import mem_moco, criterion as ref_crit            # noqa: E402
from lib.ops.module_wrappers import temporal_graph as ref_tg       # noqa: E402
from lib.solver import build as ref_solver                         # noqa: E402
from lib.evaluation.metric import accuracy as ref_accuracy         # noqa: E402

ref_tg.TemporalGraphAug.reset_parameter = ref_tg.TemporalGraphAug.reset_parameters   # shim 2
backbone_3d.R2P1D10T = lambda: resnet2p1d.generate_model(10, widen_factor=0.125)      # shim 3
backbone_3d.R2P1D18 = lambda: resnet2p1d.generate_model(18)

import oracle.encoders as oenc                                      # noqa: E402  (seed-equivalence asserts only)

torch.set_num_threads(4)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def seeded_randn(seed, *shape):
    """Big inputs are not stored: tests regenerate them from (seed, shape) with the same CPU generator."""
    return torch.randn(*shape, generator=torch.Generator().manual_seed(int(seed)))


def spec(seed, *shape):
    return np.array([seed] + list(shape), dtype=np.int64)


def sd_np(mod, prefix='w:'):
    return {prefix + k: v.clone() for k, v in mod.state_dict().items()}


def cfg_for(backbone, mem_type, feat_dim=128, T=8):
    return NS(INPUT=NS(VIDEO_LENGTH=T, MODALITY='RGB'),
              MODEL=NS(BACKBONE=backbone, BACKBONE_TYPE='3D', POOLING_TYPE='avg', DROPOUT=0.0,
                       PRETRAINED=False, PRETRAIN_PATH='none'),
              SOLVER=NS(NO_PARTIALBN=True, BASE_LR=0.06, WEIGHT_DECAY=5e-4, BIAS_LR_FACTOR=2,
                        WEIGHT_DECAY_BIAS=0, MOMENTUM=0.9, NESTEROV=False, USE_TRICK=False,
                        OPTIMIZER_NAME='SGD', STEPS=[80, 120, 160], GAMMA=0.1, WARMUP_FACTOR=0.01,
                        WARMUP_ITERS=10, WARMUP_METHOD='linear', LR_SCHEDULER='step', MAX_EPOCHS=200),
              CROSS=NS(FEAT_DIM=feat_dim, HEAD_TYPE='mlp', MODALITY='visual', CRITERION='crossentropy'),
              CONTRAST=NS(MEM_TYPE=mem_type, NCE_K=20, NCE_T=0.07, NCE_M=0.5, ALPHA=0.999))


# ------------------------------------------------------------------ 1. per-op vectors
def gen_ops():
    g = torch.Generator().manual_seed(101)
    rn = lambda *s: torch.randn(*s, generator=g)
    out = {}
    # conv3d for every (kernel, stride, pad) family on the path, odd channel counts included
    convs = {
        'stem_s': (nn.Conv3d(3, 22, (1, 7, 7), (1, 2, 2), (0, 3, 3), bias=False), (2, 3, 4, 20, 20)),
        'stem_t': (nn.Conv3d(22, 16, (7, 1, 1), 1, (3, 0, 0), bias=False), (2, 22, 6, 5, 5)),
        's3d_t_s2': (nn.Conv3d(10, 10, (7, 1, 1), (2, 1, 1), (3, 0, 0), bias=False), (2, 10, 8, 5, 5)),
        'c1x3x3': (resnet2p1d.conv1x3x3(16, 37), (2, 16, 3, 9, 9)),
        'c1x3x3_s2': (resnet2p1d.conv1x3x3(16, 23, 2), (2, 16, 3, 9, 9)),
        'c3x1x1': (resnet2p1d.conv3x1x1(37, 16), (2, 37, 5, 4, 4)),
        'c3x1x1_s2': (resnet2p1d.conv3x1x1(23, 32, 2), (2, 23, 6, 4, 4)),
        'c1x1x1_s2': (resnet2p1d.conv1x1x1(16, 32, 2), (2, 16, 4, 8, 8)),
        'c1x1x1': (resnet2p1d.conv1x1x1(19, 8, 1), (2, 19, 3, 5, 5)),
        'c3x3x3': (resnet.conv3x3x3(9, 12), (2, 9, 4, 7, 7)),
        'c3x3x3_s2': (resnet.conv3x3x3(9, 12, 2), (2, 9, 5, 7, 7)),
        'c7x7x7': (nn.Conv3d(3, 8, 7, (1, 2, 2), (3, 3, 3), bias=False), (1, 3, 6, 16, 16)),
    }
    for name, (m, shp) in convs.items():
        torch.manual_seed(7)
        m.weight.data = rn(*m.weight.shape) * 0.2
        x = rn(*shp).requires_grad_(True)
        y = m(x)
        dy = rn(*y.shape)
        y.backward(dy)
        out.update({name + ':x': x, name + ':w': m.weight, name + ':y': y, name + ':dy': dy,
                    name + ':dx': x.grad, name + ':dw': m.weight.grad,
                    name + ':cfg': np.array(list(m.kernel_size) + list(m.stride) + list(m.padding))})
    # BatchNorm3d training mode (S3D's eps/momentum and the defaults), with backward
    for name, kw in {'bn_s3d': dict(eps=1e-3, momentum=0.001), 'bn_def': {}}.items():
        bn = nn.BatchNorm3d(5, **kw)
        bn.weight.data = rn(5).abs() + 0.5
        bn.bias.data = rn(5)
        bn.running_mean.data = rn(5) * 0.1
        bn.running_var.data = rn(5).abs() + 0.5
        rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
        x = (rn(3, 5, 4, 6, 6) * 2 + 1).requires_grad_(True)
        y = torch.relu(bn(x))
        dy = rn(*y.shape)
        y.backward(dy)
        out.update({name + ':x': x, name + ':g': bn.weight, name + ':b': bn.bias, name + ':rm0': rm0,
                    name + ':rv0': rv0, name + ':rm1': bn.running_mean, name + ':rv1': bn.running_var,
                    name + ':y_relu': y, name + ':dy': dy, name + ':dx': x.grad,
                    name + ':dg': bn.weight.grad, name + ':db': bn.bias.grad,
                    name + ':hp': np.array([bn.eps, bn.momentum])})
    # MaxPool3d variants on the path (s3d_1.py:10,16,22,87; resnet2p1d.py:178), post-ReLU style inputs (ties!)
    pools = {'mp133': ((1, 3, 3), (1, 2, 2), (0, 1, 1)), 'mp333s2': ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
             'mp222': ((2, 2, 2), (2, 2, 2), (0, 0, 0)), 'mp333s1': ((3, 3, 3), (1, 1, 1), (1, 1, 1)),
             'mp122': ((1, 2, 2), (1, 2, 2), (0, 0, 0))}
    for name, (k, s, p) in pools.items():
        x = torch.relu(rn(2, 3, 6, 9, 9)).requires_grad_(True)
        y = nn.MaxPool3d(k, s, p)(x)
        dy = rn(*y.shape)
        y.backward(dy)
        out.update({name + ':x': x, name + ':y': y, name + ':dy': dy, name + ':dx': x.grad,
                    name + ':cfg': np.array(list(k) + list(s) + list(p))})
    npz('ops', **out)


# ------------------------------------------------------------------ 2. blocks
def gen_blocks():
    out = {}
    torch.manual_seed(11)
    blk = resnet2p1d.BasicBlock(16, 16)
    x = torch.randn(2, 16, 4, 10, 10)
    blk.train()
    out.update(sd_np(blk, 'bb:w:'))
    out['bb:x'] = x
    out['bb:y'] = blk(x)
    out.update(sd_np(blk, 'bb:after:'))          # running stats after one train-mode forward

    torch.manual_seed(12)
    ds = nn.Sequential(resnet2p1d.conv1x1x1(16, 32, 2), nn.BatchNorm3d(32))
    blk2 = resnet2p1d.BasicBlock(16, 32, stride=2, downsample=ds)
    out.update(sd_np(blk2, 'bbs:w:'))
    x2 = torch.randn(2, 16, 4, 10, 10)
    out['bbs:x'] = x2
    out['bbs:y'] = blk2(x2)

    torch.manual_seed(13)
    sep = s3d_1.SepConv3d(3, 64, kernel_size=7, stride=2, padding=3)
    xs = torch.randn(1, 3, 8, 24, 24)
    out.update(sd_np(sep, 'sep:w:'))
    out['sep:x'] = xs
    out['sep:y'] = sep(xs)

    torch.manual_seed(14)
    mix = s3d_1.Mixed_3b()
    xm = torch.relu(torch.randn(1, 192, 4, 6, 6))
    out.update(sd_np(mix, 'm3b:w:'))
    out['m3b:x'] = xm
    out['m3b:y'] = mix(xm)

    torch.manual_seed(15)
    bt = resnet.Bottleneck(16, 4, stride=2,
                           downsample=nn.Sequential(nn.Conv3d(16, 16, 1, stride=2, bias=False), nn.BatchNorm3d(16)))
    xb = torch.randn(2, 16, 4, 8, 8)
    out.update(sd_np(bt, 'r3b:w:'))
    out['r3b:x'] = xb
    out['r3b:y'] = bt(xb)
    npz('blocks', **out)


# ------------------------------------------------------------------ 3. whole encoders
def gen_models():
    out = {}
    # tiny R(2+1)D-10 (widen 0.125): weights committed; train- and eval-mode outputs + input grad + a weight grad
    torch.manual_seed(21)
    m = resnet2p1d.generate_model(10, widen_factor=0.125)
    out.update(sd_np(m, 'r2t:w:'))
    x = seeded_randn(210, 4, 3, 8, 64, 64).requires_grad_(True)
    m.train()
    y = m(x)
    y.square().sum().backward()
    out.update({'r2t:xspec': spec(210, 4, 3, 8, 64, 64), 'r2t:y_train': y, 'r2t:dx': x.grad,
                'r2t:dw_conv1_s': m.conv1_s.weight.grad, 'r2t:dw_l4_conv2_t': m.layer4[0].conv2_t.weight.grad,
                'r2t:dw_fc': m.fc.weight.grad, 'r2t:dg_bn1_s': m.bn1_s.weight.grad})
    out.update(sd_np(m, 'r2t:after:'))
    m.eval()
    out['r2t:y_eval'] = m(x)
    npz('r2p1d_tiny', **out)

    # seed-equivalence: the oracle builders reproduce the reference's weights from the same seed,
    # so full-size models need only (seed, input, output) fixtures.
    def check_same(ref_ctor, ora_ctor, seed):
        torch.manual_seed(seed)
        a = ref_ctor()
        torch.manual_seed(seed)
        b = ora_ctor()
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys()), 'state-dict keys differ'
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
        return a

    full = {}
    s3d = check_same(s3d_1.S3D, oenc.S3D, 31)
    s3d.train()
    xs = seeded_randn(32, 2, 3, 16, 96, 96)
    s3d.fc = nn.Identity()
    with torch.no_grad():
        ys = s3d(xs)
    full.update({'s3d:seed': 31, 's3d:xspec': spec(32, 2, 3, 16, 96, 96), 's3d:y_train': ys,
                 's3d:rm_base0_bn_s': s3d.base[0].bn_s.running_mean})
    r18 = check_same(lambda: resnet2p1d.generate_model(18), lambda: oenc.R2Plus1D(18), 33)
    r18.train()
    xr = seeded_randn(34, 4, 3, 8, 96, 96)
    r18.fc = nn.Identity()
    with torch.no_grad():
        yr = r18(xr)
    full.update({'r18:seed': 33, 'r18:xspec': spec(34, 4, 3, 8, 96, 96), 'r18:y_train': yr})
    r3 = check_same(lambda: resnet.resnet18(sample_size=96, sample_duration=16),
                    lambda: oenc.R3D(18, 96, 16), 35)
    r3.train()
    x3 = seeded_randn(36, 4, 3, 16, 96, 96)
    with torch.no_grad():
        y3 = r3(x3)
    full.update({'r3d18:seed': 35, 'r3d18:xspec': spec(36, 4, 3, 16, 96, 96), 'r3d18:y_train': y3})
    check_same(lambda: resnet.resnet50(sample_size=224, sample_duration=32), lambda: oenc.R3D(50, 224, 32), 37)
    npz('encoders_seeded', **full)


# ------------------------------------------------------------------ 4. head / queue / loss
def gen_moco():
    out = {}
    torch.manual_seed(41)
    head = ref_head.ProjectHead(24, 16, 'mlp')
    xh = torch.randn(5, 24)
    out.update(sd_np(head, 'head:w:'))
    out['head:x'] = xh
    out['head:y'] = head(xh)

    for tag, (K, D, b, steps) in {'k8': (8, 16, 3, 5), 'k256': (256, 128, 2, 3)}.items():
        torch.manual_seed(42)
        mo = mem_moco.RGBMoCo(D, K=K, T=0.07)
        crit = ref_crit.NCESoftmaxLoss()
        out[tag + ':mem0'] = mo.memory.clone()
        for s in range(steps):
            q = nn.functional.normalize(torch.randn(b, D)).requires_grad_(True)
            k = nn.functional.normalize(torch.randn(b, D))
            logits, labels = mo(q, k)
            loss = crit(logits)
            loss.backward()
            # metric.py:65 uses .view on a transposed tensor: top-5 raises on current torch, top-1 runs
            p1, = ref_accuracy(logits.detach(), labels, topk=(1,))
            out.update({'%s:q%d' % (tag, s): q, '%s:k%d' % (tag, s): k, '%s:logits%d' % (tag, s): logits,
                        '%s:labels%d' % (tag, s): labels, '%s:loss%d' % (tag, s): loss,
                        '%s:dq%d' % (tag, s): q.grad, '%s:mem%d' % (tag, s + 1): mo.memory.clone(),
                        '%s:ptr%d' % (tag, s + 1): mo.index, '%s:prec1_%d' % (tag, s): p1})
    # all_k path: enqueue a gathered batch larger than the local one, wrapping (ptr 6, n 4, K 8)
    torch.manual_seed(43)
    mo = mem_moco.RGBMoCo(16, K=8, T=0.07)
    mo.index = 6
    q = nn.functional.normalize(torch.randn(2, 16))
    k = nn.functional.normalize(torch.randn(2, 16))
    all_k = nn.functional.normalize(torch.randn(4, 16))
    out['allk:mem0'] = mo.memory.clone()
    lg, _ = mo(q, k, all_k=all_k)
    out.update({'allk:q': q, 'allk:k': k, 'allk:all_k': all_k, 'allk:logits': lg, 'allk:mem1': mo.memory.clone(),
                'allk:ptr1': mo.index})
    npz('moco', **out)


# ------------------------------------------------------------------ 5. graph block
def gen_graph():
    out = {}
    for T in (2, 4, 8, 16):
        out['hop:T%d' % T] = ref_tg.TemporalGraph(tem_len=T, max_hop=3).temporal_graph
    torch.manual_seed(51)
    aug = ref_tg.TemporalGraphAug(in_channels=32)
    aug.gcns[0].conv.weight.data.mul_(0.5)
    x = torch.randn(2, 32, 8, 6, 6, requires_grad=True)
    out.update(sd_np(aug, 'aug:w:'))
    hop = ref_tg.TemporalGraph(tem_len=8, max_hop=3).temporal_graph
    sim = aug._get_sim_adj(x)
    pre = aug._parser_temporal_graph(sim, hop)
    torch.manual_seed(52)
    adj = aug._sample_adj_with_rel_ber(pre)
    torch.manual_seed(52)
    u = torch.rand(pre.shape)                      # the uniforms rsample() drew (same seed, same shape)
    y = aug.gcns[0](x, adj)
    dy = torch.randn(y.shape)
    y.backward(dy)
    out.update({'aug:x': x, 'aug:sim': sim, 'aug:pre': pre, 'aug:u': u, 'aug:adj': adj, 'aug:y': y,
                'aug:dy': dy, 'aug:dx': x.grad, 'aug:dw_gcn': aug.gcns[0].conv.weight.grad,
                'aug:dw_gq': aug.g_q[0].weight.grad, 'aug:dw_gk': aug.g_k[0].weight.grad})
    # whole forward with the same seed (hop graph + sim + sample + gcn)
    torch.manual_seed(53)
    with torch.no_grad():
        out['aug:y_full_seed53'] = aug(x)
    torch.manual_seed(53)
    out['aug:u_full_seed53'] = torch.rand(pre.shape)
    npz('graph', **out)


def gen_graph_options():
    """The constructor options of TemporalGraphAug the shipped configs never set (temporal_graph.py:66-129): no sub-sampling,
    average instead of max pooling, BatchNorm behind the similarity convs, and a 3-layer GCN stack with an explicit
    inter_channels (with the default inter_channels=None the reference itself cannot build more than one layer: :95-98 pass
    None as in_features).  Whole-block forward + backward under a fixed seed; the uniforms rsample() drew are stored."""
    out = {}
    variants = {'nosub': dict(sub_sample=False), 'avg': dict(max_pool=False), 'bn': dict(bn_layer=True),
                'gcn3': dict(inter_channels=8, num_gcn_layers=3), 'bias': dict(bias=True)}
    for i, (tag, kw) in enumerate(variants.items()):
        torch.manual_seed(70 + i)
        aug = ref_tg.TemporalGraphAug(in_channels=16, **kw)
        aug.train()
        for g in aug.gcns:
            g.conv.weight.data.mul_(0.5)
        x = torch.randn(3, 16, 4, 6, 6, requires_grad=True)
        out.update(sd_np(aug, tag + ':w:'))
        torch.manual_seed(90 + i)
        y = aug(x)
        torch.manual_seed(90 + i)
        u = torch.rand(3, 4, 4)                      # the uniforms rsample() drew (same seed, same shape)
        dy = torch.randn(y.shape)
        y.backward(dy)
        out.update({tag + ':x': x, tag + ':u': u, tag + ':y': y, tag + ':dy': dy, tag + ':dx': x.grad})
        for n, prm in aug.named_parameters():
            out[tag + ':g:' + n] = prm.grad
        if kw.get('bn_layer'):
            out.update(sd_np(aug, tag + ':after:'))  # running statistics after the forward
    npz('graph_options', **out)


def gen_input():
    """VideoNormalize + VideoToTensor of the reference on random uint8 frames (consistency_transforms.py:11-65)."""
    import importlib.util
    import types

    class _Inert(types.ModuleType):                               # shim 6
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return 0
    for name in ('cv2', 'albumentations', 'albumentations.augmentations', 'albumentations.augmentations.functional'):
        sys.modules.setdefault(name, _Inert(name))
        if '.' in name:                                           # `import a.b.c as F` walks the attributes
            parent, leaf = name.rsplit('.', 1)
            setattr(sys.modules[parent], leaf, sys.modules[name])
    spec_ = importlib.util.spec_from_file_location('ref_consistency_transforms',
                                                   os.path.join(REF, 'lib', 'data', 'transform', 'consistency_transforms.py'))
    ct = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(ct)
    rng = np.random.RandomState(7)
    out = {}
    for tag, (T, H, W), mean, std in (('a', (4, 9, 11), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
                                      ('b', (3, 8, 8), (0.5, 0.45, 0.4), (0.25, 0.3, 0.2))):
        frames = [rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8) for _ in range(T)]
        frames[0][0, 0] = (0, 255, 128)                            # the extremes are in
        norm = ct.VideoNormalize(mean=mean, std=std)(frames)
        ten = ct.VideoToTensor(backbone_type='3D')(norm)
        assert ten.dtype == torch.float32 and tuple(ten.shape) == (3, T, H, W)
        out.update({tag + ':frames': np.stack(frames), tag + ':mean': np.array(mean), tag + ':std': np.array(std),
                    tag + ':norm0': norm[0], tag + ':tensor': ten})
    npz('input', **out)


# ------------------------------------------------------------------ 6. SimSiam / step traces / solver
def gen_steps():
    out = {}
    # SimSiam loss + a few grads on the tiny encoder
    torch.manual_seed(61)
    cfg = cfg_for('R2P1D10T', 'simsiam', feat_dim=32)
    model, ema = ref_build.create_visual_model(cfg)
    assert ema is None
    model.train()
    out.update(sd_np(model, 'ss:w:'))
    imgs = seeded_randn(610, 8, 6, 8, 48, 48)
    loss = model(imgs)
    loss.backward()
    sm = model.model
    out.update({'ss:xspec': spec(610, 8, 6, 8, 48, 48), 'ss:loss': loss, 'ss:dw_pred_l2': sm.prediction.l2.weight.grad,
                'ss:dw_proj_l1': sm.projection.l1[0].weight.grad,
                'ss:dg_proj_l3_bn': sm.projection.l3[1].weight.grad,
                'ss:dw_conv1_s': sm.encoder.base_model.conv1_s.weight.grad})

    # MoCo: two full iterations following tools/train_video_contrast_dis.py:395-454 with the
    # reference's model / queue / criterion / optimiser-builder classes (single process, so
    # ShuffleBN is a local permutation and all_k == the shuffled k batch).
    torch.manual_seed(62)
    cfg = cfg_for('R2P1D10T', 'moco', feat_dim=32)
    model, ema = ref_build.create_visual_model(cfg)
    for p1, p2 in zip(model.parameters(), ema.parameters()):          # _momentum_update(m=0), :146
        p2.data.mul_(0).add_(p1.detach().data, alpha=1)
    contrast = mem_moco.RGBMoCo(32, K=20, T=0.07)
    crit = ref_crit.NCESoftmaxLoss()
    opt = ref_solver.make_optimizer(cfg, model)
    sched = ref_solver.make_lr_scheduler(cfg, opt)
    out.update(sd_np(model, 'mo:w:'))
    # key encoder == query encoder at this point (params copied, buffers at their defaults)
    out['mo:mem0'] = contrast.memory.clone()
    out['mo:group_lr'] = np.array([g['lr'] for g in opt.param_groups])
    out['mo:group_wd'] = np.array([g['weight_decay'] for g in opt.param_groups])
    out['mo:group_names'] = np.array([n for n, _ in model.named_parameters()])
    model.train()
    ema.eval()
    for m in ema.modules():
        if 'BatchNorm' in m.__class__.__name__:
            m.train()
    for it in range(3):
        images = seeded_randn(620 + it, 8, 6, 8, 48, 48)
        shuffle_ids = torch.randperm(8)
        x1, x2 = torch.chunk(images, 2, dim=1)
        reverse_ids = torch.argsort(shuffle_ids)
        with torch.no_grad():
            k_sh = ema(x2[shuffle_ids])
        all_k = k_sh
        feat_k = all_k[reverse_ids]
        opt.zero_grad()
        feat_q = model(x1)
        logits, labels = contrast(feat_q, feat_k, all_k=all_k)
        loss = crit(logits)
        loss.backward()
        opt.step()
        for p1, p2 in zip(model.parameters(), ema.parameters()):
            p2.data.mul_(0.999).add_(p1.detach().data, alpha=1 - 0.999)
        out.update({'mo:xspec%d' % it: spec(620 + it, 8, 6, 8, 48, 48), 'mo:shuffle%d' % it: shuffle_ids, 'mo:loss%d' % it: loss,
                    'mo:logits%d' % it: logits, 'mo:q%d' % it: feat_q, 'mo:k%d' % it: feat_k})
    out.update(sd_np(model, 'mo:after:'))
    ek = ema.state_dict()
    for key in ('model.encoder.base_model.conv1_s.weight', 'model.encoder.base_model.bn1_s.running_mean',
                'model.encoder.base_model.bn1_s.running_var', 'model.encoder.base_model.layer4.0.conv2_t.weight',
                'model.encoder.base_model.layer2.0.downsample.1.running_var',
                'model.proj_head.head.0.bias', 'model.proj_head.head.2.weight'):
        out['mo:afterk:' + key] = ek[key].clone()
    out['mo:mem3'] = contrast.memory.clone()
    out['mo:ptr3'] = contrast.index
    # LR schedule (reference class, runnable thanks to shim 5)
    lrs = []
    for e in range(0, 200):
        lrs.append(sched.get_last_lr()[0] if hasattr(sched, 'get_last_lr') else opt.param_groups[0]['lr'])
        opt.step()
        sched.step()
    out['lr:weights_epoch0_199'] = np.array(lrs)
    npz('steps', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['ops', 'blocks', 'models', 'moco', 'graph', 'graph_options', 'input', 'steps']
    for w in which:
        globals()['gen_' + w]()
