"""Shared parity helpers: run the SAME weights / inputs through the HIP product path and the CPU
oracle and report max relative error.  Used by tests/ and by __graft_entry__.smoke()."""
import copy
from types import SimpleNamespace as NS

import torch

from oracle import encoders as oenc, moco as omoco, wrappers as owrap


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make_cfg(pkg, backbone, mem_type='moco', feat_dim=128, K=256, T=8, aug=False, **solver):
    cfg = pkg.get_defaults()
    cfg.merge_from_list(['MODEL.BACKBONE', backbone, 'MODEL.BACKBONE_TYPE', '3D', 'MODEL.DROPOUT', 0.0,
                         'MODEL.PRETRAINED', False, 'MODEL.AUG_FLAG', aug, 'INPUT.VIDEO_LENGTH', T,
                         'CONTRAST.MEM_TYPE', mem_type, 'CONTRAST.NCE_K', K, 'CONTRAST.NCE_T', 0.07,
                         'CONTRAST.ALPHA', 0.999, 'CROSS.FEAT_DIM', feat_dim, 'SOLVER.BASE_LR', 0.06,
                         'SOLVER.LR_SCHEDULER', 'step', 'SOLVER.STEPS', [80, 120, 160], 'SOLVER.WARMUP_FACTOR', 0.01,
                         'SOLVER.WARMUP_ITERS', 10, 'SOLVER.MAX_EPOCHS', 200])
    for k, v in solver.items():
        setattr(cfg.SOLVER, k, v)
    return cfg


def register_tiny(pkg):
    """Tiny R(2+1)D-10 (widen 0.125) in both registries -- the model of tests/golden/steps.npz."""
    bb = pkg.lib.modeling.backbone.backbone_3d
    bb.register('R2P1D10T', lambda: bb.resnet2p1d.generate_model(10, widen_factor=0.125))
    oenc.BACKBONES['R2P1D10T'] = lambda: oenc.R2Plus1D(10, widen_factor=0.125)


def oracle_moco(backbone, feat_dim, K, T, state, mem0, lr_factor):
    model, ema = owrap.create_visual_model(backbone, T, feat_dim, 'mlp', 'moco')
    model.load_state_dict(state)
    ema.load_state_dict(state)
    contrast = omoco.RGBMoCo(feat_dim, K=K, T=0.07)
    contrast.memory.copy_(mem0)
    opt = omoco.make_optimizer(model, 0.06, 0.9, 5e-4)
    for g in opt.param_groups:
        g['lr'] *= lr_factor
    model.train()
    omoco.set_key_encoder_mode(ema)
    return model, ema, contrast, opt


def run_moco_parity(pkg, device, backbone, images_list, shuffles, feat_dim=128, K=256, T=8, use_graph=False,
                    seed=123):
    """Same init, same clips, same permutations through MoCoTrainer (HIP) and the oracle step; returns a
    dict of max relative errors (loss, logits, q, params after the last step, key params, queue)."""
    cfg = make_cfg(pkg, backbone, 'moco', feat_dim, K, T)
    tr = pkg.MoCoTrainer(cfg, device, use_graph=use_graph, seed=seed)
    state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    mem0 = tr.contrast.memory.detach().cpu().clone()
    model, ema, contrast, opt = oracle_moco(backbone, feat_dim, K, T, state, mem0,
                                            omoco.warmup_multistep_factor(0, (80, 120, 160), 0.1, 0.01, 10))
    crit = omoco.NCESoftmaxLoss()
    errs = {}
    for it, (images, sh) in enumerate(zip(images_list, shuffles)):
        out = tr.train_step(images.to(device), shuffle_ids=sh)
        ref = omoco.moco_train_step(model, ema, contrast, crit, opt, images, 0.999, shuffle_ids=sh)
        errs['loss%d' % it] = rel(out['loss'].reshape(()), ref['loss'])
        errs['logits%d' % it] = rel(out['logits'], ref['logits'])
        errs['q%d' % it] = rel(out['q'], ref['q'])
    torch.cuda.synchronize()
    sd = tr.model.state_dict()
    errs['params'] = max(rel(sd[k].float(), v.float()) for k, v in model.state_dict().items()
                         if v.dtype.is_floating_point and v.abs().max() > 0)
    ograds = {n: p.grad for n, p in model.named_parameters()}
    gerr = {}
    for n, p in tr.model.named_parameters():
        if ograds[n].abs().max() > 1e-12:
            gerr[n] = rel(p.grad, ograds[n])
    errs['grads'] = max(gerr.values())
    errs['_worst_grad'] = max(gerr, key=gerr.get)
    sk = tr.model_ema.state_dict()
    errs['key_params'] = max(rel(sk[k].float(), v.float()) for k, v in ema.state_dict().items()
                             if v.dtype.is_floating_point and v.abs().max() > 0)
    errs['queue'] = rel(tr.contrast.memory, contrast.memory)
    errs['ptr'] = abs(int(tr.ptr_dev) - contrast.index) + abs(tr.contrast.index - contrast.index)
    return errs


def run_tiny_moco_parity(pkg, device, steps=1):
    register_tiny(pkg)
    g = torch.Generator().manual_seed(99)
    imgs = [torch.randn(2, 6, 8, 32, 32, generator=g) for _ in range(steps)]
    shs = [torch.randperm(2, generator=g) for _ in range(steps)]
    errs = run_moco_parity(pkg, device, 'R2P1D10T', imgs, shs, feat_dim=32, K=16, T=8)
    assert errs.pop('ptr') == 0
    errs.pop('_worst_grad')
    return max(errs.values())
