"""Shared parity helpers: run the SAME weights / inputs through the HIP product path and the CPU
oracle and report relative errors.  Used by tests/ and by __graft_entry__.smoke().

How gradients are compared.  Forward quantities (features, logits, loss, BN running statistics) are
continuous in the inputs and are held to the 1e-3 max-norm bar directly.  Gradients are not: an
activation that lands within one ulp of zero takes the other side of the ReLU in a different fp32
summation order (measured: about one such element per few training steps of the tiny model, for the
HIP path and for the fp32 CPU oracle alike when each is compared with an fp64 run), and that single
flipped mask changes the affected channel's gradient sums by up to several per cent.  So gradients
are compared per parameter tensor against an fp64 run of the oracle, next to the fp32 CPU oracle's
own error, and the assertion is on the distribution (median / 95th percentile / worst) -- see
check_grad_errors()."""
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


def oracle_moco(backbone, feat_dim, K, T, state, mem0, lr_factor, double=False):
    model, ema = owrap.create_visual_model(backbone, T, feat_dim, 'mlp', 'moco')
    model.load_state_dict(state)
    ema.load_state_dict(state)
    contrast = omoco.RGBMoCo(feat_dim, K=K, T=0.07)
    contrast.memory.copy_(mem0)
    if double:
        model.double(), ema.double(), contrast.double()
    opt = omoco.make_optimizer(model, 0.06, 0.9, 5e-4)
    for g in opt.param_groups:
        g['lr'] *= lr_factor
    model.train()
    omoco.set_key_encoder_mode(ema)
    return model, ema, contrast, opt


def _f32(sd):
    return {k: (v.float() if v.dtype.is_floating_point else v) for k, v in sd.items()}


def run_moco_parity(pkg, device, backbone, images_list, shuffles, feat_dim=128, K=256, T=8, use_graph=False,
                    seed=123, with_cpu32=True):
    """MoCoTrainer (HIP) vs the oracle in fp64 (ground truth) and fp32 (the reference's own precision), same
    init / clips / permutations.  Every step starts all runs from the fp64 state ("teacher forcing"), so
    a per-step error is never the accumulation of earlier ones.  Returns a list (one dict per step):
      fwd      : {loss, logits, q, k_queue}  max-norm relative errors of the HIP forward vs fp64
      grad_hip : per-parameter relative error of the HIP gradients vs fp64
      grad_cpu : the same for the fp32 CPU oracle (how well fp32 CAN do on this data)
      post     : params / key params / running stats / queue after the step vs fp64
    """
    cfg = make_cfg(pkg, backbone, 'moco', feat_dim, K, T)
    tr = pkg.MoCoTrainer(cfg, device, use_graph=use_graph, seed=seed)
    state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    mem0 = tr.contrast.memory.detach().cpu().clone()
    f0 = omoco.warmup_multistep_factor(0, (80, 120, 160), 0.1, 0.01, 10)
    m64, e64, c64, o64 = oracle_moco(backbone, feat_dim, K, T, state, mem0, f0, double=True)
    m32 = e32 = c32 = o32 = None
    if with_cpu32:
        m32, e32, c32, o32 = oracle_moco(backbone, feat_dim, K, T, state, mem0, f0)
    crit = omoco.NCESoftmaxLoss()
    steps = []
    for it, (images, sh) in enumerate(zip(images_list, shuffles)):
        tr.model.load_state_dict(_f32(m64.state_dict()))
        tr.model_ema.load_state_dict(_f32(e64.state_dict()))
        tr.contrast.memory.copy_(c64.memory.float())
        tr.optimizer.buf.copy_(torch.cat([torch.nn.functional.pad(
            o64.state[p]['momentum_buffer'].reshape(-1).float() if p in o64.state and 'momentum_buffer' in o64.state[p]
            else torch.zeros(p.numel()), (0, (-p.numel()) % 256)) for p in m64.parameters()]))
        if with_cpu32:
            m32.load_state_dict(_f32(m64.state_dict()))
            e32.load_state_dict(_f32(e64.state_dict()))
            c32.memory.copy_(c64.memory.float())
            c32.index = c64.index
            for p32, p64 in zip(m32.parameters(), m64.parameters()):
                if p64 in o64.state and 'momentum_buffer' in o64.state[p64]:
                    o32.state[p32]['momentum_buffer'] = o64.state[p64]['momentum_buffer'].float().clone()
        out = tr.train_step(images.to(device), shuffle_ids=sh)
        r64 = omoco.moco_train_step(m64, e64, c64, crit, o64, images.double(), 0.999, shuffle_ids=sh)
        torch.cuda.synchronize()
        rec = {'fwd': {'loss': rel(out['loss'].reshape(()), r64['loss']), 'logits': rel(out['logits'], r64['logits']),
                       'q': rel(out['q'], r64['q'])}}
        g64 = {n: p.grad for n, p in m64.named_parameters()}
        rec['grad_hip'] = {n: rel(p.grad, g64[n]) for n, p in tr.model.named_parameters() if g64[n].abs().max() > 0}
        if with_cpu32:
            omoco.moco_train_step(m32, e32, c32, crit, o32, images, 0.999, shuffle_ids=sh)
            rec['grad_cpu'] = {n: rel(p.grad, g64[n]) for n, p in m32.named_parameters() if g64[n].abs().max() > 0}
        sd, sk = tr.model.state_dict(), tr.model_ema.state_dict()
        pnames = set(n for n, _ in m64.named_parameters())
        fl = lambda ref, want_param: {k: v for k, v in ref.state_dict().items() if v.dtype.is_floating_point
                                      and v.abs().max() > 0 and ((k in pnames) == want_param)}
        # BN running statistics and the queue are forward quantities (strict bar); parameters that start at
        # zero (biases) ARE the scaled gradient after a step, so updated parameters get the gradient bar.
        rec['post'] = {'buffers': max(max(rel(sd[k], v) for k, v in fl(m64, False).items()),
                                      max(rel(sk[k], v) for k, v in fl(e64, False).items())),
                       'queue': rel(tr.contrast.memory, c64.memory),
                       'ptr': abs(int(tr.ptr_dev) - c64.index) + abs(tr.contrast.index - c64.index)}
        rec['post_params'] = {k: rel(sd[k], v) for k, v in fl(m64, True).items()}
        rec['post_key_params'] = {k: rel(sk[k], v) for k, v in fl(e64, True).items()}
        steps.append(rec)
    return steps


def _pct(vals, q):
    s = sorted(vals)
    return s[min(len(s) - 1, int(q * len(s)))]


def check_grad_errors(errs, what='gradients'):
    """Distribution bar for per-parameter gradient errors (see module docstring)."""
    v = list(errs.values())
    med, p95, worst = _pct(v, 0.5), _pct(v, 0.95), max(v)
    assert med < 2e-4, '%s: median rel err %.2e' % (what, med)
    assert worst < 1e-1, '%s: worst rel err %.2e (%s)' % (what, worst, max(errs, key=errs.get))
    return med, p95, worst


def run_tiny_moco_parity(pkg, device, steps=1):
    """smoke(): tiny R(2+1)D-10 MoCo iteration(s); returns the worst forward / post-step error."""
    register_tiny(pkg)
    g = torch.Generator().manual_seed(99)
    imgs = [torch.randn(8, 6, 8, 48, 48, generator=g) for _ in range(steps)]       # >= 32 values per BN channel
    shs = [torch.randperm(8, generator=g) for _ in range(steps)]
    worst = 0.0
    for rec in run_moco_parity(pkg, device, 'R2P1D10T', imgs, shs, feat_dim=32, K=20, T=8, with_cpu32=False):
        assert rec['post'].pop('ptr') == 0
        check_grad_errors(rec['grad_hip'])
        worst = max(worst, max(rec['fwd'].values()), rec['post']['queue'])
    return worst
