"""Optimiser / LR-schedule factories (reference: lib/solver/build.py:24-71).

``make_optimizer(cfg, model)`` returns a HipSGD whose ``param_groups`` mirror the reference's
(one group PER PARAMETER; names containing "bias" get lr * BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS),
but whose ``step()`` is ONE multi-tensor kernel over the flat parameter arena instead of one
launch per tensor."""
import torch

from .lr_scheduler import WarmupMultiStepLR
from ...engine import ops
from ...engine.arena import CHUNK, arena_of


class HipSGD(object):
    """torch.optim.SGD semantics (momentum, weight decay, optional nesterov; dampening 0) on the arena."""

    def __init__(self, model, groups, momentum=0.9, nesterov=False):
        self.arena = arena_of(model)
        if len(groups) != len(self.arena.params):
            raise ValueError('one param group per parameter expected')
        self.param_groups = groups
        for g in groups:
            g.setdefault('initial_lr', g['lr'])
            g.setdefault('momentum', momentum)
            g.setdefault('nesterov', nesterov)
        self.momentum, self.nesterov = momentum, nesterov
        dev = self.arena.flat.device
        self.buf = torch.zeros_like(self.arena.flat)          # momentum buffers (zero == "first step" semantics)
        self.chunk_lr = torch.zeros(self.arena.total // CHUNK, dtype=torch.float32, device=dev)
        self.chunk_wd = torch.zeros_like(self.chunk_lr)
        self._uploaded = None
        self._sync_tables()

    def _sync_tables(self):
        key = tuple((g['lr'], g['weight_decay']) for g in self.param_groups)
        if key != self._uploaded:
            self.chunk_lr.copy_(self.arena.chunk_table([g['lr'] for g in self.param_groups]), non_blocking=False)
            self.chunk_wd.copy_(self.arena.chunk_table([g['weight_decay'] for g in self.param_groups]))
            self._uploaded = key

    def zero_grad(self, set_to_none=False):
        self.arena.rebind()
        ops.fill(self.arena.grad, 0.0)

    def step(self, grad_clip=None):
        """grad_clip: optional device (norm, coef, skip, -) record from clip_grad_norm() / ops.grad_unscale_clip(): the
        coefficient is applied to every gradient inside the update kernel; a set skip flag (non-finite gradient under
        loss scaling) leaves parameters and momentum untouched."""
        self._sync_tables()
        ops.sgd_step(self.arena.flat, self.arena.grad, self.buf, self.chunk_lr, self.chunk_wd, 1.0,
                     self.momentum, self.nesterov, grad_clip)

    def clip_grad_norm(self, max_norm, out=None):
        """torch.nn.utils.clip_grad_norm_ as the reference applies it between backward and step
        (tools/train_video_contrast_dis.py:420-423), over the flat gradient arena, without a host sync: returns the
        device pair (total_norm, coef); pass it to step(grad_clip=...) -- the gradients themselves are left unscaled."""
        return ops.grad_clip_coef(self.arena.grad, max_norm, out)

    def state_dict(self):
        """torch.optim.SGD's wire format (what the reference checkpoints hold, tools/...dis.py:274-286): per-parameter
        `momentum_buffer`s keyed by parameter index, one param group per parameter.  The buffers are copies."""
        a = self.arena
        state = {i: {'momentum_buffer': self.buf[o:o + n].view(p.shape).clone()}
                 for i, (o, n, p) in enumerate(zip(a.offsets, a.sizes, a.params))}
        groups = [dict({k: v for k, v in g.items() if k != 'params'}, params=[i]) for i, g in enumerate(self.param_groups)]
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        a = self.arena
        if 'momentum_buffer' in sd:                       # flat layout written by earlier builds of this package
            self.buf.copy_(sd['momentum_buffer'])
        else:
            self.buf.zero_()
            for i, (o, n) in enumerate(zip(a.offsets, a.sizes)):
                st = sd['state'].get(i, sd['state'].get(str(i)))
                if st is not None and st.get('momentum_buffer') is not None:
                    self.buf[o:o + n].copy_(st['momentum_buffer'].reshape(-1))
        if len(sd['param_groups']) != len(self.param_groups):
            raise ValueError('optimizer state has %d param groups, this model needs %d'
                             % (len(sd['param_groups']), len(self.param_groups)))
        for g, s_ in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in s_.items() if k != 'params'})
        # the update kernel takes ONE momentum / nesterov setting: re-read it from the loaded groups
        moms = set(float(g.get('momentum', self.momentum)) for g in self.param_groups)
        nest = set(bool(g.get('nesterov', self.nesterov)) for g in self.param_groups)
        if len(moms) != 1 or len(nest) != 1:
            raise NotImplementedError('per-group momentum / nesterov settings are not supported by the fused SGD kernel')
        self.momentum, self.nesterov = moms.pop(), nest.pop()
        self._uploaded = None


def clip_value_of(cfg):
    """SOLVER.CLIP_GRADIENT: 'none' (every shipped YAML) or the max 2-norm (tools/train_video_contrast_dis.py:420)."""
    v = getattr(cfg.SOLVER, 'CLIP_GRADIENT', 'none')
    if isinstance(v, str):
        if v.lower() == 'none':
            return None
        v = float(v)
    v = float(v)
    if not v > 0:
        raise ValueError('SOLVER.CLIP_GRADIENT must be "none" or a positive max-norm (got %r)' % (v,))
    return v


def make_optimizer(cfg, model):
    if cfg.SOLVER.USE_TRICK:
        raise NotImplementedError('SOLVER.USE_TRICK (TSN-style policies) is a downstream-only path')
    if cfg.SOLVER.OPTIMIZER_NAME != 'SGD':
        raise NotImplementedError('only SGD is used by the pre-training configs')
    groups = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            raise NotImplementedError('frozen parameters are not on the pre-training path')
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        if 'bias' in key:
            lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
        groups.append({'params': [value], 'lr': lr, 'weight_decay': wd, 'name': key})
    return HipSGD(model, groups, momentum=cfg.SOLVER.MOMENTUM, nesterov=cfg.SOLVER.NESTEROV)


def make_lr_scheduler(cfg, optimizer):
    return WarmupMultiStepLR(optimizer=optimizer, milestones=cfg.SOLVER.STEPS, gamma=cfg.SOLVER.GAMMA,
                             warmup_factor=cfg.SOLVER.WARMUP_FACTOR, warmup_iters=cfg.SOLVER.WARMUP_ITERS,
                             warmup_method=cfg.SOLVER.WARMUP_METHOD, mode=cfg.SOLVER.LR_SCHEDULER,
                             max_epochs=cfg.SOLVER.MAX_EPOCHS)
