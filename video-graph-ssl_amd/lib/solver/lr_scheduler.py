"""Warm-up + step / poly / cosine LR schedule (reference: lib/solver/lr_scheduler.py:8-69).
Stand-alone (no torch.optim dependency): works on any object with ``param_groups`` whose groups
carry ``initial_lr``.  The reference class crashes on Python >= 3.10 (collections.Iterable, :54);
this is its intended arithmetic.  Like torch's _LRScheduler, constructing it applies epoch 0."""
import math
from bisect import bisect_right


class WarmupMultiStepLR(object):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method='linear', mode='step', last_epoch=-1, max_epochs=100):
        if not isinstance(milestones, int) and list(milestones) != sorted(milestones):
            raise ValueError('Milestones should be a list of increasing integers. Got {}'.format(milestones))
        if warmup_method not in ('constant', 'linear'):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got {}".format(warmup_method))
        if mode not in ('step', 'poly', 'cos'):
            raise NotImplementedError('currently not suported: {} scheduler'.format(mode))
        self.optimizer, self.milestones, self.gamma, self.mode = optimizer, milestones, gamma, mode
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.max_epochs = max_epochs
        self.base_lrs = [g.setdefault('initial_lr', g['lr']) for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def factor(self, epoch):
        w = 1.0
        if epoch < self.warmup_iters:
            if self.warmup_method == 'constant':
                w = self.warmup_factor
            else:
                a = float(epoch) / self.warmup_iters
                w = self.warmup_factor * (1 - a) + a
        if self.mode == 'step':
            if isinstance(self.milestones, int):
                f = self.gamma ** (epoch // self.milestones)
            else:
                f = self.gamma ** bisect_right(list(self.milestones), epoch)
        elif self.mode == 'poly':
            f = pow((1 - 1.0 * epoch / self.max_epochs), 0.9)
        else:
            f = 0.5 * (1. + math.cos(1.0 * epoch / self.max_epochs * math.pi))
        return w * f

    def get_lr(self):
        f = self.factor(self.last_epoch)
        return [b * f for b in self.base_lrs]

    def get_last_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g['lr'] = lr

    def state_dict(self):
        return {'last_epoch': self.last_epoch}

    def load_state_dict(self, sd):
        self.step(sd['last_epoch'])
