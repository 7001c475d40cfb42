"""Oracle MoCo queue / InfoNCE / optimiser-side pieces.  Test infrastructure -- see oracle/__init__.py.

Follows lib/memory/mem_moco.py, lib/memory/criterion.py, lib/solver/{build,lr_scheduler}.py,
lib/evaluation/metric.py and the step sequence of tools/train_video_contrast_dis.py:395-454.
Device-agnostic (the reference hard-codes .cuda(), mem_moco.py:25,78 / criterion.py:43).
"""
from bisect import bisect_right

import torch
import torch.nn as nn
import torch.nn.functional as F


class RGBMoCo(nn.Module):
    """mem_moco.py:6-88.  memory (K,D) buffer, row-normalised randn at init (:57-58);
    `index` is a plain attribute (:12)."""

    def __init__(self, n_dim, K=65536, T=0.07):
        super().__init__()
        self.K, self.T, self.index = K, T, 0
        self.register_buffer('memory', F.normalize(torch.randn(K, n_dim)))

    def compute_logit(self, q, k, queue):
        """:29-49: [q.k | q @ queue^T] / T."""
        pos = (q * k).sum(1, keepdim=True)             # bmm of (b,1,D)x(b,D,1)
        neg = torch.mm(queue, q.t()).t()
        return torch.cat((pos, neg), dim=1) / self.T

    def forward(self, q, k, q_jig=None, all_k=None):
        """:60-88: logits from the PRE-enqueue snapshot (:72), then enqueue all_k (or k) at
        (arange + index) % K (:17-27), then advance index (:14-15)."""
        k = k.detach()
        queue = self.memory.clone().detach()
        logits = self.compute_logit(q, k, queue)
        logits_jig = self.compute_logit(q_jig, k, queue) if q_jig is not None else None
        labels = torch.zeros(q.size(0), dtype=torch.long, device=q.device)
        all_k = k if all_k is None else all_k
        with torch.no_grad():
            ids = torch.fmod(torch.arange(all_k.size(0), device=q.device) + self.index, self.K).long()
            self.memory.index_copy_(0, ids, all_k)
        self.index = (self.index + all_k.size(0)) % self.K
        if q_jig is not None:
            return logits, logits_jig, labels
        return logits, labels


class NCESoftmaxLoss(nn.Module):
    """criterion.py:34-45: CrossEntropy(logits, label 0), mean over rows."""

    def forward(self, x):
        return F.cross_entropy(x, torch.zeros(x.size(0), dtype=torch.long, device=x.device))


def accuracy(output, target, topk=(1,)):
    """lib/evaluation/metric.py:44-67 (single-label branch)."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True) * (100.0 / target.size(0)) for k in topk]


def momentum_update(model, model_ema, m):
    """tools/train_video_contrast_dis.py:177-180: p_k = m*p_k + (1-m)*p_q, parameters only."""
    with torch.no_grad():
        for p1, p2 in zip(model.parameters(), model_ema.parameters()):
            p2.mul_(m).add_(p1.detach(), alpha=1 - m)


def sgd_param_groups(model, base_lr=0.06, weight_decay=5e-4, bias_lr_factor=2.0, weight_decay_bias=0.0):
    """lib/solver/build.py:24-59 (USE_TRICK False): ONE group per parameter; names containing
    'bias' get lr*BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS (defaults.py: 2, 0)."""
    groups = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if 'bias' in name:
            groups.append({'params': [p], 'lr': base_lr * bias_lr_factor, 'weight_decay': weight_decay_bias})
        else:
            groups.append({'params': [p], 'lr': base_lr, 'weight_decay': weight_decay})
    return groups


def make_optimizer(model, base_lr=0.06, momentum=0.9, weight_decay=5e-4, nesterov=False,
                   bias_lr_factor=2.0, weight_decay_bias=0.0):
    return torch.optim.SGD(sgd_param_groups(model, base_lr, weight_decay, bias_lr_factor, weight_decay_bias),
                           momentum=momentum, nesterov=nesterov)


def warmup_multistep_factor(epoch, milestones=(80, 120, 160), gamma=0.1, warmup_factor=0.01,
                            warmup_iters=10, warmup_method='linear'):
    """lib/solver/lr_scheduler.py:41-69, mode 'step' with a milestone list (the reference's
    collections.Iterable check crashes on Python>=3.10; this is the intended arithmetic)."""
    w = 1.0
    if epoch < warmup_iters:
        if warmup_method == 'constant':
            w = warmup_factor
        elif warmup_method == 'linear':
            a = float(epoch) / warmup_iters
            w = warmup_factor * (1 - a) + a
    return w * gamma ** bisect_right(list(milestones), epoch)


def set_key_encoder_mode(model_ema):
    """tools/...dis.py:383-389: key encoder in eval mode but every BatchNorm in train mode."""
    model_ema.eval()
    for m in model_ema.modules():
        if 'BatchNorm' in m.__class__.__name__:
            m.train()


def moco_train_step(model, model_ema, contrast, criterion, optimizer, images, alpha=0.999,
                    shuffle_ids=None, clip_gradient=None):
    """One single-process iteration of _train_moco (tools/...dis.py:395-454).  With one
    process ShuffleBN (:189-231) reduces to: permute the key batch, encode, un-permute; all_k == k.
    Returns dict(loss, logits, q, k, prec1, prec5)."""
    x1, x2 = torch.chunk(images, 2, dim=1)
    b = x2.size(0)
    if shuffle_ids is None:
        shuffle_ids = torch.randperm(b)
    reverse_ids = torch.argsort(shuffle_ids)
    with torch.no_grad():
        k_shuf = model_ema(x2[shuffle_ids])
    all_k = k_shuf                                  # world size 1: gather == identity (:222)
    feat_k = all_k[reverse_ids]                     # :225-229
    all_k = k_shuf                                  # NOTE: all_k stays in SHUFFLED order (:222,231)
    optimizer.zero_grad()
    feat_q = model(x1)
    logits, labels = contrast(feat_q, feat_k, all_k=all_k)
    loss = criterion(logits)
    loss.backward()
    if clip_gradient is not None:                   # SOLVER.CLIP_GRADIENT != 'none' (:420-423)
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_gradient)
    optimizer.step()
    prec1, prec5 = accuracy(logits.detach(), labels, topk=(1, 5))
    momentum_update(model, model_ema, alpha)
    return dict(loss=loss.detach(), logits=logits.detach(), q=feat_q.detach(), k=feat_k, all_k=all_k,
                prec1=prec1, prec5=prec5)


def simsiam_train_step(model, optimizer, images, clip_gradient=None):
    """_train_simsiam (tools/...dis.py:479-523)."""
    optimizer.zero_grad()
    loss = model(images)
    loss.backward()
    if clip_gradient is not None:                   # :496-499
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_gradient)
    optimizer.step()
    return dict(loss=loss.detach())
