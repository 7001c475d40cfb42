"""Per-iteration metrics of the pre-training loop (reference: lib/evaluation/metric.py).

``accuracy(output, target, topk)`` and ``AverageMeter`` keep the reference's signatures (:9-24, :44-67), so the trainer's
``prec1, prec5 = accuracy(output, labels, topk=(1, 5))`` (tools/train_video_contrast_dis.py:428) drops in.  What differs is how
the numbers are produced: "the label is among the top k" <=> fewer than k other columns score >= the label's column, so a
counting kernel (gca_rank_ge) replaces the top-k sort, and the result stays on the device (the reference's three
``.item()`` calls per iteration, :429-431, are the caller's choice here).  For InfoNCE logits (label 0) the count already
comes out of the logits kernel: ``accuracy_from_rank``.  MAPMetric / map() belong to the downstream multi-label tools and
are out of scope."""
import torch

from ...engine import ops


class AverageMeter(object):
    """Computes and stores the average and current value (metric.py:9-24).  `val` may be a Python number or a 1-element
    device tensor; tensors are accumulated on the device (no sync) and `avg` is then a tensor too."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count


def rank_ge(output, target):
    """(b, ncol) fp32 device logits, (b,) int64 labels -> (b,) int32: columns other than the label scoring >= the label."""
    if output.dim() != 2 or target.dim() != 1 or target.shape[0] != output.shape[0]:
        raise ValueError('accuracy: single-label (b, ncol) logits and (b,) labels are supported '
                         '(the multi-label branch of metric.py:53-62 is a downstream-only path)')
    b, ncol = output.shape
    out = torch.empty(b, dtype=torch.int32, device=output.device)
    ops.H.call('gca_rank_ge', ops.ptr(output.contiguous()), ops.ptr(target.to(torch.long).contiguous()), b, ncol, ops.ptr(out),
               ops.stream())
    return out


def accuracy(output, target, topk=(1,)):
    """Computes the precision@k for the specified values of k (metric.py:44-67): list of 1-element tensors, percent."""
    r = rank_ge(output, target)
    b = target.size(0)
    return [(r < k).sum().reshape(1).float().mul_(100.0 / b) for k in topk]


def accuracy_from_rank(rank_ge, topk=(1, 5)):
    """rank_ge: (b,) int32 device tensor -> list of 1-element device tensors (percent)."""
    b = rank_ge.numel()
    return [(rank_ge < k).sum().reshape(1).float() * (100.0 / b) for k in topk]
