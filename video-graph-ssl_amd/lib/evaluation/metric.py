"""top-k precision for InfoNCE logits (reference: lib/evaluation/metric.py:44-67 with label 0).
With the target in column 0, "label in top-k" <=> fewer than k negatives score >= the positive;
that count (`rank_ge`) comes out of the logits kernel, so no top-k sort and no host sync."""
import torch


def accuracy_from_rank(rank_ge, topk=(1, 5)):
    """rank_ge: (b,) int32 device tensor -> list of 1-element device tensors (percent)."""
    b = rank_ge.numel()
    return [(rank_ge < k).sum().reshape(1).float() * (100.0 / b) for k in topk]
