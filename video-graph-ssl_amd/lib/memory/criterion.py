"""InfoNCE criterion (reference: lib/memory/criterion.py:34-45) on the HIP kernels:
CrossEntropy(logits, label 0), mean over rows = mean_i( logsumexp(logits_i) - logits_i0 )."""
import torch
import torch.nn as nn

from ...engine import ops


class _NCELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        logits = logits.contiguous()
        loss, lse = ops.nce_loss_fwd(logits)
        ctx.logits, ctx.lse = logits, lse
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return ops.nce_loss_bwd(ctx.logits, ctx.lse, gscale_dev=g.contiguous().reshape(1))


class NCESoftmaxLoss(nn.Module):
    def forward(self, x):
        return _NCELossFn.apply(x)
