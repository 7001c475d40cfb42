"""One torch.autograd node per engine model.

The reference drives training with ``loss.backward()`` (tools/train_video_contrast_dis.py:419).
To keep that API without paying one autograd node per kernel, a whole engine forward (dozens of
fused HIP ops recorded on a Tape) is exposed to autograd as a single Function: its backward
replays the tape, which accumulates parameter gradients straight into ``param.grad``."""
from types import SimpleNamespace

import torch

from .tape import Tape, Var


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, holder):
        ctx.holder = holder
        return holder.outv.t

    @staticmethod
    def backward(ctx, g):
        h = ctx.holder
        if h.done:
            raise RuntimeError('engine tape was already consumed (backward twice through the same forward)')
        h.done = True
        h.outv.grad = g.contiguous().view_as(h.outv.t)
        h.tape.backward()
        return None, None


def run_module(mod, x):
    """Forward `mod.fwd` on the engine; differentiable (as one node) when grad mode is on."""
    params = [p for p in mod.parameters() if p.requires_grad]
    need = torch.is_grad_enabled() and len(params) > 0
    tape = Tape(recording=need)
    outv = mod.fwd(tape, Var(x, False))
    if not need:
        return outv.t
    holder = SimpleNamespace(tape=tape, outv=outv, done=False)
    return _EngineFn.apply(params[0], holder)
