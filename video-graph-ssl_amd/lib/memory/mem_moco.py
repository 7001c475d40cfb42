"""MoCo queue (reference: lib/memory/mem_moco.py) on the HIP kernels.

``RGBMoCo.forward(q, k, q_jig=None, all_k=None) -> (logits (b,K+1), labels (b,))`` with the
reference's side effects: logits use the PRE-enqueue queue (:72), then ``all_k`` (or ``k``) is written
at rows (index + i) % K (:17-27) and ``index`` advances (:14-15).  Device-agnostic construction; the
kernels run wherever the buffers live (GPU required).  Differences by design:
  * no K x D ``clone()`` per step: only the n rows about to be overwritten are saved, and the
    backward kernel reads those rows from the saved copy (same gradient as against the snapshot);
  * `index` stays a plain Python attribute and `state_dict()` holds `memory` only, exactly as in the reference
    (so checkpoints load in both directions); the reference therefore loses the pointer on resume (SURVEY.md
    section 5).  `MoCoTrainer.state_dict()` stores it next to the reference's keys as `queue_index`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...engine import ops


class _MoCoLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, mem):
        logits, _, _ = ops.moco_logits_fwd(q.contiguous(), k, mem.memory, 1.0 / mem.T)
        ctx.mem, ctx.k, ctx.ticket = mem, k, mem._ticket
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        mem = ctx.mem
        ov = mem._overwritten.get(ctx.ticket)
        if ov is None or mem._ticket != ctx.ticket + 1:
            raise RuntimeError('RGBMoCo: backward must run before the next forward (queue snapshot semantics)')
        start, rows = ov
        dq = ops.moco_logits_bwd(ctx.k, mem.memory, 1.0 / mem.T, dlogits=dlogits.contiguous(),
                                 ov_start=start, ov_rows=rows)
        return dq, None, None


class BaseMoCo(nn.Module):
    def __init__(self, K=65536, T=0.07):
        super().__init__()
        self.K, self.T, self.index = K, T, 0
        self._ticket = 0
        self._overwritten = {}

    def _update_pointer(self, bsz):
        self.index = (self.index + bsz) % self.K

    def _update_memory(self, k, queue):
        if k.shape[0] > self.K:
            raise ValueError('cannot enqueue %d keys into a queue of %d' % (k.shape[0], self.K))
        saved = ops.queue_enqueue(queue, k.contiguous(), self.index, save=True)
        self._overwritten = {self._ticket: (self.index, saved)}
        self._ticket += 1


class RGBMoCo(BaseMoCo):
    def __init__(self, n_dim, K=65536, T=0.07):
        super().__init__(K, T)
        self.register_buffer('memory', F.normalize(torch.randn(K, n_dim)))     # init-time only (:57-58)

    def forward(self, q, k, q_jig=None, all_k=None):
        k = k.detach().contiguous()
        logits = _MoCoLogitsFn.apply(q, k, self)
        logits_jig = _MoCoLogitsFn.apply(q_jig, k, self) if q_jig is not None else None
        labels = torch.zeros(q.size(0), dtype=torch.long, device=q.device)
        all_k = k if all_k is None else all_k.detach()
        self._update_memory(all_k, self.memory)
        self._update_pointer(all_k.size(0))
        if q_jig is not None:
            return logits, logits_jig, labels
        return logits, labels
