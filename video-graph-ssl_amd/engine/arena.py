"""Flat parameter / gradient arenas.

All parameters of a model live in ONE contiguous fp32 buffer (each tensor padded to a multiple
of 256 elements = 1 KiB, so every per-tensor slice is float4-aligned and a (lr, weight-decay)
pair can be attached per 256-element chunk).  The EMA key-encoder update, SGD and the
gradient all-reduce then touch one buffer with one launch / one collective instead of one per
tensor (the reference does 1 launch per tensor: tools/train_video_contrast_dis.py:177-180,
lib/solver/build.py:24-59).
"""
import torch

CHUNK = 256


class ParamArena:
    def __init__(self, module):
        params = [(n, p) for n, p in module.named_parameters()]
        if not params:
            raise ValueError('module has no parameters')
        dev = params[0][1].device
        self.names, self.offsets, self.sizes = [], [], []
        off = 0
        for n, p in params:
            self.names.append(n)
            self.offsets.append(off)
            self.sizes.append(p.numel())
            off += (p.numel() + CHUNK - 1) // CHUNK * CHUNK
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.params = []
        for (n, p), o, s in zip(params, self.offsets, self.sizes):
            self.flat[o:o + s].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + s].view(p.shape)
            p.grad = self.grad[o:o + s].view(p.shape)
            self.params.append(p)
        module._gca_arena = self

    def chunk_table(self, per_param_values):
        """Expand one value per parameter into one value per 256-element chunk (host tensor)."""
        out = torch.zeros(self.total // CHUNK, dtype=torch.float32)
        for o, s, v in zip(self.offsets, self.sizes, per_param_values):
            out[o // CHUNK:(o + (s + CHUNK - 1) // CHUNK * CHUNK) // CHUNK] = float(v)
        return out

    def rebind(self):
        """Re-point .grad at the arena (optimisers / user code may have set p.grad = None)."""
        for p, o, s in zip(self.params, self.offsets, self.sizes):
            if p.grad is None or p.grad.data_ptr() != self.grad[o:o + s].data_ptr():
                p.grad = self.grad[o:o + s].view(p.shape)


def arena_of(module):
    a = getattr(module, '_gca_arena', None)
    if a is None:
        a = ParamArena(module)
    return a
