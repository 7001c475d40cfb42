"""Minimal reverse-mode tape for the HIP engine.

The encoder is a static DAG of a few dozen coarse fused ops (conv+BN+ReLU(+residual), pools,
linears), so instead of one autograd node per ATen call the engine records one closure per
fused op and replays them in reverse.  torch.autograd only ever sees ONE node per model
(engine/autograd_bridge.py), which keeps the reference's ``loss.backward()`` API working.
"""
import torch

from . import ops


class Var:
    """A tensor travelling through the engine plus its (lazily created) gradient buffer."""
    __slots__ = ('t', 'grad', 'needs_grad')

    def __init__(self, t, needs_grad=False):
        self.t = t
        self.grad = None
        self.needs_grad = needs_grad

    def grad_buffer(self):
        """-> (buffer, accumulate): kernels that can add in place write straight into it."""
        if self.grad is None:
            self.grad = torch.empty_like(self.t)
            return self.grad, False
        return self.grad, True

    def add_grad(self, g):
        if self.grad is None:
            self.grad = g
        else:
            ops.axpy(self.grad, g, 1.0)


class LazyVar(Var):
    """z = relu(y * scale[c] + shift[c]) -- the output of a conv -> BatchNorm -> ReLU unit -- NOT yet written.  A consumer that
    can apply the BatchNorm + ReLU where it reads its input (engine.layers: convs on the LDS-halo / streaming kernels) takes
    (y, scale, shift) and z never exists; any other consumer reads `.t`, which materialises z once (gca_bn_apply: what the
    producer would have launched anyway)."""
    __slots__ = ('y', 'scale', 'shift', '_z')

    def __init__(self, y, scale, shift, needs_grad=False):
        self.y, self.scale, self.shift, self._z = y, scale, shift, None
        self.grad = None
        self.needs_grad = needs_grad

    @property
    def t(self):
        if self._z is None:
            N, Cc = self.y.shape[0], self.y.shape[1]
            self._z = ops.bn_apply(self.y, self.scale, self.shift, None, True, N, Cc, self.y[0, 0].numel())
        return self._z

    @property
    def materialized(self):
        return self._z is not None

    def grad_buffer(self):
        if self.grad is None:
            self.grad = torch.empty_like(self.y)
            return self.grad, False
        return self.grad, True


CURRENT = [-1]        # index of the closure being run by Tape.backward (read by the gradient log of engine/layers.py)


class Tape:
    def __init__(self, recording):
        self.recording = recording
        self.fns = []

    def record(self, fn):
        if self.recording:
            self.fns.append(fn)

    def backward(self, upto=0):
        """Run the recorded closures in reverse, down to (and including) index `upto`.  Calling it again with a smaller
        index continues where the previous call stopped: the multi-GPU trainer runs the backward pass in stages and
        starts the all-reduce of a gradient bucket as soon as the last closure that writes into it has been issued."""
        while len(self.fns) > upto:
            CURRENT[0] = len(self.fns) - 1
            self.fns.pop()()
        CURRENT[0] = -1
        if ops.DEFER[0] is not None:
            ops.DEFER[0].flush()          # the weight-gradient reductions of this stage: one launch (ops.DeferredReduce)
