"""ContrastWrapper / SimSiam / GraphWrapper (reference: lib/modeling/graph_wrappers.py) on the engine.

``GraphWrapper.forward(x)`` keeps the reference behaviour (MoCo: (b,3,T,H,W) -> (b,FEAT_DIM) unit
rows; SimSiam: (b,6,T,H,W) -> scalar loss) and is differentiable through torch.autograd as ONE
node (engine/autograd_bridge.py); the trainer's fused step calls ``fwd`` directly."""
import torch
import torch.nn as nn

from .project_head import PredictionMLP, ProjectHead, ProjectionMLP
from ...engine import layers as L, ops
from ...engine.autograd_bridge import run_module
from ...engine.tape import Var


class ContrastWrapper(nn.Module):
    def __init__(self, encoder, hid_dim=128, head_type='mlp'):
        super().__init__()
        self.encoder = encoder
        self.proj_head = ProjectHead(encoder.feature_dim, hid_dim, head_type)

    def fwd(self, tape, xv):
        return self.proj_head.fwd(tape, self.encoder.fwd(tape, xv))

    def forward(self, x, bb_grad=True):
        out = run_module(self, x)
        return out if bb_grad else out.detach()


class D(nn.Module):
    """Negative cosine similarity with stop-gradient on the target (graph_wrappers.py:93-108)."""

    def __init__(self, fun_type='v2'):
        super().__init__()
        if fun_type not in ('v1', 'v2'):
            raise ValueError('Unknown type in simsiam D!')
        self.fun_type = fun_type


class SimSiam(nn.Module):
    def __init__(self, encoder, hid_dim=1024):
        super().__init__()
        self.encoder = encoder
        self.feature_dim = encoder.feature_dim
        self.projection = ProjectionMLP(self.feature_dim, hid_dim, hid_dim)
        self.prediction = PredictionMLP(hid_dim, hid_dim // 2, hid_dim)
        self.d = D()

    def fwd(self, tape, xv):
        """loss = D(p1, sg z2)/2 + D(p2, sg z1)/2 with both views through the encoder WITH grad (:48-71)."""
        x = xv.t
        x1, x2 = torch.chunk(x, 2, dim=1)          # views, read in place through the batch stride
        z1 = self.projection.fwd(tape, self.encoder.fwd(tape, Var(x1)))
        p1 = self.prediction.fwd(tape, z1)
        z2 = self.projection.fwd(tape, self.encoder.fwd(tape, Var(x2)))
        p2 = self.prediction.fwd(tape, z2)
        b = p1.t.shape[0]
        loss = torch.empty(1 + b, dtype=torch.float32, device=x.device)
        dp1 = ops.negcos(p1.t, z2.t, 0.5, loss, accumulate=False)
        dp2 = ops.negcos(p2.t, z1.t, 0.5, loss, accumulate=True)
        lv = Var(loss[:1], tape.recording)

        def back():
            # d loss / d p (already scaled by 1/2 and 1/b); upstream scalar gradient folds in here
            g = lv.grad
            if g is not None:
                gs = float(g.item()) if g.numel() == 1 and not torch.cuda.is_current_stream_capturing() else 1.0
                if gs != 1.0:
                    ops.scale_(dp1, gs)
                    ops.scale_(dp2, gs)
            st = ops.LOSS_SCALE_STATE[0]
            if st is not None:                       # fp16 storage: the loss gradient is seeded x the (device-resident) loss scale
                ops.scale_dev_(dp1, st)
                ops.scale_dev_(dp2, st)
            p1.add_grad(dp1)
            p2.add_grad(dp2)
        tape.record(back)
        return lv

    def forward(self, x):
        return run_module(self, x).reshape(())


class GraphWrapper(nn.Module):
    def __init__(self, encoder, hid_dim=1024, head_type='mlp', mem_type='simsiam'):
        super().__init__()
        if mem_type == 'simsiam':
            self.model = SimSiam(encoder=encoder, hid_dim=hid_dim)
        else:
            self.model = ContrastWrapper(encoder=encoder, hid_dim=hid_dim, head_type=head_type)

    def fwd(self, tape, xv):
        return self.model.fwd(tape, xv)

    def forward(self, x):
        return self.model(x)
