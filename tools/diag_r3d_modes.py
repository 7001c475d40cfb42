#!/usr/bin/env python3
"""Full-width 3D-ResNet-18 forward+backward vs the fp64 oracle under each conv arithmetic mode, next to the fp32 CPU
oracle's own distance from fp64 (how much of the error is fp32 itself)."""
import importlib, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
os.environ.setdefault('GCA_AUTOTUNE', '0')
pkg = importlib.import_module('video-graph-ssl_amd')
import parity
from oracle import encoders as oenc
tp = importlib.import_module('video-graph-ssl_amd.engine.tape')
DEV = torch.device('cuda:0')
bb = pkg.lib.modeling.backbone.backbone_3d
torch.manual_seed(17)
m = bb.resnet.resnet18(sample_size=64, sample_duration=16)
x = torch.randn(4, 3, 16, 64, 64)
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}


def reference(double):
    ref = oenc.R3D(18, sample_size=64, sample_duration=16)
    ref.load_state_dict(sd)
    ref = (ref.double() if double else ref).train()
    xr = (x.double() if double else x.clone()).requires_grad_(True)
    yr = ref(xr)
    return ref, xr, yr.reshape(yr.shape[0], -1)


r64, x64, y64 = reference(True)
dy = torch.randn(y64.shape, dtype=torch.float64)
y64.backward(dy)
g64 = {n: q.grad for n, q in r64.named_parameters()}
r32, x32, y32 = reference(False)
y32.backward(dy.float())
e32 = sorted(parity.rel(q.grad, g64[n]) for n, q in r32.named_parameters())
print('cpu-fp32 oracle: y %.2e  grad median %.2e p95 %.2e max %.2e  dx %.2e' %
      (parity.rel(y32, y64), e32[len(e32) // 2], e32[int(.95 * len(e32))], e32[-1], parity.rel(x32.grad, x64.grad)))
m.to(DEV).train()
for mode in ('f32', 'bf16x6', 'bf16x3'):
    pkg.engine.ops.set_conv_math(mode)
    for q in m.parameters():
        q.grad = None
    tape = tp.Tape(True); xv = tp.Var(x.to(DEV), True)
    out = m.fwd(tape, xv); out.grad = dy.float().to(DEV).reshape(-1, y64.shape[1]); tape.backward()
    errs = {n: parity.rel(q.grad, g64[n]) for n, q in m.named_parameters() if q.grad is not None}
    e = sorted(errs.values())
    worst = max(errs, key=errs.get)
    print('%-7s y %.2e  grad median %.2e p95 %.2e max %.2e (%s)  dx %.2e' %
          (mode, parity.rel(out.t.reshape(y64.shape), y64), e[len(e) // 2], e[int(.95 * len(e))], e[-1], worst, parity.rel(xv.grad, x64.grad)))
