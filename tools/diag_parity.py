"""Diagnostic (not a test): per-parameter gradient errors HIP vs oracle for an encoder, in layer order."""
import importlib
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import parity
from oracle import encoders as oenc

pkg = importlib.import_module('video-graph-ssl_amd')
tp = importlib.import_module('video-graph-ssl_amd.engine.tape')
DEV = torch.device('cuda:0')
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,3,8,64,64').split(','))
which = sys.argv[2] if len(sys.argv) > 2 else 'r2t'
bb = pkg.lib.modeling.backbone.backbone_3d
torch.manual_seed(1)
if which == 'r2t':
    m = bb.resnet2p1d.generate_model(10, widen_factor=0.125); o = oenc.R2Plus1D(10, widen_factor=0.125)
elif which == 'r18':
    m = bb.R2P1D18(); o = oenc.R2Plus1D(18)
elif which == 's3d':
    m = bb.S3D(); o = oenc.S3D()
    m.fc = pkg.engine.layers.HipIdentity(); o.fc = torch.nn.Identity()
o.load_state_dict(m.state_dict())
m.to(DEV).train(); o.train()
x = torch.randn(shape)
xo = x.clone().requires_grad_(True)
yo = o(xo)
dy = torch.randn_like(yo)
yo.backward(dy)
tape = tp.Tape(True); xv = tp.Var(x.to(DEV), True)
out = m.fwd(tape, xv); out.grad = dy.to(DEV); tape.backward()
print('y   ', parity.rel(out.t, yo))
print('dx  ', parity.rel(xv.grad, xo.grad))
og = dict(o.named_parameters())
for n, p in m.named_parameters():
    print('%-40s %.3e   |g|=%.3e' % (n, parity.rel(p.grad, og[n].grad), float(og[n].grad.abs().max())))
ob = dict(o.named_buffers())
worst = max((parity.rel(b.float(), ob[n].float()), n) for n, b in m.named_buffers() if b.dtype.is_floating_point)
print('worst buffer', worst)
