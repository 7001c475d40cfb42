"""Diagnostic (not collected): S3D forward/backward per-parameter gradient error, HIP vs fp64 oracle vs fp32 oracle."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import parity
from oracle import encoders as oenc
pkg = importlib.import_module('video-graph-ssl_amd')
tp = pkg.engine.tape
DEV = torch.device('cuda:0')
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bb = pkg.lib.modeling.backbone.backbone_3d
torch.manual_seed(17)
m = bb.S3D(); m.fc = pkg.engine.layers.HipIdentity()
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
def mk(dbl):
    r = oenc.S3D(); r.fc = torch.nn.Identity(); r.load_state_dict(sd)
    return (r.double() if dbl else r).train()
r64, r32 = mk(True), mk(False)
m.to(DEV).train()
x = torch.randn(4, 3, 16, size, size)
t0 = time.time()
xr = x.double().requires_grad_(True); y64 = r64(xr); y64 = y64.reshape(4, -1)
dy = torch.randn(y64.shape, dtype=torch.float64); y64.backward(dy)
x32 = x.clone().requires_grad_(True); y32 = r32(x32).reshape(4, -1); y32.backward(dy.float())
print('oracle time %.1fs' % (time.time() - t0))
tape = tp.Tape(True); xv = tp.Var(x.to(DEV), True)
out = m.fwd(tape, xv); out.grad = dy.float().to(DEV).reshape(out.t.shape); tape.backward()
print('fwd err hip %.2e cpu32 %.2e' % (parity.rel(out.t.reshape(4, -1), y64), parity.rel(y32, y64)))
g64 = {n: q.grad for n, q in r64.named_parameters()}; g32 = {n: q.grad for n, q in r32.named_parameters()}
rows = [(parity.rel(q.grad, g64[n]), parity.rel(g32[n], g64[n]), n) for n, q in m.named_parameters() if q.grad is not None and float(g64[n].abs().max()) > 1e-12]
rows.sort(reverse=True)
med = lambda i: sorted(r[i] for r in rows)[len(rows) // 2]
print('params %d  median hip %.2e cpu32 %.2e   dx hip %.2e cpu32 %.2e' % (len(rows), med(0), med(1), parity.rel(xv.grad, xr.grad), parity.rel(x32.grad, xr.grad)))
for r in rows[:6]: print('  hip %.2e  cpu32 %.2e  %s' % r)
