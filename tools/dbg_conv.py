import importlib, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GCA_AUTOTUNE'] = '0'
pkg = importlib.import_module('video-graph-ssl_amd'); ops = pkg.engine.ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
def rel(a, b): return float((a.cpu() - b).abs().max() / b.abs().max())
for shape, K, k, s, p in [((2, 4, 2, 6, 6), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)), ((2, 4, 3, 6, 6), 8, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
                          ((2, 4, 2, 6, 6), 8, (1, 3, 3), (1, 1, 1), (0, 0, 0))]:
    x = torch.randn(shape); w = torch.randn((K, shape[1]) + k)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p); dy = torch.randn_like(yr); yr.backward(dy)
    plan = ops.conv_plan(shape, K, k, s, p, dev)
    y, (ss, sq) = ops.conv_fwd(plan, x.to(dev), ops.conv_pack(plan, 0, w.to(dev)), None, stats=True)
    dx = ops.conv_dgrad(plan, dy.to(dev), ops.conv_pack(plan, 1, w.to(dev)))
    dw = torch.zeros_like(w).to(dev); ops.conv_wgrad(plan, x.to(dev), dy.to(dev), dw, accumulate=True)
    print(k, p, 'cfg', plan.cfg(0), 'y', rel(y, yr.detach()), 'dx', rel(dx, xr.grad), 'dw', rel(dw, wr.grad),
          'sum', rel(ss.sum(1), yr.detach().sum((0, 2, 3, 4))), 'sq', rel(sq.sum(1), (yr.detach() ** 2).sum((0, 2, 3, 4))))
    if rel(y, yr.detach()) > 1e-4:
        torch.set_printoptions(precision=3, linewidth=200)
        print(y[0, 0, 0].cpu()); print(yr[0, 0, 0].detach())
print('--- split-K')
shape, K, k, s, p = (2, 20, 3, 12, 12), 100, (1, 3, 3), (1, 2, 2), (0, 1, 1)
x = torch.randn(shape); w = torch.randn((K, shape[1]) + k)
yr = F.conv3d(x, w, None, s, p)
plan = ops.ConvPlan(*shape, K, k, s, p, dev); plan.tuned = [True, True, True]
for code, sp in ((32, 1), (32, 3), (64, 2)):
    plan.g.tune_fwd_bm = code; plan.g.tune_fwd_splits = sp; plan.refresh()
    y, (ss, sq) = ops.conv_fwd(plan, x.to(dev), ops.conv_pack(plan, 0, w.to(dev)), None, stats=True)
    print(code, sp, plan.cfg(0), 'y', rel(y, yr), 'sum', rel(ss.sum(1), yr.sum((0, 2, 3, 4))))
    if rel(y, yr) > 1e-4:
        d = (y.cpu() - yr).abs()
        print('bad frac', float((d > 1e-3).float().mean()), 'per-channel bad', (d > 1e-3).float().mean((0, 2, 3, 4))[:40])
        print('ratio', (y.cpu() / yr)[0, :6, 0, 0, :4])
