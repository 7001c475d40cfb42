import importlib, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops
x = torch.randn(32, 64, 16, 56, 56, device='cuda')
sc, sh = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda')
plan = ops.pool_plan(tuple(x.shape), (3, 3, 3), (2, 2, 2), (1, 1, 1))
y, am = ops.maxpool_fwd(plan, x, True, sc, sh)
dy = torch.randn_like(y)
dx = torch.empty_like(x)
print('tiled' if os.environ.get('GCA_POOL_TILED', '1') != '0' else 'untiled',
      'fwd %.3f ms' % bench.ev_time_ms(lambda: ops.maxpool_fwd(plan, x, True, sc, sh), 10, 2),
      'bwd %.3f ms' % bench.ev_time_ms(lambda: ops.maxpool_bwd(plan, dy, am, dx, False), 10, 2),
      float(y.sum()), int(am.sum() % 1000003), float(ops.maxpool_bwd(plan, dy, am).sum()))
