import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GCA_AUTOTUNE'] = '0'
pkg = importlib.import_module('video-graph-ssl_amd'); ops = pkg.engine.ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
N, C, D, H, W, K = 1, 1, 1, 4, 8, 2
x = torch.arange(N*C*D*H*W, dtype=torch.float32).view(N, C, D, H, W)
dy = torch.zeros(N, K, D, H, W); 
for o in range(32): dy[0, 0, 0, o // 8, o % 8] = 1.0 if o == int(sys.argv[1]) else 0.0
dy[0, 1] = 1.0
plan = ops.ConvPlan(N, C, D, H, W, K, 1, 1, 0, dev)
dw = torch.zeros(K, C, 1, 1, 1, device=dev)
ops.conv_wgrad(plan, x.to(dev), dy.to(dev), dw, accumulate=False)
print('dw', dw.flatten().tolist(), 'expected', [float(x.flatten()[int(sys.argv[1])]), float(x.sum())])
