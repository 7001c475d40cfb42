"""Diagnostic: per-BN-layer gradient (wrt BN output and wrt conv output) HIP vs fp64 oracle at a given step."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import parity
from oracle import moco as omoco
pkg = importlib.import_module('video-graph-ssl_amd')
L = pkg.engine.layers
DEV = torch.device('cuda:0')
parity.register_tiny(pkg)
target = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gen = torch.Generator().manual_seed(5)
imgs = [torch.randn(8, 6, 8, 48, 48, generator=gen) for _ in range(4)]
shs = [torch.randperm(8, generator=gen) for _ in range(4)]
cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 20, 8)
tr = pkg.MoCoTrainer(cfg, DEV, use_graph=False, seed=123)
state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
mem0 = tr.contrast.memory.detach().cpu().clone()
f0 = omoco.warmup_multistep_factor(0)
m64, e64, c64, _ = parity.oracle_moco('R2P1D10T', 32, 20, 8, state, mem0, f0)
m64.double(); e64.double(); c64.double()
o64 = omoco.make_optimizer(m64, 0.06, 0.9, 5e-4)
for g in o64.param_groups: g['lr'] *= f0
crit = omoco.NCESoftmaxLoss()
hooks = {}
def mk(name):
    def h(mod, gin, gout):
        hooks[name] = (gout[0].detach().clone(), gin[0].detach().clone(), mod.weight.detach().clone(), mod.running_var.clone())
    return h
for n, mod in m64.named_modules():
    if isinstance(mod, torch.nn.BatchNorm3d):
        mod.register_full_backward_hook(mk(n))
names = {id(mod): n for n, mod in tr.model.named_modules()}
for it in range(target + 1):
    f32 = lambda sd: {k: v.float() if v.dtype.is_floating_point else v for k, v in sd.items()}
    tr.model.load_state_dict(f32(m64.state_dict())); tr.model_ema.load_state_dict(f32(e64.state_dict()))
    tr.contrast.memory.copy_(c64.memory.float())
    L.DEBUG_GRADS = {} if it == target else None
    hooks.clear()
    out = tr.train_step(imgs[it].to(DEV), shuffle_ids=shs[it])
    r64 = omoco.moco_train_step(m64, e64, c64, crit, o64, imgs[it].double(), 0.999, shuffle_ids=shs[it])
torch.cuda.synchronize()
print('step', target, 'q err', parity.rel(out['q'], r64['q']))
for bid, (dz, dy) in L.DEBUG_GRADS.items():
    n = names[bid]
    gout, gin, w, rv = hooks[n]
    # oracle grad_output of BN is pre-ReLU; ours (dz) is wrt the post-ReLU z: compare dy (wrt conv output) only,
    print('%-45s dy err %.2e   |dy| %.2e  shape %s' % (n, parity.rel(dy, gin), float(gin.abs().max()), tuple(dy.shape)))
