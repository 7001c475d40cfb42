"""Diagnostic: gradient error of (a) the HIP path and (b) the fp32 CPU oracle, both against an fp64 CPU run."""
import importlib, sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import parity
from oracle import moco as omoco, wrappers as owrap
pkg = importlib.import_module('video-graph-ssl_amd')
DEV = torch.device('cuda:0')
parity.register_tiny(pkg)
gen = torch.Generator().manual_seed(5)
imgs = [torch.randn(8, 6, 8, 48, 48, generator=gen) for _ in range(4)]
shs = [torch.randperm(8, generator=gen) for _ in range(4)]
cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 20, 8)
tr = pkg.MoCoTrainer(cfg, DEV, use_graph=False, seed=123)
state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
mem0 = tr.contrast.memory.detach().cpu().clone()
f0 = omoco.warmup_multistep_factor(0)
m32, e32, c32, o32 = parity.oracle_moco('R2P1D10T', 32, 20, 8, state, mem0, f0)
m64, e64, c64, o64 = parity.oracle_moco('R2P1D10T', 32, 20, 8, state, mem0, f0)
m64.double(); e64.double(); c64.double()
o64 = omoco.make_optimizer(m64, 0.06, 0.9, 5e-4)
for g in o64.param_groups: g['lr'] *= f0
crit = omoco.NCESoftmaxLoss()
for it in range(4):
    # every step starts all three from the fp64 state so errors do not accumulate across steps
    sd64 = {k: v.clone() for k, v in m64.state_dict().items()}
    sk64 = {k: v.clone() for k, v in e64.state_dict().items()}
    m32.load_state_dict({k: v.float() if v.dtype.is_floating_point else v for k, v in sd64.items()})
    e32.load_state_dict({k: v.float() if v.dtype.is_floating_point else v for k, v in sk64.items()})
    c32.memory.copy_(c64.memory.float()); c32.index = c64.index
    tr.model.load_state_dict({k: v.float() if v.dtype.is_floating_point else v for k, v in sd64.items()})
    tr.model_ema.load_state_dict({k: v.float() if v.dtype.is_floating_point else v for k, v in sk64.items()})
    tr.contrast.memory.copy_(c64.memory.float())
    out = tr.train_step(imgs[it].to(DEV), shuffle_ids=shs[it])
    r32 = omoco.moco_train_step(m32, e32, c32, crit, o32, imgs[it], 0.999, shuffle_ids=shs[it])
    r64 = omoco.moco_train_step(m64, e64, c64, crit, o64, imgs[it].double(), 0.999, shuffle_ids=shs[it])
    torch.cuda.synchronize()
    g64 = {n: p.grad for n, p in m64.named_parameters()}
    g32 = {n: p.grad for n, p in m32.named_parameters()}
    rows = []
    for n, p in tr.model.named_parameters():
        rows.append((parity.rel(p.grad, g64[n]), parity.rel(g32[n], g64[n]), parity.rel(p.grad, g32[n]), n))
    rows.sort(reverse=True)
    print('step %d  q: hip-vs-64 %.2e  cpu32-vs-64 %.2e' % (it, parity.rel(out['q'], r64['q']), parity.rel(r32['q'], r64['q'])))
    for r in rows[:5]:
        print('   hip-vs-64 %.2e   cpu32-vs-64 %.2e   hip-vs-cpu32 %.2e   %s' % r)
