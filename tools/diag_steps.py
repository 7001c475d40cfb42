"""Diagnostic (not a test): per-key errors of the multi-step MoCo parity scenario."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import parity
from oracle import moco as omoco
pkg = importlib.import_module('video-graph-ssl_amd')
DEV = torch.device('cuda:0')
parity.register_tiny(pkg)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gen = torch.Generator().manual_seed(5)
imgs = [torch.randn(8, 6, 8, 48, 48, generator=gen) for _ in range(4)]
shs = [torch.randperm(8, generator=gen) for _ in range(4)]
cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 20, 8)
tr = pkg.MoCoTrainer(cfg, DEV, use_graph=False, seed=123)
state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
mem0 = tr.contrast.memory.detach().cpu().clone()
model, ema, contrast, opt = parity.oracle_moco('R2P1D10T', 32, 20, 8, state, mem0, omoco.warmup_multistep_factor(0))
crit = omoco.NCESoftmaxLoss()
for it in range(nsteps):
    out = tr.train_step(imgs[it].to(DEV), shuffle_ids=shs[it])
    ref = omoco.moco_train_step(model, ema, contrast, crit, opt, imgs[it], 0.999, shuffle_ids=shs[it])
    torch.cuda.synchronize()
    og = {n: p.grad for n, p in model.named_parameters()}
    ge = sorted(((parity.rel(p.grad, og[n]), n) for n, p in tr.model.named_parameters()), reverse=True)[:4]
    sk = tr.model_ema.state_dict()
    ke = sorted(((parity.rel(sk[k].float(), v.float()), k) for k, v in ema.state_dict().items() if v.dtype.is_floating_point), reverse=True)[:4]
    sq = tr.model.state_dict()
    qe = sorted(((parity.rel(sq[k].float(), v.float()), k) for k, v in model.state_dict().items() if v.dtype.is_floating_point), reverse=True)[:3]
    print('step', it, 'loss', parity.rel(out['loss'].reshape(()), ref['loss']), 'q', parity.rel(out['q'], ref['q']))
    print('  grads', ge)
    print('  key  ', ke)
    print('  query', qe)
