"""Diagnostic (not collected): per-parameter gradient error of teacher-forced SimSiam steps, HIP vs fp64 oracle."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import parity
from oracle import moco as omoco, wrappers as owrap
pkg = importlib.import_module('video-graph-ssl_amd')
DEV = torch.device('cuda:0')
parity.register_tiny(pkg)
cfg = parity.make_cfg(pkg, 'R2P1D10T', 'simsiam', 32, 16, 8)
tr = pkg.SimSiamTrainer(cfg, DEV, use_graph=False, seed=11)
state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
ref, _ = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'simsiam')
ref.load_state_dict(state); ref.double().train()
r32, _ = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'simsiam'); r32.train()
opt = omoco.make_optimizer(ref, 0.06, 0.9, 5e-4)
for gr in opt.param_groups: gr['lr'] *= 0.01
gen = torch.Generator().manual_seed(21)
for step in range(3):
    x = torch.randn(8, 6, 8, 48, 48, generator=gen)
    sd32 = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in ref.state_dict().items()}
    tr.model.load_state_dict(sd32); r32.load_state_dict(sd32)
    out = tr.train_step(x.to(DEV))
    l32 = r32(x); r32.zero_grad(); l32 = r32(x); l32.backward()
    want = omoco.simsiam_train_step(ref, opt, x.double())
    g64 = {n: q.grad for n, q in ref.named_parameters()}; g32 = {n: q.grad for n, q in r32.named_parameters()}
    rows = []
    for n, q in tr.model.named_parameters():
        s = float(g64[n].abs().max())
        if s > 1e-12:
            rows.append((parity.rel(q.grad, g64[n]), parity.rel(g32[n], g64[n]), n, s))
    rows.sort(reverse=True)
    print('step', step, 'loss', float(out['loss']), float(want['loss']), float(l32), 'median hip %.2e cpu32 %.2e' % (sorted(r[0] for r in rows)[len(rows)//2], sorted(r[1] for r in rows)[len(rows)//2]))
    for r in rows[:4]: print('   %.3e  cpu32 %.3e  %-50s scale %.3e' % r)
print('--- replay step-2 data on a fresh trainer, and step-2 data twice on the old one')
gen = torch.Generator().manual_seed(21)
xs = [torch.randn(8, 6, 8, 48, 48, generator=gen) for _ in range(3)]
ref2, _ = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'simsiam'); ref2.load_state_dict(state); ref2.double().train()
opt2 = omoco.make_optimizer(ref2, 0.06, 0.9, 5e-4)
for gr in opt2.param_groups: gr['lr'] *= 0.01
for i in range(2): omoco.simsiam_train_step(ref2, opt2, xs[i].double())
sd2 = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in ref2.state_dict().items()}
omoco.simsiam_train_step(ref2, opt2, xs[2].double())
g64 = {n: q.grad for n, q in ref2.named_parameters()}
def report(tag, t):
    errs = sorted(parity.rel(q.grad, g64[n]) for n, q in t.model.named_parameters() if float(g64[n].abs().max()) > 1e-12)
    print(tag, 'median %.2e worst %.2e' % (errs[len(errs)//2], errs[-1]))
t2 = pkg.SimSiamTrainer(cfg, DEV, use_graph=False, seed=11)
t2.model.load_state_dict(sd2); t2.train_step(xs[2].to(DEV)); report('fresh trainer, 1st call   ', t2)
t2.model.load_state_dict(sd2); t2.train_step(xs[2].to(DEV)); report('fresh trainer, 2nd call   ', t2)
t2.model.load_state_dict(sd2); t2.train_step(xs[2].to(DEV)); report('fresh trainer, 3rd call   ', t2)
t2._packer = None
t2.model.load_state_dict(sd2); t2.train_step(xs[2].to(DEV)); report('same, packer dropped      ', t2)
print('--- per-parameter errors at the top of the network (fresh trainer, step-2 data)')
for n, q in t2.model.named_parameters():
    if float(g64[n].abs().max()) > 1e-12 and ('projection' in n or 'prediction' in n or 'layer4.1' in n or 'layer4.0' in n):
        print('   %.3e  %-55s scale %.3e' % (parity.rel(q.grad, g64[n]), n, float(g64[n].abs().max())))
print('--- gradient wrt the encoder features (both views)')
from importlib import import_module
gw = import_module('video-graph-ssl_amd.lib.modeling.graph_wrappers')
Var = gw.Var
ops = pkg.engine.ops
saved = {}
def fwd_dbg(self, tape, xv):
    x = xv.t
    x1, x2 = torch.chunk(x, 2, dim=1)
    f1 = self.encoder.fwd(tape, Var(x1)); tape.record(lambda: saved.__setitem__('df1', f1.grad.clone()))
    saved['f1'] = f1.t.clone()
    z1 = self.projection.fwd(tape, f1); tape.record(lambda: saved.__setitem__('dz1', z1.grad.clone()))
    p1 = self.prediction.fwd(tape, z1)
    f2 = self.encoder.fwd(tape, Var(x2)); tape.record(lambda: saved.__setitem__('df2', f2.grad.clone()))
    z2 = self.projection.fwd(tape, f2); tape.record(lambda: saved.__setitem__('dz2', z2.grad.clone()))
    p2 = self.prediction.fwd(tape, z2)
    b = p1.t.shape[0]
    loss = torch.empty(1 + b, dtype=torch.float32, device=x.device)
    dp1 = ops.negcos(p1.t, z2.t, 0.5, loss, accumulate=False)
    dp2 = ops.negcos(p2.t, z1.t, 0.5, loss, accumulate=True)
    saved['dp1'], saved['dp2'] = dp1.clone(), dp2.clone()
    lv = Var(loss[:1], tape.recording)
    def back():
        p1.add_grad(dp1); p2.add_grad(dp2)
    tape.record(back)
    return lv
gw.SimSiam.fwd = fwd_dbg
ref3, _ = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'simsiam'); ref3.load_state_dict(sd2); ref3.double().train()
oh = {}
cnt = [0]
def enc_hook(m, i, o):
    k = cnt[0]; cnt[0] += 1
    o.register_hook(lambda g: oh.__setitem__('df%d' % (k + 1), g.clone()))
def proj_hook(m, i, o):
    k = cnt[0]
    o.register_hook(lambda g: oh.__setitem__('dz%d' % k, g.clone()))
ref3.model.encoder.register_forward_hook(enc_hook)
ref3.model.projection.register_forward_hook(proj_hook)
l = ref3(xs[2].double()); l.backward()
t3 = pkg.SimSiamTrainer(cfg, DEV, use_graph=False, seed=11)
t3.model.load_state_dict(sd2); t3.train_step(xs[2].to(DEV)); torch.cuda.synchronize()
for k in ('dz1', 'dz2', 'df1', 'df2'):
    print(k, 'err %.3e' % parity.rel(saved[k], oh[k]), 'scale %.3e' % float(oh[k].abs().max()))
print('--- smallest |pre-activation| at the last block output (fp64 oracle), both views')
blk = ref3.model.encoder.base_model.layer4[0]
pre = []
bo, do = [], []
h1 = blk.bn2_t.register_forward_hook(lambda m, i, o: bo.append(o.detach().clone()))
h2 = blk.downsample.register_forward_hook(lambda m, i, o: do.append(o.detach().clone()))
ref3(xs[2].double())
for v in range(2):
    p_ = (bo[v] + do[v]).flatten()
    idx = p_.abs().argsort()[:4]
    print('view', v, 'smallest |pre|:', ['%.3e' % float(p_[i]) for i in idx], 'typical |pre| %.3e' % float(p_.abs().median()))
