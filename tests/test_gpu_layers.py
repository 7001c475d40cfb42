"""Well-conditioned backward checks, one layer at a time, at the sizes and launch shapes bench.py runs.

Model-level gradient comparisons of a freshly initialised network are chaotic (ReLU / max-pool decisions flip under
one rounding: DESIGN.md section 2), so their bars are distributions.  A single layer's backward is NOT chaotic: for
fixed (x, dy) the input gradient and the weight gradient are linear maps.  These tests therefore take (x, dy) of
every convolution from ONE iteration of the CPU oracle at the full BASELINE size and hold each layer's forward,
dgrad and wgrad to a tight max-norm bar -- a wrong tile / split / box / class for any one geometry shows up as that
layer's number, whatever the model-level statistic does.

  configs[1]  R(2+1)D-18, (32, 3, 16, 112, 112): all 37 encoder convolutions + the 2 head Linears, committed tuned
              launch shapes (profiles/tune_cache.json, autotuner on), bf16x6 and f32.
  configs[3]  the three TemporalGraphAug sites of S3D at 224x224 crops, 4 clips: (192, 8, 28, 28), (512, 4, 14, 14),
              (832, 2, 7, 7): the nine 1x1x1 convolutions, the Gram / adjacency kernels and the message passing,
              forward and backward, against the oracle block in fp64.
  configs[0]  the YAML's own workload: b = 2, K = 256, S3D at 16 frames and R(2+1)D-18 at 8 frames, 112 x 112 --
              full MoCo iterations from configs/visual_moco.yaml's keys against oracle.moco.moco_train_step.
"""
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import ROOT, rel_err
import parity

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture
def tuned_launch_shapes(pkg):
    """bench.py's configuration: committed tune cache + autotuner on (the rest of the suite pins the heuristic shapes)."""
    ops = pkg.engine.ops
    saved_auto, saved_cache, saved_math = ops.AUTOTUNE, dict(ops._TUNE_CACHE), ops.get_conv_math()
    ops.AUTOTUNE = True
    ops.load_tune_cache(os.path.join(ROOT, 'profiles', 'tune_cache.json'))
    ops._conv_plan.cache_clear()
    yield ops
    ops.AUTOTUNE = saved_auto
    ops._TUNE_CACHE.clear()
    ops._TUNE_CACHE.update(saved_cache)
    ops._TUNE_DIRTY[0] = False
    ops.set_conv_math(saved_math)
    ops._conv_plan.cache_clear()


# ------------------------------------------------------------------------------------------------ configs[1], per layer
_C1_CACHE = {}


def _configs1_layer_records():
    """One query-branch iteration of the fp32 CPU oracle on configs[1] (R(2+1)D-18 + head, 32 clips x 16 x 112 x 112, InfoNCE
    against a K = 4096 queue), with every Conv3d / Linear's input, output gradient, input gradient and weight gradient
    recorded by hooks.  ~10 s on the GPU box's host cores; ~6 GB of host memory; computed once per session."""
    if 'recs' in _C1_CACHE:
        return _C1_CACHE['recs']
    from oracle import moco as omoco, wrappers as owrap
    torch.manual_seed(1)
    model, _ = owrap.create_visual_model('R2P1D18', 16, 128, 'mlp', 'moco')
    model.train()
    contrast = omoco.RGBMoCo(128, K=4096, T=0.07)
    gen = torch.Generator().manual_seed(1)
    x1 = torch.randn(32, 3, 16, 112, 112, generator=gen)
    k = F.normalize(torch.randn(32, 128, generator=gen))
    recs, order = {}, []

    def fwd_hook(name):
        def hook(m, inp, out):
            recs[name] = dict(m=m, x=inp[0].detach(), y=out.detach().clone(), needs_dx=inp[0].requires_grad)
            order.append(name)
            # a tensor hook registered BEFORE any in-place consumer (the head's ReLU(inplace=True)) receives the gradient with
            # respect to the layer's own output (module backward hooks refuse in-place consumers)
            out.register_hook(lambda g, name=name: recs[name].__setitem__('dy', g.detach().clone()))
        return hook
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv3d, nn.Linear)):
            m.register_forward_hook(fwd_hook(name))
    q = model(x1)
    logits, _ = contrast(q, k)
    omoco.NCESoftmaxLoss()(logits).backward()
    out = []
    for name in order:
        r = recs[name]
        m = r.pop('m')
        r.update(name=name, w=m.weight.detach(), dw=m.weight.grad.detach(), bias=None if m.bias is None else m.bias.detach(),
                 db=None if m.bias is None else m.bias.grad.detach())
        if isinstance(m, nn.Conv3d):
            r.update(k=tuple(m.kernel_size), s=tuple(m.stride), p=tuple(m.padding))
            # the layer's OWN input gradient (a tensor hook on x would see the sum over all of x's consumers): ATen's
            # backward-data convolution of the recorded dy, the kernel autograd ran
            r['dx'] = torch.nn.grad.conv3d_input(r['x'].shape, r['w'], r['dy'], m.stride, m.padding) if r['needs_dx'] else None
        else:
            r.update(k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0))
            r['dx'] = r['dy'] @ r['w'] if r['needs_dx'] else None
        out.append(r)
    _C1_CACHE['recs'] = out
    return out


@pytest.mark.parametrize('math', ['bf16x6', 'f32'])
def test_configs1_every_layer_fwd_dgrad_wgrad_under_tuned_shapes(pkg, tuned_launch_shapes, math):
    """All 39 GEMM layers of configs[1], each fed the oracle's own (x, dy): forward, input gradient and weight gradient
    against the oracle's (fp32 oneDNN) results.  Bar 5e-5 max-norm per layer and pass -- two fp32-grade implementations
    with different summation orders over up to 1.6 M products agree to ~1e-6; a wrong tile/box/class is O(1).  The bias
    gradient of the head Linears rides along."""
    ops = tuned_launch_shapes
    ops.set_conv_math(math)
    recs = _configs1_layer_records()
    assert len(recs) == 39
    worst, tuned = {}, 0
    for r in recs:
        x, dy, w = r['x'].to(DEV), r['dy'].to(DEV), r['w'].to(DEV)
        lin = x.dim() == 2
        if lin:
            x, dy, w = x.reshape(*x.shape, 1, 1, 1), dy.reshape(*dy.shape, 1, 1, 1), w.reshape(*w.shape, 1, 1, 1)
        plan = ops.conv_plan(tuple(x.shape), w.shape[0], r['k'], r['s'], r['p'], DEV)
        bias = None if r['bias'] is None else r['bias'].to(DEV)
        wp0 = ops.conv_pack(plan, 0, w)
        if bias is None:
            y, (ss, sq) = ops.conv_fwd(plan, x, wp0, None, stats=True, w_raw=w)
        else:
            y = ops.conv_fwd(plan, x, wp0, bias, w_raw=w)
        e = dict(fwd=rel_err(y.reshape(r['y'].shape), r['y']))
        if bias is None:                                         # BatchNorm statistics from the same epilogue
            yr = r['y'].double()
            e['stat_sum'] = float((ss.sum(1).double().cpu() - yr.sum((0, 2, 3, 4))).abs().max() / yr.abs().sum((0, 2, 3, 4)).max())
            e['stat_sq'] = rel_err(sq.sum(1), (yr * yr).sum((0, 2, 3, 4)))
        if r['dx'] is not None:
            wp1 = ops.conv_pack(plan, 1, w)
            dx = ops.conv_dgrad(plan, dy, wp1, w_raw=w)
            e['dgrad'] = rel_err(dx.reshape(r['dx'].shape), r['dx'])
        dw = torch.zeros_like(w)
        ops.conv_wgrad(plan, x, dy, dw, accumulate=True)
        e['wgrad'] = rel_err(dw.reshape(r['dw'].shape), r['dw'])
        ops.conv_wgrad(plan, x, dy, dw, accumulate=True)         # += into a live gradient buffer
        e['wgrad_acc'] = rel_err(dw.reshape(r['dw'].shape), 2 * r['dw'])
        if r['db'] is not None:
            db = torch.zeros(w.shape[0], device=DEV)
            ops.bias_grad(dy, x.shape[0], w.shape[0], 1, db, True)
            e['bias_grad'] = rel_err(db, r['db'])
        g = plan.g
        tuned += int(any((g.tune_fwd_bm, g.tune_dgrad_bm, g.tune_wgrad_tile)))
        for key, v in e.items():
            assert v < 5e-5, (math, r['name'], tuple(x.shape), r['k'], r['s'], key, v, plan.cfg(0), plan.cfg(1), plan.cfg(2))
            if v > worst.get(key, (0.0, ''))[0]:
                worst[key] = (v, r['name'])
        del x, dy, w, y, dw
    assert tuned >= 30, tuned                                    # the plans really carry measured launch shapes
    print('configs[1] %s per-layer worst: %s' % (math, {k_: '%.1e @ %s' % v for k_, v in worst.items()}))


# ------------------------------------------------------------------------------------------------ configs[3], graph sites
@pytest.mark.parametrize('C,T,HW', [(192, 8, 28), (512, 4, 14), (832, 2, 7)])
@pytest.mark.parametrize('math,xscale', [('bf16x6', 0.1), ('f32', 0.1), ('bf16x6', 1.0)])
def test_configs3_graph_block_at_224_sites_vs_fp64_oracle(pkg, math, xscale, C, T, HW):
    """TemporalGraphAug as configs[3] runs it (S3D at 224 x 224 crops, 4 clips: before Mixed_3b / Mixed_4c / Mixed_5b):
    block forward and backward through the product module against the oracle block in fp64, RelaxedBernoulli noise
    injected, plus each piece on its own -- the three 1x1x1 convolutions (forward, dgrad, wgrad), the Gram / softmax /
    hop-weight / sample chain and the message passing.  The block has no ReLU and no BatchNorm, but its softmax sits on Gram
    logits that are sums of C/2 x H/2 x W/2 (18816 at the first site) products of max-pooled -- hence positive-mean -- maps:
    a common offset of ~70 (activations scaled by 0.1) to ~7000 (unit variance, what BatchNorm feeds the block in the
    model), and a softmax turns the ABSOLUTE rounding error of such a sum into a relative error of its output.  That is a
    property of fp32, not of a kernel: the yardstick for everything downstream of the softmax is therefore the fp32 CPU
    oracle's own distance from fp64 on the same data (bar = 4 x that, floor 2e-5; measured here 2e-5..9e-5 for both).  The
    pieces that do not pass through the softmax (the convolutions, the message passing) keep 1e-5."""
    from oracle import graph as ograph
    ops = pkg.engine.ops
    default = ops.get_conv_math()
    ops.set_conv_math(math)
    try:
        torch.manual_seed(100 + C)
        ref = ograph.TemporalGraphAug(C).double()
        B = 4
        x = torch.randn(B, C, T, HW, HW) * xscale
        u = torch.rand(B, T, T)
        dout = torch.randn(B, C, T, HW, HW)
        xr = x.double().requires_grad_(True)
        yr = ref(xr, u=u.double())
        yr.backward(dout.double())
        # the fp32 CPU oracle on the same data: how far fp32 lands from fp64 through this softmax
        ref32 = ograph.TemporalGraphAug(C)
        ref32.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        x32 = x.clone().requires_grad_(True)
        y32 = ref32(x32, u=u)
        y32.backward(dout)
        e32 = max([rel_err(y32, yr), rel_err(x32.grad, xr.grad)] +
                  [rel_err(p32.grad, pr.grad) for p32, pr in zip(ref32.parameters(), ref.parameters())])
        bar = max(2e-5, 4 * e32)
        # ---- the product module, reference-shaped API
        tg = pkg.lib.ops.module_wrappers.temporal_graph
        aug = tg.TemporalGraphAug(C)
        aug.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        aug.to(DEV).train()
        aug.noise = u.to(DEV)
        from importlib import import_module
        tp = import_module('video-graph-ssl_amd.engine.tape')
        tape, xv = tp.Tape(True), tp.Var(x.to(DEV), True)
        yv = aug.fwd(tape, xv)
        yv.grad = dout.to(DEV)
        tape.backward()
        assert rel_err(yv.t, yr) < bar
        assert rel_err(xv.grad, xr.grad) < bar
        for (n, p), (_, pr) in zip(aug.named_parameters(), ref.named_parameters()):
            assert rel_err(p.grad, pr.grad) < bar, n
        # ---- the pieces, through the op wrappers
        wq, wk, wg = (ref.state_dict()[n] for n in ('g_q.0.weight', 'g_k.0.weight', 'gcns.0.conv.weight'))
        xq = x.to(DEV)
        for w64 in (wq, wk, wg):
            w = w64.float().to(DEV)
            plan = ops.conv_plan(tuple(x.shape), w.shape[0], 1, 1, 0, DEV)
            yw = F.conv3d(x.double(), w64)
            assert rel_err(ops.conv_fwd(plan, xq, ops.conv_pack(plan, 0, w), w_raw=w), yw) < 1e-5
            dy = torch.randn(yw.shape)
            dxr = F.conv_transpose3d(dy.double(), w64)
            assert rel_err(ops.conv_dgrad(plan, dy.to(DEV), ops.conv_pack(plan, 1, w), w_raw=w), dxr) < 1e-5
            dwr = torch.einsum('bkthw,bcthw->kc', dy.double(), x.double()).reshape(w64.shape)
            dw = torch.zeros_like(w)
            ops.conv_wgrad(plan, xq, dy.to(DEV), dw, accumulate=True)
            assert rel_err(dw, dwr) < 1e-5
        # Gram + softmax + hop weights + sample, fp64 oracle on the same pooled maps
        gq = F.max_pool3d(F.conv3d(x.double(), wq), (1, 2, 2))
        gk = F.max_pool3d(F.conv3d(x.double(), wk), (1, 2, 2))
        gqr, gkr = gq.clone().requires_grad_(True), gk.clone().requires_grad_(True)
        sim_r = F.softmax(torch.matmul(gqr.transpose(2, 1).reshape(B, T, -1), gkr.transpose(2, 1).reshape(B, T, -1).permute(0, 2, 1)), -1)
        pre_r = ref.hop_weighted(sim_r, ograph.hop_distance(T, 3))
        adj_r = ograph.relaxed_bernoulli_rsample(pre_r, u.double(), 1.0)
        dadj = torch.randn(B, T, T)
        adj_r.backward(dadj.double())
        gqd, gkd = gq.float().to(DEV).contiguous(), gk.float().to(DEV).contiguous()
        sim, pre, adj = ops.graph_adj_fwd(gqd, gkd, u.to(DEV), 3, 0.5, 1.0)
        assert rel_err(sim, sim_r) < bar and rel_err(pre, pre_r) < bar and rel_err(adj, adj_r) < bar
        dgq, dgk = ops.graph_adj_bwd(dadj.to(DEV), gqd, gkd, sim, pre, adj, 3, 0.5, 1.0)
        assert rel_err(dgq, gqr.grad) < bar and rel_err(dgk, gkr.grad) < bar
        # message passing (einsum + skip) and its two gradients
        s = torch.randn(B, C, T, HW, HW)
        sr, ar = s.double().requires_grad_(True), adj_r.detach().clone().requires_grad_(True)
        outr = torch.einsum('bij,bcjhw->bcihw', ar, sr) + sr
        outr.backward(dout.double())
        adjd = adj_r.detach().float().to(DEV)
        out = ops.graph_gcn_fwd(adjd, s.to(DEV))
        assert rel_err(out, outr) < 1e-5
        ds, da = ops.graph_gcn_bwd(adjd, s.to(DEV), dout.to(DEV))
        assert rel_err(ds, sr.grad) < 1e-5 and rel_err(da, ar.grad) < 2e-5
        print('graph site C=%d T=%d %s x%.1f: fp32 CPU oracle vs fp64 %.1e -> bar %.1e' % (C, T, math, xscale, e32, bar))
    finally:
        ops.set_conv_math(default)


# ------------------------------------------------------------------------------------------------ configs[0]
YAML_VISUAL_MOCO_KEYS = """
MODEL:
  BACKBONE_TYPE: '3D'
  BACKBONE: 'S3D'
  PRETRAINED: False
  DROPOUT: 0.
INPUT:
  BASE_SIZE: [112, 112]
  CROP_SIZE: [112, 112]
  VIDEO_LENGTH: 16
SOLVER:
  OPTIMIZER_NAME: 'SGD'
  BASE_LR: 0.06
  LR_SCHEDULER: 'step'
  STEPS: [80, 120, 160]
  WARMUP_FACTOR: 0.01
  WARMUP_ITERS: 10
  MAX_EPOCHS: 200
  WEIGHT_DECAY: 0.0005
CONTRAST:
  MEM_TYPE: 'moco'
  NCE_K: 16384
  NCE_T: 0.07
  ALPHA: 0.999
CROSS:
  FEAT_DIM: 128
  HEAD_TYPE: 'mlp'
"""


@pytest.mark.parametrize('math,backbone,frames', [('bf16x6', 'S3D', 16), ('bf16x6', 'R2P1D18', 8), ('f32', 'R2P1D18', 8)])
def test_configs0_moco_plumbing_steps_from_yaml(pkg, tmp_path, math, backbone, frames):
    """BASELINE configs[0]: configs/visual_moco.yaml's keys (abridged to the ones the pre-training path reads; the file
    itself is not on the GPU box) -> get_defaults().merge_from_file -> MoCoTrainer, 2 clips x 112 x 112, queue 256: the
    YAML's own backbone (S3D, 16 frames: S3D needs T >= 16 for its (2,H,W) average pool) and R(2+1)D-18 at the
    8 frames BASELINE names.  Three iterations (eager, eager, hipGraph), teacher-forced from oracle.moco.moco_train_step in
    fp64: loss / logits / q / enqueued rows <= 1e-3, queue pointer and untouched rows exact, top-k rank counter exact on
    rows without ties."""
    from oracle import moco as omoco
    ops = pkg.engine.ops
    default = ops.get_conv_math()
    ops.set_conv_math(math)
    try:
        p = tmp_path / 'visual_moco.yaml'
        p.write_text(YAML_VISUAL_MOCO_KEYS)
        cfg = pkg.get_defaults()
        cfg.merge_from_file(str(p))
        b, K = 2, 256
        cfg.merge_from_list(['CONTRAST.NCE_K', K, 'MODEL.BACKBONE', backbone, 'INPUT.VIDEO_LENGTH', frames])
        cfg.freeze()
        tr = pkg.MoCoTrainer(cfg, DEV, use_graph=True, seed=1)
        state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
        mem0 = tr.contrast.memory.detach().cpu().clone()
        f0 = omoco.warmup_multistep_factor(0, (80, 120, 160), 0.1, 0.01, 10)
        m64, e64, c64, o64 = parity.oracle_moco(backbone, 128, K, frames, state, mem0, f0, double=True)
        crit = omoco.NCESoftmaxLoss()
        gen = torch.Generator().manual_seed(21)
        f32 = lambda sd: {k: (v.float() if v.dtype.is_floating_point else v) for k, v in sd.items()}
        for step in range(3):
            images = torch.randn(b, 6, frames, 112, 112, generator=gen)
            sh = torch.randperm(b, generator=gen)
            tr.model.load_state_dict(f32(m64.state_dict()))
            tr.model_ema.load_state_dict(f32(e64.state_dict()))
            tr.contrast.memory.copy_(c64.memory.float())
            tr.optimizer.buf.copy_(torch.cat([F.pad(
                o64.state[q]['momentum_buffer'].reshape(-1).float() if q in o64.state and 'momentum_buffer' in o64.state[q]
                else torch.zeros(q.numel()), (0, (-q.numel()) % 256)) for q in m64.parameters()]))
            mem_before = tr.contrast.memory.detach().cpu().clone()
            out = tr.train_step(images.to(DEV), shuffle_ids=sh)
            want = omoco.moco_train_step(m64, e64, c64, crit, o64, images.double(), 0.999, shuffle_ids=sh)
            torch.cuda.synchronize()
            errs = dict(loss=rel_err(out['loss'].reshape(()), want['loss']), logits=rel_err(out['logits'], want['logits']),
                        q=rel_err(out['q'], want['q']))
            lo, hi = b * step, b * (step + 1)
            mem = tr.contrast.memory.detach().cpu()
            assert int(tr.ptr_dev) == c64.index == tr.contrast.index == hi % K
            errs['queue_rows'] = rel_err(mem[lo:hi], c64.memory[lo:hi])
            assert torch.equal(mem[hi:], mem_before[hi:]) and torch.equal(mem[:lo], mem_before[:lo])
            for k_, v in errs.items():
                assert v < 1e-3, (math, backbone, step, k_, v)
            # rank counter (accuracy(), tools/...dis.py:428): exact wherever the positive does not tie a negative within 1e-4
            wl = want['logits']
            rank = (wl[:, 1:] >= wl[:, :1]).sum(1)
            clear = ((wl[:, 1:] - wl[:, :1]).abs().min(1).values > 1e-4 * wl.abs().max())
            assert torch.equal(out['rank'].cpu().long()[clear], rank[clear])
            # BatchNorm running statistics of both encoders: forward quantities
            for mod, ref in ((tr.model, m64), (tr.model_ema, e64)):
                rsd = ref.state_dict()
                for k_, v in mod.state_dict().items():
                    if k_.endswith('running_var'):
                        assert rel_err(v, rsd[k_]) < 1e-3, (step, k_)
        assert tr._segments[0].graph is not None
        tr.close()
    finally:
        ops.set_conv_math(default)


# ------------------------------------------------------------------------------------------------ bf16x6 on adversarial data
def test_bf16x6_on_wide_range_tiny_and_cancelling_operands(pkg):
    """The split-product arithmetic (x = hi + mid + lo in bf16, six products) on data randn never produces:
      (a) a per-reduction dynamic range of 2^40 (magnitudes log-uniform over 2^-20 .. 2^20 inside every dot product),
      (b) small magnitudes: operands of 2^-58 .. 2^-50 on both sides (products ~1e-33, still normal fp32 numbers),
      (c) operands log-uniform from 1e-35 up to 1 inside one reduction -- below 2^-110 the mid / lo parts of the split leave
          bf16's normal range, next to terms that do not,
      (d) exact-cancellation pairs: every product appears twice with opposite signs plus a 1e-6-sized remainder,
      (e) a reduction in which EVERY activation is ~1e-35 (documented limit: see below).
    Truth = fp64.  The fp32-MFMA kernel (an fmaf chain) is the yardstick; the bar for (a)-(d) is 1e-5 relative to
    sum |x w| per output (the natural scale of a dot product's rounding error), for forward, dgrad and wgrad, on an LDS-halo
    geometry, a temporal conv and a strided gather-kernel geometry.
    (e) is the one place where bf16x6 is NOT fp32-grade: when all of a reduction's operands sit below 2^-110 the low parts of
    the split are bf16 subnormals (or zero), so the result degrades towards a 1- or 2-part split.  No tensor of this path
    lives there (BatchNorm keeps activations O(1); gradients are >= 1e-12 with the loss scales used) -- the test pins the
    measured behaviour (finite, within 1e-2 of the truth = better than a single bf16 product) so that a change shows up."""
    ops = pkg.engine.ops
    default, auto = ops.get_conv_math(), ops.AUTOTUNE
    torch.manual_seed(77)

    def wide(shape, lo, hi):
        e = torch.rand(shape) * (hi - lo) + lo
        return torch.exp2(e) * torch.where(torch.rand(shape) < 0.5, -1.0, 1.0)
    cases = []
    for shape, K, k, s, p in (((2, 48, 4, 12, 12), 40, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                              ((2, 24, 6, 10, 10), 32, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
                              ((2, 5, 4, 9, 9), 20, (1, 3, 3), (1, 2, 2), (0, 1, 1))):
        wsh = (K, shape[1]) + k
        osh = tuple(F.conv3d(torch.zeros(shape), torch.zeros(wsh), None, s, p).shape)
        cases.append(('range 2^40', 1e-5, shape, K, k, s, p, wide(shape, -20, 20), wide(wsh, -20, 20), wide(osh, -20, 20)))
        cases.append(('small', 1e-5, shape, K, k, s, p, wide(shape, -58, -50), wide(wsh, -58, -50), wide(osh, -58, -50)))
        cases.append(('1e-35..1', 1e-5, shape, K, k, s, p, wide(shape, -116, 0), wide(wsh, -3, 3), wide(osh, -116, 0)))
        cases.append(('all 1e-35', 1e-2, shape, K, k, s, p, wide(shape, -117, -115), wide(wsh, -1, 1), wide(osh, -1, 1)))
        if shape[1] % 2 == 0:
            # cancellation: channels come in pairs (c, c+1) with x equal and w opposite up to a tiny remainder
            x, w = torch.randn(shape), torch.randn(wsh)
            x[:, 1::2] = x[:, 0::2]
            w[:, 1::2] = -w[:, 0::2] * (1 + 1e-6 * torch.randn(w[:, 0::2].shape))
            cases.append(('cancelling', 1e-5, shape, K, k, s, p, x, w, torch.randn(osh)))
    try:
        ops.AUTOTUNE = False
        for tag, bar, shape, K, k, s, p, x, w, dy in cases:
            x64, w64, dy64 = (t.double() for t in (x, w, dy))
            xr, wr = x64.clone().requires_grad_(True), w64.clone().requires_grad_(True)
            yr = F.conv3d(xr, wr, None, s, p)
            yr.backward(dy64)
            xa, wa = x64.abs().requires_grad_(True), w64.abs().requires_grad_(True)
            ya = F.conv3d(xa, wa, None, s, p)                    # sum |x w| per output; its gradients: sum |dy w|, sum |dy x|
            ya.backward(dy64.abs())
            res = {}
            for mode in ('f32', 'bf16x6'):
                ops.set_conv_math(mode)
                ops._conv_plan.cache_clear()
                plan = ops.conv_plan(shape, K, k, s, p, DEV)
                xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
                y = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), w_raw=wd)
                dx = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd), w_raw=wd)
                dw = torch.zeros_like(wd)
                ops.conv_wgrad(plan, xd, dyd, dw, accumulate=True)
                tiny = 1e-300
                res[mode] = dict(fwd=float(((y.double().cpu() - yr.detach()).abs() / ya.detach().clamp_min(tiny)).max()),
                                 dgrad=float(((dx.double().cpu() - xr.grad).abs() / xa.grad.clamp_min(tiny)).max()),
                                 wgrad=float(((dw.double().cpu() - wr.grad).abs() / wa.grad.clamp_min(tiny)).max()))
                assert all(torch.isfinite(t).all() for t in (y, dx, dw)), (tag, mode)
            print('%-11s %s k=%s: bf16x6 %s | f32 %s' % (tag, shape, k, {a: '%.1e' % v for a, v in res['bf16x6'].items()},
                                                         {a: '%.1e' % v for a, v in res['f32'].items()}))
            for key in ('fwd', 'dgrad', 'wgrad'):
                assert res['bf16x6'][key] < bar, (tag, shape, k, key, res)
                assert res['f32'][key] < 1e-5, (tag, shape, k, key, res)
    finally:
        ops.set_conv_math(default)
        ops.AUTOTUNE = auto
        ops._conv_plan.cache_clear()
