"""GPU parity of blocks / encoders / full training steps (HIP engine through the reference-shaped
API) against the golden fixtures and the CPU oracle.  Bar: 1e-3 relative fp32 (north_star)."""
import pytest
import torch
import torch.nn as nn

from conftest import rel_err
import parity

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture(params=['f32', 'bf16x6', 'bf16x3'])
def conv_math(request, pkg):
    """Both arithmetic modes of the conv kernels (include/gca_hip.h gca_set_conv_math) are held to north_star's 1e-3."""
    default = pkg.engine.ops.get_conv_math()
    pkg.engine.ops.set_conv_math(request.param)
    yield request.param
    pkg.engine.ops.set_conv_math(default)


def _engine(pkg):
    from importlib import import_module
    return import_module('video-graph-ssl_amd.engine.tape'), pkg.engine.layers


def _run(pkg, mod, x, dy=None):
    """forward (+ backward with upstream dy) of any module exposing fwd(tape, Var)."""
    tp, _ = _engine(pkg)
    tape = tp.Tape(dy is not None)
    xv = tp.Var(x, dy is not None)
    out = mod.fwd(tape, xv)
    if dy is not None:
        out.grad = dy
        tape.backward()
    return out.t, xv.grad


def test_r2plus1d_blocks_golden(pkg, golden):
    g = golden('blocks')
    r2 = pkg.lib.modeling.backbone.backbone_3d.resnet2p1d
    blk = r2.BasicBlock(16, 16)
    blk.load_state_dict(g.group('bb:w:'))
    blk.to(DEV).train()
    y, _ = _run(pkg, blk, g.t('bb:x').to(DEV))
    assert rel_err(y, g.t('bb:y')) < 1e-4
    after = g.group('bb:after:')
    for k, v in blk.state_dict().items():           # running stats after one train-mode forward
        assert rel_err(v.float(), after[k].float()) < 1e-4, k
    ds = nn.Sequential(r2.conv1x1x1(16, 32, 2), r2.HipBatchNorm3d(32))
    b2 = r2.BasicBlock(16, 32, stride=2, downsample=ds)
    b2.load_state_dict(g.group('bbs:w:'))
    b2.to(DEV).train()
    y2, _ = _run(pkg, b2, g.t('bbs:x').to(DEV))
    assert rel_err(y2, g.t('bbs:y')) < 1e-4


def test_s3d_blocks_golden(pkg, golden):
    g = golden('blocks')
    s3 = pkg.lib.modeling.backbone.backbone_3d.s3d_1
    sep = s3.SepConv3d(3, 64, kernel_size=7, stride=2, padding=3)
    sep.load_state_dict(g.group('sep:w:'))
    sep.to(DEV).train()
    y, _ = _run(pkg, sep, g.t('sep:x').to(DEV))
    assert rel_err(y, g.t('sep:y')) < 1e-4
    mix = s3.Mixed_3b()
    mix.load_state_dict(g.group('m3b:w:'))
    mix.to(DEV).train()
    ym, _ = _run(pkg, mix, g.t('m3b:x').to(DEV))
    assert rel_err(ym, g.t('m3b:y')) < 1e-4


def test_r3d_bottleneck_golden(pkg, golden):
    g = golden('blocks')
    r3 = pkg.lib.modeling.backbone.backbone_3d.resnet
    ds = nn.Sequential(r3.HipConv3d(16, 16, 1, 2), r3.HipBatchNorm3d(16))
    bt = r3.Bottleneck(16, 4, stride=2, downsample=ds)
    bt.load_state_dict(g.group('r3b:w:'))
    bt.to(DEV).train()
    y, _ = _run(pkg, bt, g.t('r3b:x').to(DEV))
    assert rel_err(y, g.t('r3b:y')) < 1e-4


def test_r2plus1d_tiny_fwd_bwd_golden(pkg, golden, conv_math):
    """Whole tiny R(2+1)D-10: train-mode output, input gradient, weight/BN gradients, running stats."""
    g = golden('r2p1d_tiny')
    r2 = pkg.lib.modeling.backbone.backbone_3d.resnet2p1d
    m = r2.generate_model(10, widen_factor=0.125)
    m.load_state_dict(g.group('r2t:w:'))
    m.to(DEV).train()
    x = g.x('r2t:xspec').to(DEV)
    yref = g.t('r2t:y_train')
    y, dx = _run(pkg, m, x, dy=(2 * yref).to(DEV))          # d/dy of sum(y^2), evaluated at the reference y
    assert rel_err(y, yref) < 1e-3
    if conv_math == 'f32':
        gerr, bar = rel_err, 1e-3
    else:
        # The split-product modes round differently from the fixture's fp32 chain, so (a) pre-activations within rounding
        # of zero flip their ReLU -- isolated O(1) errors in a max-norm, invisible in training -- and (b) bf16x3's 2^-17
        # products are amplified by the cancellation in BatchNorm's backward on this 8-channel model.  Gradients of
        # these modes are therefore held to a norm-wise bar (measured: bf16x6 2e-4; bf16x3 1e-3 on dx with the gather
        # kernels, 6.9e-3 once some layers run on the LDS-halo kernels -- another summation order, another set of flipped
        # masks; every conv of both kernel families meets 5e-5 per op in this mode, tests/test_gpu_ops.py).
        gerr = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm())
        bar = 4e-3 if conv_math == 'bf16x6' else 1.5e-2
    assert gerr(dx, g.t('r2t:dx')) < bar
    assert gerr(m.conv1_s.weight.grad, g.t('r2t:dw_conv1_s')) < bar
    assert gerr(m.layer4[0].conv2_t.weight.grad, g.t('r2t:dw_l4_conv2_t')) < bar
    assert gerr(m.fc.weight.grad, g.t('r2t:dw_fc')) < bar
    assert gerr(m.bn1_s.weight.grad, g.t('r2t:dg_bn1_s')) < bar
    after = g.group('r2t:after:')
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            assert rel_err(v, after[k]) < 1e-3, k


@pytest.mark.parametrize('tag,name,strip', [('s3d', 'S3D', True), ('r18', 'R2P1D18', True), ('r3d18', None, False)])
def test_full_size_encoders_seeded_golden(pkg, golden, conv_math, tag, name, strip):
    """Full-width S3D / R(2+1)D-18 / 3D-ResNet-18: seeded weights == the reference's (asserted when the
    fixture was generated), output on the stored clip must match the reference's."""
    g = golden('encoders_seeded')
    bb = pkg.lib.modeling.backbone.backbone_3d
    torch.manual_seed(int(g.t(tag + ':seed')))
    m = getattr(bb, name)() if name else bb.resnet.resnet18(sample_size=96, sample_duration=16)
    if strip:
        m.fc = pkg.engine.layers.HipIdentity()
    m.to(DEV).train()
    y, _ = _run(pkg, m, g.x(tag + ':xspec').to(DEV))
    # fp32 MFMA and bf16x6 (fp32-grade) meet north_star's 1e-3 on every encoder.  bf16x3 (the opt-in fast mode, 2^-17
    # products) does on R(2+1)D-18 and 3D-ResNet-18 but NOT on the 60-layer S3D chain (measured 5.3e-3): asserted here so
    # that the documented limit (conv arithmetic modes) stays true.
    assert rel_err(y, g.t(tag + ':y_train')) < (1e-2 if (conv_math, tag) == ('bf16x3', 's3d') else 1e-3)
    if tag == 's3d':
        assert rel_err(m.base[0].bn_s.running_mean, g.t('s3d:rm_base0_bn_s')) < 1e-3


def test_project_head_and_api_autograd(pkg, golden):
    """ProjectHead golden + the reference-style API: model(x) -> RGBMoCo -> NCESoftmaxLoss -> backward()."""
    g = golden('moco')
    head = pkg.lib.modeling.project_head.ProjectHead(24, 16, 'mlp')
    head.load_state_dict(g.group('head:w:'))
    head.to(DEV)
    y, _ = _run(pkg, head, g.t('head:x').to(DEV))
    assert rel_err(y, g.t('head:y')) < 1e-5

    from oracle import moco as omoco, wrappers as owrap
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 16, 8)
    torch.manual_seed(3)
    model, model_ema = pkg.create_visual_model(cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    contrast = pkg.create_contrast(cfg, 0).to(DEV)
    mem0 = contrast.memory.detach().cpu().clone()
    crit = pkg.create_criterion(cfg, 0)
    x = torch.randn(6, 3, 8, 64, 64)
    k = torch.nn.functional.normalize(torch.randn(6, 32))
    feat_q = model(x.to(DEV))                       # GraphWrapper.forward, differentiable as ONE autograd node
    logits, labels = contrast(feat_q, k.to(DEV))
    loss = crit(logits)
    loss.backward()
    om, _ = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'moco')
    om.load_state_dict(state)
    om.train()
    oc = omoco.RGBMoCo(32, K=16, T=0.07)
    oc.memory.copy_(mem0)
    ol, _ = oc(om(x), k)
    oloss = omoco.NCESoftmaxLoss()(ol)
    oloss.backward()
    assert rel_err(logits, ol) < 1e-3 and rel_err(loss, oloss) < 1e-3
    assert torch.equal(labels.cpu(), torch.zeros(6, dtype=torch.long)) and contrast.index == 6
    assert rel_err(contrast.memory, oc.memory) < 1e-6
    og = dict(om.named_parameters())
    for n, p in model.named_parameters():
        assert rel_err(p.grad, og[n].grad) < 2e-3, n


def test_moco_two_step_trace_golden(pkg, golden, conv_math):
    """Three full MoCo iterations (tools/train_video_contrast_dis.py:395-454) through MoCoTrainer against the
    trace recorded with the reference's own model / queue / criterion / optimiser classes."""
    g = golden('steps')
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 20, 8)
    tr = pkg.MoCoTrainer(cfg, DEV, use_graph=False, seed=0)
    w = g.group('mo:w:')
    tr.model.load_state_dict(w)
    tr.model_ema.load_state_dict(w)
    tr.contrast.memory.copy_(g.t('mo:mem0'))
    # bf16x3 (opt-in fast mode): two SGD updates of this 8-channel model amplify its 2^-17 products to 0.6-1.2e-2 on the third
    # iteration's q depending on which kernel family runs which layer (5.8e-3 with the gather kernels, 1.2e-2 with the stem
    # kernel on conv1_s; loss 1.0e-3); fp32 MFMA and bf16x6 stay inside 1e-3
    bar = 2e-2 if conv_math == 'bf16x3' else 1e-3
    for it in range(3):
        out = tr.train_step(g.x('mo:xspec%d' % it).to(DEV), shuffle_ids=g.t('mo:shuffle%d' % it))
        assert rel_err(out['loss'].reshape(()), g.t('mo:loss%d' % it)) < bar
        assert rel_err(out['logits'], g.t('mo:logits%d' % it)) < bar
        assert rel_err(out['q'], g.t('mo:q%d' % it)) < bar
    after = g.group('mo:after:')
    if conv_math == 'bf16x3':
        # weights norm-wise; the BatchNorm biases are skipped: they start at 0, so after three steps they ARE the summed
        # gradients (~3e-4 on this 8-channel model), of which bf16x3's rounding noise is 7-13 % -- why this mode is opt-in
        perr = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / (b.double().norm() + 1e-30))
        pbar = 2e-2
    else:
        perr, pbar = rel_err, bar
    for k, v in tr.model.state_dict().items():
        if v.dtype.is_floating_point and not (conv_math == 'bf16x3' and k.endswith('.bias')):
            assert perr(v, after[k]) < pbar, k
    ek = tr.model_ema.state_dict()
    for k, v in g.group('mo:afterk:').items():
        assert rel_err(ek[k], v) < bar, k
    assert rel_err(tr.contrast.memory, g.t('mo:mem3')) < bar
    assert int(tr.ptr_dev) == int(g.t('mo:ptr3')) == tr.contrast.index == 4


@pytest.mark.parametrize('use_graph,math', [(False, 'f32'), (True, 'bf16x6'), (True, 'f32')])      # (eager bf16x6: the first two steps of the graph runs)
def test_moco_steps_vs_oracle(pkg, use_graph, math):
    """5 MoCo iterations (hipGraph capture kicks in at the 3rd) incl. queue wrap, against the oracle run in
    fp64, next to the fp32 CPU oracle.  Forward / post-step state: 1e-3 max-norm.  Gradients: distribution
    bar of parity.check_grad_errors (isolated ReLU-boundary flips are inherent to fp32, see tests/parity.py);
    the HIP path must not be systematically worse than the fp32 CPU path."""
    default = pkg.engine.ops.get_conv_math()
    pkg.engine.ops.set_conv_math(math)
    try:
        _moco_steps_vs_oracle(pkg, use_graph)
    finally:
        pkg.engine.ops.set_conv_math(default)


def _moco_steps_vs_oracle(pkg, use_graph):
    parity.register_tiny(pkg)
    gen = torch.Generator().manual_seed(5)
    imgs = [torch.randn(8, 6, 8, 48, 48, generator=gen) for _ in range(5)]
    shs = [torch.randperm(8, generator=gen) for _ in range(5)]
    steps = parity.run_moco_parity(pkg, DEV, 'R2P1D10T', imgs, shs, feat_dim=32, K=20, T=8, use_graph=use_graph)
    med_hip, med_cpu = [], []
    for it, rec in enumerate(steps):
        assert rec['post'].pop('ptr') == 0
        assert max(rec['fwd'].values()) < 1e-3, (it, rec['fwd'])
        assert max(rec['post'].values()) < 1e-3, (it, rec['post'])
        parity.check_grad_errors(rec['post_params'], 'step %d updated parameters' % it)
        parity.check_grad_errors(rec['post_key_params'], 'step %d EMA key parameters' % it)
        m, _, _ = parity.check_grad_errors(rec['grad_hip'], 'step %d HIP gradients' % it)
        med_hip.append(m)
        med_cpu.append(parity._pct(list(rec['grad_cpu'].values()), 0.5))
    assert sorted(med_hip)[len(med_hip) // 2] < 10 * sorted(med_cpu)[len(med_cpu) // 2] + 1e-5, (med_hip, med_cpu)


def test_simsiam_loss_grads_golden(pkg, golden):
    g = golden('steps')
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, 'R2P1D10T', 'simsiam', 32, 16, 8)
    model, ema = pkg.create_visual_model(cfg)
    assert ema is None
    model.load_state_dict(g.group('ss:w:'))
    model.to(DEV).train()
    loss = model(g.x('ss:xspec').to(DEV))
    assert rel_err(loss, g.t('ss:loss')) < 1e-3
    loss.backward()
    sm = model.model
    assert rel_err(sm.prediction.l2.weight.grad, g.t('ss:dw_pred_l2')) < 2e-3
    assert rel_err(sm.projection.l1[0].weight.grad, g.t('ss:dw_proj_l1')) < 2e-3
    assert rel_err(sm.projection.l3[1].weight.grad, g.t('ss:dg_proj_l3_bn')) < 2e-3
    assert rel_err(sm.encoder.base_model.conv1_s.weight.grad, g.t('ss:dw_conv1_s')) < 2e-3


def test_temporal_graph_block_fwd_bwd_golden(pkg, golden, conv_math):
    g = golden('graph')
    tg = pkg.lib.ops.module_wrappers.temporal_graph
    for T in (2, 4, 8, 16):
        assert torch.equal(tg.TemporalGraph(tem_len=T, max_hop=3).temporal_graph, g.t('hop:T%d' % T))
    aug = tg.TemporalGraphAug(in_channels=32)
    aug.load_state_dict(g.group('aug:w:'))
    aug.to(DEV)
    aug.noise = g.t('aug:u').to(DEV)
    y, dx = _run(pkg, aug, g.t('aug:x').to(DEV), dy=g.t('aug:dy').to(DEV))
    assert rel_err(y, g.t('aug:y')) < 1e-4
    assert rel_err(dx, g.t('aug:dx')) < 1e-3
    assert rel_err(aug.gcns[0].conv.weight.grad, g.t('aug:dw_gcn')) < 1e-3
    assert rel_err(aug.g_q[0].weight.grad, g.t('aug:dw_gq')) < 1e-3
    assert rel_err(aug.g_k[0].weight.grad, g.t('aug:dw_gk')) < 1e-3
    aug.noise = g.t('aug:u_full_seed53').to(DEV)
    y2, _ = _run(pkg, aug, g.t('aug:x').to(DEV))
    assert rel_err(y2, g.t('aug:y_full_seed53')) < 1e-4


@pytest.mark.parametrize('use_graph,math', [(False, 'f32'), (True, 'bf16x6')])
def test_simsiam_trainer_steps_vs_oracle(pkg, use_graph, math):
    """SimSiamTrainer (tape engine, fused SGD, hipGraph) against _train_simsiam restated by the oracle in fp64
    (tools/train_video_contrast_dis.py:479-523).  Teacher-forced like parity.run_moco_parity: every step starts
    from the fp64 run's parameters and momentum, so a step's error is that step's error."""
    default = pkg.engine.ops.get_conv_math()
    pkg.engine.ops.set_conv_math(math)
    try:
        _simsiam_trainer_steps_vs_oracle(pkg, use_graph)
    finally:
        pkg.engine.ops.set_conv_math(default)


def _simsiam_trainer_steps_vs_oracle(pkg, use_graph):
    from oracle import moco as omoco, wrappers as owrap
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, 'R2P1D10T', 'simsiam', 32, 16, 8)
    tr = pkg.SimSiamTrainer(cfg, DEV, use_graph=use_graph, seed=11)
    state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    ref, _ = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'simsiam')
    ref.load_state_dict(state)
    ref.double().train()
    opt = omoco.make_optimizer(ref, 0.06, 0.9, 5e-4)
    f0 = omoco.warmup_multistep_factor(0, (80, 120, 160), 0.1, 0.01, 10)
    for gr in opt.param_groups:
        gr['lr'] *= f0
    gen = torch.Generator().manual_seed(21)
    medians = []
    for step in range(4):                                  # the 4th step is a hipGraph replay when use_graph
        x = torch.randn(8, 6, 8, 48, 48, generator=gen)
        tr.model.load_state_dict({k: (v.float() if v.dtype.is_floating_point else v) for k, v in ref.state_dict().items()})
        tr.optimizer.buf.copy_(torch.cat([torch.nn.functional.pad(
            opt.state[q]['momentum_buffer'].reshape(-1).float() if q in opt.state and 'momentum_buffer' in opt.state[q]
            else torch.zeros(q.numel()), (0, (-q.numel()) % 256)) for q in ref.parameters()]))
        out = tr.train_step(x.to(DEV))
        want = omoco.simsiam_train_step(ref, opt, x.double())
        # the loss is a mean of cosines (range [-1, 1]) that sits near 0 at initialisation: the bar is on that scale
        assert abs(float(out['loss']) - float(want['loss'])) < 1e-4, step
        g64 = {n: q.grad for n, q in ref.named_parameters()}
        errs = sorted(parity.rel(q.grad, g64[n]) for n, q in tr.model.named_parameters() if float(g64[n].abs().max()) > 1e-12)
        medians.append(errs[len(errs) // 2])
        assert errs[-1] < 2.5e-1, (step, errs[-1])
        rsd = ref.state_dict()
        for k, v in tr.model.state_dict().items():         # BN running statistics: forward quantities, strict bar
            if 'running_' in k:
                assert rel_err(v, rsd[k].float()) < 1e-3, (step, k)
    # Gradients: median per-tensor error 2e-5..5e-5 on a normal step.  With this seed step 2 has a pre-activation of
    # 1.1e-5 (typical 0.95) at the LAST block's output ReLU (tools/diag_simsiam.py): the HIP forward, 1e-5 away from
    # fp64 like any fp32 path, lands on the other side, and one flipped mask at the top of this 2048-element layer
    # moves every gradient below it by a few per cent.  A kernel bug would show on every step, a flip on one.
    assert sorted(medians)[2] < 2e-4 and max(medians) < 1e-1, medians


def test_checkpoint_resume_is_exact(pkg):
    """MoCoTrainer.state_dict() keeps the reference's checkpoint keys (tools/...dis.py:274-286; the optimizer entry is
    torch.optim.SGD's own format) plus the queue pointer; a fresh trainer that loads it continues bit-identically."""
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 20, 8)
    gen = torch.Generator().manual_seed(3)
    xs = [torch.randn(8, 6, 8, 48, 48, generator=gen).to(DEV) for _ in range(3)]
    a = pkg.MoCoTrainer(cfg, DEV, use_graph=False, seed=5)
    a.train_step(xs[0]); a.train_step(xs[1])
    sd = a.state_dict(epoch=7)
    assert set(sd) >= {'epoch', 'state_dict', 'optimizer', 'contrast', 'model_ema'} and sd['queue_index'] == 16
    assert set(sd['optimizer']) == {'state', 'param_groups'} and sd['optimizer']['param_groups'][0]['params'] == [0]
    # the optimizer entry loads into a real torch.optim.SGD built the reference's way (one group per parameter)
    ref_opt = torch.optim.SGD([{'params': [torch.nn.Parameter(p.detach().clone().cpu())]} for p in a.model.parameters()],
                              lr=0.1, momentum=0.9)
    ref_opt.load_state_dict({'state': {i: {'momentum_buffer': v['momentum_buffer'].cpu()} for i, v in sd['optimizer']['state'].items()},
                             'param_groups': [dict({k: v for k, v in g.items() if k in ref_opt.param_groups[0]}, params=g['params'])
                                              for g in sd['optimizer']['param_groups']]})
    import io
    buf = io.BytesIO()
    torch.save(sd, buf)                               # round trip through the file format
    sd2 = torch.load(io.BytesIO(buf.getvalue()), map_location='cpu', weights_only=False)
    b = pkg.MoCoTrainer(cfg, DEV, use_graph=False, seed=99)
    assert b.load_state_dict(sd2) == 7
    oa, ob = a.train_step(xs[2]), b.train_step(xs[2])
    assert torch.equal(oa['loss'], ob['loss']) and torch.equal(oa['logits'], ob['logits'])
    for (n, p), (_, q) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(p, q), n
    assert torch.equal(a.contrast.memory, b.contrast.memory) and int(a.ptr_dev) == int(b.ptr_dev) == 4


@pytest.mark.parametrize('math', ['f32', 'bf16x6'])
@pytest.mark.parametrize('name', ['S3D', 'R3D18'])
def test_full_width_encoder_backward_vs_fp64_oracle(pkg, name, math):
    """Forward AND backward of the full-width S3D (a3) and 3D-ResNet (a5) through the HIP engine against the oracle in
    fp64 on the same weights (the R(2+1)D family has its own step-level tests).  Features: 1e-3.  Gradients: through
    S3D's 77 BatchNorms, 13 max pools and ReLUs fp32 itself is chaotic -- the fp32 CPU oracle is 1e-2 (median over
    parameter tensors) away from fp64 (tools/diag_s3d_bwd.py) -- so the HIP path is held to the fp32 oracle's OWN error
    (x3), not to an absolute bar; the 3D-ResNet passes the absolute distribution bar of parity.check_grad_errors in fp32
    MFMA (median 7e-6: the fma chain + fp64 BN sums track fp64 far better than a standard fp32 implementation does -- the
    fp32 CPU oracle's own median is 5.8e-4).  bf16x6 is held, on both encoders, to the fp32 CPU oracle's own error (x3):
    measured on the 3D-ResNet 5.9e-4 median, i.e. exactly as far from fp64 as the reference's fp32 CPU path is
    (tools/diag_r3d_modes.py)."""
    default = pkg.engine.ops.get_conv_math()
    pkg.engine.ops.set_conv_math(math)
    try:
        _full_width_backward(pkg, name, math)
    finally:
        pkg.engine.ops.set_conv_math(default)


def _full_width_backward(pkg, name, math):
    from oracle import encoders as oenc
    bb = pkg.lib.modeling.backbone.backbone_3d
    torch.manual_seed(17)
    if name == 'S3D':
        m = bb.S3D()
        m.fc = pkg.engine.layers.HipIdentity()
        mk = lambda: oenc.S3D()
    else:
        m = bb.resnet.resnet18(sample_size=64, sample_duration=16)
        mk = lambda: oenc.R3D(18, sample_size=64, sample_duration=16)
    x = torch.randn(4, 3, 16, 64, 64)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}

    def reference(double):
        ref = mk()
        if name == 'S3D':
            ref.fc = torch.nn.Identity()
        ref.load_state_dict(sd)
        ref = (ref.double() if double else ref).train()
        xr = (x.double() if double else x.clone()).requires_grad_(True)
        yr = ref(xr)
        yr = yr.reshape(yr.shape[0], -1)
        return ref, xr, yr
    r64, x64, y64 = reference(True)
    dy = torch.randn(y64.shape, dtype=torch.float64)
    y64.backward(dy)
    m.to(DEV).train()
    for q in m.parameters():
        q.grad = None
    y, dx = _run(pkg, m, x.to(DEV), dy.float().to(DEV).reshape(-1, y64.shape[1]))
    assert rel_err(y.reshape(y64.shape), y64.detach().float()) < 1e-3
    g64 = {n: q.grad for n, q in r64.named_parameters()}
    errs = {n: parity.rel(q.grad, g64[n]) for n, q in m.named_parameters() if q.grad is not None and float(g64[n].abs().max()) > 1e-12}
    assert len(errs) > 50
    if name == 'S3D' or math != 'f32':
        r32, x32, y32 = reference(False)
        y32.backward(dy.float())
        e32 = {n: parity.rel(q.grad, g64[n]) for n, q in r32.named_parameters() if n in errs}
        med = lambda d: sorted(d.values())[len(d) // 2]
        assert med(errs) < 3 * med(e32) + 1e-4, (med(errs), med(e32))
        assert parity.rel(dx, x64.grad) < 3 * parity.rel(x32.grad, x64.grad) + 1e-4
    else:
        parity.check_grad_errors(errs)
        assert parity.rel(dx, x64.grad) < 2e-2


@pytest.mark.parametrize('tag,kw', [('nosub', dict(sub_sample=False)), ('avg', dict(max_pool=False)), ('bn', dict(bn_layer=True)),
                                    ('gcn3', dict(inter_channels=8, num_gcn_layers=3)), ('bias', dict(bias=True))])
def test_temporal_graph_block_constructor_options_golden(pkg, golden, tag, kw):
    """The TemporalGraphAug options no shipped config sets (temporal_graph.py:66-129): no sub-sampling, average pooling,
    BatchNorm behind the similarity convs, a 3-layer GCN stack, biased similarity convs -- whole-block forward + backward
    against the REFERENCE's own class (tests/golden/make_golden.py gen_graph_options), state-dict keys included."""
    g = golden('graph_options')
    tg = pkg.lib.ops.module_wrappers.temporal_graph
    aug = tg.TemporalGraphAug(in_channels=16, **kw)
    w = g.group(tag + ':w:')
    assert sorted(aug.state_dict().keys()) == sorted(w.keys())
    aug.load_state_dict(w)
    aug.to(DEV).train()
    aug.noise = g.t(tag + ':u').to(DEV)
    y, dx = _run(pkg, aug, g.t(tag + ':x').to(DEV), dy=g.t(tag + ':dy').to(DEV))
    assert rel_err(y, g.t(tag + ':y')) < 1e-4
    assert rel_err(dx, g.t(tag + ':dx')) < 1e-3
    grads = g.group(tag + ':g:')
    gmax = max(float(v.abs().max()) for v in grads.values())
    for n, p in aug.named_parameters():
        # (a bias / BatchNorm shift of g_k moves every logit of a softmax row by the same amount: its true gradient is exactly
        # zero and both sides hold rounding noise ~1e-5 -- hence the floor relative to the block's largest gradient)
        scale = max(float(grads[n].abs().max()), 1e-2 * gmax)
        assert float((p.grad.cpu() - grads[n]).abs().max()) < 1e-3 * scale, (n, gmax)
    if tag == 'bn':
        after = g.group(tag + ':after:')
        for k_, v in aug.state_dict().items():
            if 'running' in k_:
                assert rel_err(v, after[k_]) < 1e-4, k_
    with pytest.raises(NotImplementedError):
        tg.TemporalGraphAug(16, mask_frame=True)
    with pytest.raises(ValueError):
        tg.TemporalGraphAug(16, num_gcn_layers=2)


def test_reference_metric_api_accuracy_and_average_meter(pkg):
    """accuracy(output, target, topk) / AverageMeter with the reference's signatures (lib/evaluation/metric.py:9-24,44-67; the
    call of tools/train_video_contrast_dis.py:428) against the oracle's restatement of the top-k form, on InfoNCE-shaped
    logits (label 0) and on arbitrary labels, without leaving the device."""
    from oracle import moco as omoco
    ev = pkg.lib.evaluation
    torch.manual_seed(4)
    for b, ncol in ((32, 4097), (7, 65537), (5, 11)):
        out = torch.randn(b, ncol)
        out[:, 0] += 2.5                                   # some rows have the label on top, some do not
        for target in (torch.zeros(b, dtype=torch.long), torch.randint(0, ncol, (b,))):
            want = omoco.accuracy(out, target, topk=(1, 5))
            got = ev.accuracy(out.to(DEV), target.to(DEV), topk=(1, 5))
            assert all(g_.is_cuda and g_.shape == (1,) for g_ in got)
            assert [round(float(g_), 4) for g_ in got] == [round(float(w_), 4) for w_ in want], (b, ncol)
        _, _, rank = pkg.engine.ops.moco_logits_fwd(torch.nn.functional.normalize(torch.randn(b, 128)).to(DEV),
                                                    torch.nn.functional.normalize(torch.randn(b, 128)).to(DEV),
                                                    torch.nn.functional.normalize(torch.randn(ncol - 1, 128)).to(DEV), 1 / 0.07,
                                                    want_rank=True)
        assert len(ev.accuracy_from_rank(rank, (1, 5))) == 2
    m = ev.AverageMeter()
    m.update(3.0, 2); m.update(torch.tensor([6.0], device=DEV), 1)
    assert abs(float(m.avg) - 4.0) < 1e-6 and m.count == 3 and float(m.val) == 6.0
    with pytest.raises(ValueError):
        ev.accuracy(torch.randn(4, 5, 6).to(DEV), torch.zeros(4, dtype=torch.long).to(DEV))


@pytest.mark.parametrize('backbone,T', [('R2P1D10T', 8), ('S3D', 16)])
def test_dropout_branch_of_prepare_video_model(pkg, backbone, T):
    """MODEL.DROPOUT > 0 (visual_wrappers.py:109-110: the backbone's fc becomes nn.Dropout; it is also get_defaults()' value,
    0.5): the model builds, trains and -- with the same keep mask -- equals the oracle wrapper.  torch's device generator
    draws the mask, so the check re-seeds it and replays the draw."""
    from oracle import wrappers as owrap
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, backbone, 'moco', 32, 64, T)
    cfg.MODEL.DROPOUT = 0.5
    torch.manual_seed(9)
    model, ema = pkg.create_visual_model(cfg)
    fc = model.model.encoder.base_model.fc
    assert isinstance(fc, nn.Dropout) and fc.p == 0.5
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    size = 48 if backbone != 'S3D' else 64
    x = torch.randn(4, 3, T, size, size)
    torch.manual_seed(77)
    q = model(x.to(DEV))
    q.sum().backward()
    # the oracle with the SAME mask: replay the device draw
    torch.manual_seed(77)
    C = model.model.encoder.feature_dim
    shape = (4, C) if backbone != 'S3D' else (4, C, 1)       # S3D at 16 x 64 x 64: T' = 2 -> one (2,H,W) window per channel
    mask = (torch.empty(shape, device=DEV).bernoulli_(0.5) * 2.0).cpu()
    ref, _ = owrap.create_visual_model(backbone, T, 32, 'mlp', 'moco', dropout=0.5)
    ref.load_state_dict(sd)
    ref.double().train()

    class FixedMask(nn.Module):
        def forward(self, t):
            return t * mask.double().reshape(t.shape)
    ref.model.encoder.base_model.fc = FixedMask()
    qr = ref(x.double())
    qr.sum().backward()
    assert rel_err(q, qr) < 1e-3
    gr = dict(ref.named_parameters())
    errs = [parity.rel(p.grad, gr[n].grad) for n, p in model.named_parameters() if float(gr[n].grad.abs().max()) > 0]
    # S3D's 77 BatchNorms over 4 small clips + 13 max pools make its fp32 gradients chaotic for any implementation (the fp32
    # CPU oracle itself sits 4e-2..1e-1 from fp64: tests/test_gpu_configs.py); a wrong factor in the dropout tail would be O(1)
    assert sorted(errs)[len(errs) // 2] < (1e-3 if backbone != 'S3D' else 5e-2)
    # eval mode: dropout is the identity
    model.eval()
    ref.eval()
    ref.model.encoder.base_model.fc = nn.Identity()
    with torch.no_grad():
        assert rel_err(model(x.to(DEV)), ref(x.double())) < 1e-3
    # and the default config (DROPOUT 0.5, no YAML) now constructs
    d = pkg.get_defaults()
    d.merge_from_list(['MODEL.BACKBONE', 'R2P1D10T', 'CONTRAST.MEM_TYPE', 'moco'])
    assert isinstance(pkg.create_visual_model(d)[0].model.encoder.base_model.fc, nn.Dropout)
