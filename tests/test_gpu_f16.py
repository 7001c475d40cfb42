"""fp16-STORAGE path (BASELINE configs[4]: 3D-ResNet-50, fp16 MFMA): every kernel that touches a feature map, with the map
stored as IEEE fp16, against the SAME op in fp32 on the SAME (already fp16-representable) operands.

Declared tolerances (the reference has no runnable fp16 path -- apex is not installable, SURVEY.md section 2 -- so the
bar is "fp32 result of the fp16-rounded operands, rounded once"):
  * conv fwd / dgrad (outputs stored fp16): max-normalised error <= 1.5e-3  (one fp16 rounding is 2^-11 = 4.9e-4 of the
    element, accumulation is fp32 in the MFMA);
  * conv wgrad, BN statistics / dgamma / dbeta, pooled features (outputs fp32): <= 2e-4 (fp32 accumulation of fp16 operands;
    differences to the fp32 kernels are summation order only);
  * BN apply / backward, max pool, average-pool backward (outputs fp16): <= 1.5e-3;
  * a whole 3D-ResNet MoCo iteration against the fp32 oracle: features <= 1e-2, loss <= 1e-2 relative -- the gap is the
    accumulated storage rounding of ~20 feature maps, not arithmetic (the convs accumulate in fp32).
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_H = 1.5e-3        # outputs stored fp16
TOL_F = 2e-4          # fp32 outputs of fp16 operands


@pytest.fixture(scope='module')
def ops(pkg):
    return pkg.engine.ops


def _h(t):
    """fp16-representable fp32 tensor."""
    return t.half().float()


CONV_CASES = [
    # the layer kinds of the 3D-ResNet-50 (resnet.py): 7x7x7 stem stride (1,2,2), bottleneck 1x1x1, 3x3x3 (stride 1 and 2),
    # strided 1x1x1 shortcut -- at sizes that reach the halo kernels (C >= 32, enough columns) and the gather kernels
    ((2, 3, 8, 32, 32), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),
    ((2, 64, 4, 28, 28), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ((2, 64, 6, 28, 28), 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((2, 256, 4, 14, 14), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ((2, 128, 4, 14, 14), 512, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
    ((3, 40, 3, 9, 11), 50, (3, 3, 3), (1, 1, 1), (1, 1, 1)),          # ragged tiles, channel tails
    ((2, 24, 5, 10, 10), 36, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ((4, 64, 1, 1, 1), 24, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ((2, 3, 8, 64, 64), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),           # large enough for the stem kernel (conv3d_stem.hip)
    ((3, 40, 2, 12, 12), 200, (1, 1, 1), (1, 1, 1), (0, 0, 0)),         # pointwise GEMM kernel (conv3d_pw.hip): channel and row tails
    ((2, 512, 2, 16, 16), 96, (1, 1, 1), (1, 1, 1), (0, 0, 0)),         # ... 16 k-tiles, 4 position tiles per clip
]


@pytest.mark.parametrize('shape,K,k,s,p', CONV_CASES)
def test_conv_f16_storage_vs_f32_of_rounded_operands(ops, shape, K, k, s, p):
    torch.manual_seed(0)
    x = _h(torch.randn(shape))
    w = _h(torch.randn((K, shape[1]) + tuple(k)) * (2.0 / (shape[1] * k[0] * k[1] * k[2])) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr.double(), wr.double(), None, s, p)
    dy = _h(torch.randn(yr.shape))
    yr.backward(dy.double())
    plan = ops.conv_plan(shape, K, k, s, p, torch.device(DEV), act_f16=True)
    assert plan.g.act_f16 == 1
    xd, dyd, wd = x.to(DEV).half(), dy.to(DEV).half(), w.to(DEV)
    y, (ss, sq) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None, stats=True)
    assert y.dtype is torch.float16
    assert rel_err(y.float(), yr) < TOL_H
    # BatchNorm statistics come from the fp32 accumulators (before the store rounds)
    assert float((ss.sum(1).double().cpu() - yr.sum((0, 2, 3, 4))).abs().max()) <= 1e-5 * float(yr.abs().sum((0, 2, 3, 4)).max())
    assert rel_err(sq.sum(1), (yr * yr).sum((0, 2, 3, 4))) < TOL_F
    dx = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd))
    assert dx.dtype is torch.float16
    assert rel_err(dx.float(), xr.grad) < TOL_H
    base = _h(torch.randn(shape)).to(DEV).half()
    acc = base.clone()
    ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd), acc, accumulate=True)
    assert rel_err(acc.float(), xr.grad + base.float().cpu()) < TOL_H
    dw = torch.zeros_like(wd)
    ops.conv_wgrad(plan, xd, dyd, dw, accumulate=True)
    assert dw.dtype is torch.float32
    assert rel_err(dw, wr.grad) < TOL_F
    if k == (1, 1, 1) and s == (1, 1, 1) and (shape[2] * shape[3] * shape[4]) % 8 == 0 and shape[2] * shape[3] * shape[4] >= 128:
        assert (plan.cfg(0)[3] >> 17) & 1 and (plan.cfg(1)[3] >> 17) & 1              # the pointwise GEMM kernel ran both passes
        for name in ('fwd', 'dgrad'):                                                 # ... and agrees with the gather kernels
            setattr(plan.g, 'tune_%s_bm' % name, 64)
        plan.refresh()
        assert not (plan.cfg(0)[3] >> 17) & 1
        y2, (ss2, sq2) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None, stats=True)
        dx2 = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd))
        assert rel_err(y.float(), y2.float()) < TOL_H and rel_err(dx.float(), dx2.float()) < TOL_H
        assert rel_err(ss.sum(1), ss2.sum(1)) < 1e-3 and rel_err(sq.sum(1), sq2.sum(1)) < TOL_F
    # a plan built for one storage type refuses the other instead of misreading it
    with pytest.raises(TypeError):
        ops.conv_fwd(plan, x.to(DEV), ops.conv_pack(plan, 0, wd))


def test_conv_f16_every_halo_box_and_gather_tile(ops):
    """Every launch shape the tuner may pin for an fp16 layer gives the same numbers (to fp32 accumulation order)."""
    torch.manual_seed(1)
    shape, K, k, s, p = (2, 64, 4, 28, 28), 96, (3, 3, 3), (1, 1, 1), (1, 1, 1)
    x = torch.randn(shape, device=DEV).half()
    w = torch.randn((K, 64, 3, 3, 3), device=DEV) * 0.03
    plan = ops.ConvPlan(*shape, K, k, s, p, torch.device(DEV), act_f16=True)
    plan.tuned = [True, True, True]
    dy = torch.randn(plan.out_shape, device=DEV).half()
    yr = F.conv3d(x.double(), w.half().double(), None, s, p)
    seen = set()
    for which, name in ((0, 'fwd'), (1, 'dgrad')):
        M = K if which == 0 else shape[1]
        cands = [(0, 0)] + [(c[0], c[2]) for c in plan._halo_candidates(which, M)] + [(64, 0), (128, 0)]
        ref = None
        for code, box in cands:
            setattr(plan.g, 'tune_%s_bm' % name, code)
            setattr(plan.g, 'tune_%s_box' % name, box)
            plan.refresh()
            seen.add((which, (plan.cfg(which)[3] >> 14) & 1))
            if which == 0:
                out = ops.conv_fwd(plan, x, ops.conv_pack(plan, 0, w), None).float()
                assert rel_err(out, yr) < TOL_H
            else:
                out = ops.conv_dgrad(plan, dy, ops.conv_pack(plan, 1, w)).float()
            if ref is None:
                ref = out
            assert rel_err(out, ref) < TOL_H
        setattr(plan.g, 'tune_%s_bm' % name, 0)
        setattr(plan.g, 'tune_%s_box' % name, 0)
        plan.refresh()
    assert (0, 1) in seen and (0, 0) in seen and (1, 1) in seen and (1, 0) in seen      # halo AND gather kernels ran


@pytest.mark.parametrize('shape,res', [((3, 6, 2, 5, 7), True), ((2, 8, 2, 4, 4), False), ((2, 5, 4, 50, 52), True),
                                       ((4, 16, 2, 8, 8), True)])
def test_bn_f16_storage_vs_f32_kernels(ops, shape, res):
    """bn_train_fwd / bn_apply / bn_bwd (single-launch small form and the three-pass form) on fp16 maps against the fp32
    kernels on the same values."""
    torch.manual_seed(2)
    N, Cc = shape[:2]
    SP = shape[2] * shape[3] * shape[4]
    x = _h(torch.randn(shape) * 1.5 + 0.3).to(DEV)
    r = _h(torch.randn(shape)).to(DEV) if res else None
    dz = _h(torch.randn(shape)).to(DEV)
    gam = torch.rand(Cc, device=DEV) + 0.5
    bet = torch.randn(Cc, device=DEV)

    def run(half):
        c = (lambda t: None if t is None else t.half()) if half else (lambda t: t)
        rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        ss, sq = ops.bn_stats(c(x), N, Cc, SP)
        z, mean, invstd, scale, shift = ops.bn_train_fwd(ss, sq, N * SP, gam, bet, 1e-5, 0.1, rm, rv, nbt, c(x), c(r), True,
                                                         N, Cc, SP)
        z2 = ops.bn_apply(c(x), scale, shift, c(r), True, N, Cc, SP)
        assert torch.equal(z, z2)
        dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        dres = torch.empty_like(z) if res else None
        dx = ops.bn_bwd(c(dz), z, c(x), gam, mean, invstd, 1, N, Cc, SP, dg, db, dres, False)
        out = dict(z=z, mean=mean, invstd=invstd, rm=rm, rv=rv, dx=dx, dg=dg, db=db)
        if res:
            out['dres'] = dres
            acc = c(dz).clone()
            ops.bn_bwd(c(dz), z, c(x), gam, mean, invstd, 1, N, Cc, SP, torch.zeros_like(dg), torch.zeros_like(db), acc, True)
            out['dres_acc'] = acc
        else:
            dg2, db2 = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
            out['dx_mode2'] = ops.bn_bwd(c(dz), None, c(x), gam, mean, invstd, 2, N, Cc, SP, dg2, db2, None, False, scale, shift)
        return out
    f, h = run(False), run(True)
    for key in f:
        stored_half = key in ('z', 'dx', 'dres', 'dres_acc', 'dx_mode2')
        assert h[key].dtype is (torch.float16 if stored_half else torch.float32), key
        # the ReLU mask of an element whose fp32 output is within rounding of 0 may differ: bound the COUNT of large misses
        if key.startswith('d'):
            big = ((h[key].float() - f[key]).abs() > 5e-3 * f[key].abs().max()).float().mean()
            assert float(big) < 2e-3, (key, float(big))
        else:
            assert rel_err(h[key].float(), f[key]) < (TOL_H if stored_half else TOL_F), key


@pytest.mark.parametrize('shape,k,s,p', [((2, 5, 8, 18, 18), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                         ((2, 4, 6, 20, 24), (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # LDS-tiled backward
                                         ((2, 4, 3, 9, 10), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                         ((1, 3, 4, 6, 6), (2, 2, 2), (2, 2, 2), (0, 0, 0)),
                                         ((1, 2, 5, 7, 9), (3, 2, 1), (1, 1, 1), (1, 1, 0))])
def test_pools_f16_storage_are_exact_selections(ops, shape, k, s, p):
    """Max pooling selects: on fp16 input the fp16 kernel must return exactly the values and argmax of the fp32 kernel; its
    backward scatters fp16 gradients (sums of <= 8 of them, fp32 arithmetic, one rounding)."""
    torch.manual_seed(3)
    x = _h(torch.randn(shape)).to(DEV)
    x[0, 0, :2] = 0.0                                    # ties
    plan = ops.pool_plan(shape, k, s, p)
    yf, af = ops.maxpool_fwd(plan, x)
    yh, ah = ops.maxpool_fwd(plan, x.half())
    assert yh.dtype is torch.float16 and torch.equal(yh.float(), yf) and torch.equal(ah, af)
    dy = _h(torch.randn(yf.shape)).to(DEV)
    dxf = ops.maxpool_bwd(plan, dy, af)
    dxh = ops.maxpool_bwd(plan, dy.half(), ah)
    assert dxh.dtype is torch.float16 and rel_err(dxh.float(), dxf) < TOL_H
    acc = x.half().clone()
    ops.maxpool_bwd(plan, dy.half(), ah, acc, True)
    assert rel_err(acc.float(), dxf + x) < TOL_H
    # fused BN+ReLU producer
    sc, sh = torch.rand(shape[1], device=DEV) + 0.5, torch.randn(shape[1], device=DEV)
    pf, _ = ops.maxpool_fwd(plan, x, scale=sc, shift=sh)
    ph, _ = ops.maxpool_fwd(plan, x.half(), scale=sc, shift=sh)
    assert rel_err(ph.float(), pf) < TOL_H


def test_avgpool_axpy_cast_f16(ops):
    torch.manual_seed(4)
    x = _h(torch.randn(3, 7, 4, 5, 6)).to(DEV)
    wt = torch.tensor([1., 2., 2., 1.], device=DEV)
    for w_, norm in ((None, 1.0 / 120), (wt, 1.0 / 180)):
        yf = ops.wavgpool_fwd(x, w_, norm)
        yh = ops.wavgpool_fwd(x.half(), w_, norm)
        assert yh.dtype is torch.float32 and rel_err(yh, yf) < 1e-6          # fp32 sums of the same values
        dy = torch.randn(3, 7, device=DEV)
        dxf = ops.wavgpool_bwd(dy, w_, norm, tuple(x.shape))
        dxh = ops.wavgpool_bwd(dy, w_, norm, tuple(x.shape), torch.float16)
        assert dxh.dtype is torch.float16 and rel_err(dxh.float(), dxf) < TOL_H
    a, b = x.half().clone(), _h(torch.randn(x.shape)).to(DEV).half()
    ref = (a.float() + 0.5 * b.float()).half()
    ops.axpy(a, b, 0.5)
    assert torch.equal(a, ref)
    with pytest.raises(TypeError):
        ops.axpy(a, b.float(), 1.0)
    # the clip cast: a channel slice of a (b, 6, T, H, W) batch, as the trainer passes it
    clips = torch.randn(3, 6, 4, 6, 8, device=DEV)
    for view in torch.chunk(clips, 2, dim=1):
        assert torch.equal(ops.cast_f16(view), view.half())
    assert torch.equal(ops.cast_f16(clips), clips.half())
    # ops without an fp16 kernel fail loudly instead of misreading the buffer
    with pytest.raises(TypeError):
        ops.relu_fwd(x.half())
    with pytest.raises(TypeError):
        ops.bias_grad(x.half(), 3, 7, 120, torch.zeros(7, device=DEV))


def _r3d_small(pkg, backbone, size, frames):
    from oracle import encoders as oenc
    bb = pkg.lib.modeling.backbone.backbone_3d
    depth = int(backbone[3:])
    name = '%sS%d' % (backbone, size)                       # the same network with its AvgPool3d window sized for these clips
    bb.register(name, lambda: getattr(bb.resnet, 'resnet%d' % depth)(sample_size=size, sample_duration=frames))
    oenc.BACKBONES[name] = lambda: oenc.R3D(depth, size, frames)
    return name


def test_r3d18_moco_iteration_f16_storage_vs_fp64_oracle(pkg, ops):
    """One MoCo iteration of a 3D-ResNet-18 (resnet.py BasicBlock) with fp16 feature maps against the fp64 CPU oracle
    (oracle/moco.py) on the same clips and weights: loss, logits, features, queue and BN running statistics within 1e-2
    (fp32 storage: 1e-3).  Gradients are compared in the next test, against what the same rounding does to the fp32 path."""
    import parity
    backbone = _r3d_small(pkg, 'R3D18', 64, 8)
    default = ops.get_conv_math()
    ops.set_conv_math('fp16')
    try:
        g = torch.Generator().manual_seed(5)
        images = [torch.randn(4, 6, 8, 64, 64, generator=g)]
        shuffles = [torch.randperm(4, generator=g)]
        seen = []
        orig = ops.cast_f16
        ops.cast_f16 = lambda x: (seen.append(tuple(x.shape)), orig(x))[1]
        try:
            rec = parity.run_moco_parity(pkg, torch.device(DEV), backbone, images, shuffles, feat_dim=128, K=64, T=8,
                                         use_graph=False, with_cpu32=False)[0]
        finally:
            ops.cast_f16 = orig
    finally:
        ops.set_conv_math(default)
    assert len(seen) == 2 and seen[0] == (4, 3, 8, 64, 64)          # query and key clips entered the fp16 path
    assert rec['post'].pop('ptr') == 0
    assert max(rec['fwd'].values()) < 1e-2, rec['fwd']
    assert rec['post']['queue'] < 1e-2 and rec['post']['buffers'] < 1e-2, rec['post']


@pytest.mark.parametrize('backbone,batch,frames,size', [('R3D18', 8, 16, 112), ('R3D50', 4, 16, 112)])
def test_f16_storage_distance_from_fp32_is_what_storage_rounding_explains(pkg, ops, backbone, batch, frames, size):
    """A freshly initialised 3D-ResNet MoCo step is ill-conditioned (BatchNorm over a handful of clips, a saturated softmax):
    in the fp32 path itself, ONE fp16-sized rounding of the input clip (relative 2^-11) moves the features by ~1e-3..1e-2
    and single gradient tensors by tens of per cent (tools/f16_model_check.py).  So the model-level bar for the fp16-storage
    path -- which rounds ~20 (R3D-18) to ~100 (R3D-50) maps, and their gradients -- is relative to that measured
    sensitivity, on the same weights and clips: features within 8x, median per-tensor gradient error within 4x."""
    import parity
    name = _r3d_small(pkg, backbone, size, frames)
    dev = torch.device(DEV)
    cfg = parity.make_cfg(pkg, name, 'moco', 128, 4096, frames)
    torch.manual_seed(3)
    images = torch.randn(batch, 6, frames, size, size).to(dev)
    noise = (torch.rand(images.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) - 0.5) * 2
    sh = torch.randperm(batch)
    default = ops.get_conv_math()
    res = {}
    try:
        for tag, mode, x in (('A', 'f32', images), ('B', 'fp16', images), ('C', 'f32', images * (1.0 + 2.0 ** -11 * noise))):
            ops.set_conv_math(mode)
            with pkg.MoCoTrainer(cfg, dev, use_graph=False, seed=1) as tr:
                out = tr.train_step(x, shuffle_ids=sh)
                a = tr.arena_q
                # fp16 storage leaves the arena gradients x the loss scale (the SGD kernel un-scales on the fly): out['grad_norm'][3]
                inv = 1.0 / float(out['grad_norm'][3]) if mode == 'fp16' else 1.0
                assert mode != 'fp16' or float(out['grad_norm'][2]) == 0.0           # no overflow, the step was applied
                res[tag] = dict(q=out['q'].clone(), loss=float(out['loss']),
                                grads={n: a.grad[o:o + s].clone() * inv for n, o, s in zip(a.names, a.offsets, a.sizes)})
    finally:
        ops.set_conv_math(default)

    def dist(tag):
        gA = res['A']['grads']
        e = sorted(rel_err(res[tag]['grads'][n], g) for n, g in gA.items() if float(g.abs().max()) > 0)
        return rel_err(res[tag]['q'], res['A']['q']), e[len(e) // 2]
    (qB, gB), (qC, gC) = dist('B'), dist('C')
    assert qC > 0 and gC > 0
    assert qB < 8 * qC and qB < 0.1, (qB, qC)
    assert gB < 4 * gC, (gB, gC)
    assert abs(res['B']['loss'] - res['A']['loss']) < 5e-2 * abs(res['A']['loss'])


def test_dynamic_loss_scale_kernels_skip_backoff_and_growth(pkg, ops):
    """gca_grad_unscale_clip + gca_sgd_step = what apex amp's scale_loss / patched optimizer.step() do in the reference
    (tools/train_video_contrast_dis.py:134-141,413-418), without a host sync: clean gradients are un-scaled (and clipped)
    inside the update; ONE inf or nan anywhere skips the update bit for bit, halves the scale and counts the skip;
    `interval` clean steps in a row double the scale."""
    torch.manual_seed(3)
    n = 256 * 1203
    S0 = 1024.0
    p0, g0 = torch.randn(n), torch.randn(n) * 0.01
    lr = torch.full((n // 256,), 0.06, device=DEV)
    wd = torch.full((n // 256,), 5e-4, device=DEV)
    saved = ops.LOSS_SCALE_INTERVAL
    try:
        ops.LOSS_SCALE_INTERVAL = 3
        for max_norm in (None, 0.5):
            state = ops.loss_scale_state(S0, torch.device(DEV))
            pr = torch.nn.Parameter(p0.clone())
            opt = torch.optim.SGD([pr], lr=0.06, momentum=0.9, weight_decay=5e-4)
            pd, buf = p0.to(DEV), torch.zeros(n, device=DEV)
            scale = S0
            for step, poison in enumerate((None, None, float('inf'), None, float('nan'), None, None, None, None)):
                gs = g0 * (step + 1)
                gd = (gs * scale).to(DEV)                       # what a backward pass seeded with S produces
                if poison is not None:
                    gd[n // 3] = poison
                before_p, before_b = pd.clone(), buf.clone()
                out = ops.grad_unscale_clip(gd, state, max_norm)
                ops.sgd_step(pd, gd, buf, lr, wd, 1.0, 0.9, False, out)
                o, st = out.cpu(), state.cpu()
                assert float(o[3]) == scale
                if poison is not None:
                    assert float(o[2]) == 1.0 and float(o[1]) == 0.0
                    assert torch.equal(pd, before_p) and torch.equal(buf, before_b)          # optimizer.step() skipped
                    scale = max(scale * 0.5, 1.0)
                    clean = 0
                else:
                    pr.grad = gs.clone()
                    if max_norm is not None:
                        torch.nn.utils.clip_grad_norm_([pr], max_norm)
                    opt.step()
                    assert float(o[2]) == 0.0
                    assert abs(float(o[0]) - float(gs.norm())) < 1e-4 * float(gs.norm())
                    assert rel_err(pd, pr.data) < 1e-5
                    clean = int(st[1])
                assert float(st[0]) in (scale, scale * 2.0)
                scale = float(st[0])
            st = state.cpu()
            assert float(st[2]) == 2.0 and float(st[3]) == 9.0                               # two skipped steps of nine
            # 2 clean, skip (S/2), 1 clean, skip (S/4), then 4 clean: one doubling after the third of them
            assert float(st[0]) == S0 / 4 * 2 and float(st[1]) == 1.0
    finally:
        ops.LOSS_SCALE_INTERVAL = saved


@pytest.mark.parametrize('kind', ['moco', 'simsiam'])
def test_trainer_skips_the_step_on_fp16_overflow_and_recovers(pkg, ops, kind, monkeypatch):
    """fp16 storage with a loss scale far too large (2^30: the first activation gradients overflow fp16's 65504): the
    trainer must NOT write inf / nan into the fp32 master weights, the momentum or the key encoder; it halves the scale per
    skipped step until the gradients fit, then trains.  hipGraph replay included (the decision lives on the device)."""
    import parity
    parity.register_tiny(pkg)
    monkeypatch.setenv('GCA_LOSS_SCALE', str(2.0 ** 30))
    default = ops.get_conv_math()
    ops.set_conv_math('fp16')
    try:
        dev = torch.device(DEV)
        torch.manual_seed(2)
        x = torch.randn(8, 6, 8, 48, 48, device=dev)
        if kind == 'moco':
            tr = pkg.MoCoTrainer(parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 64, 8), dev, use_graph=True, seed=4)
            arena = tr.arena_q
        else:
            tr = pkg.SimSiamTrainer(parity.make_cfg(pkg, 'R2P1D10T', 'simsiam', 32, 16, 8), dev, use_graph=True, seed=4)
            arena = tr.arena
        p0 = arena.flat.clone()
        skipped = applied = 0
        scales = []
        for step in range(40):
            before = arena.flat.clone()
            out = tr.train_step(x)
            rec = out['grad_norm'].cpu()
            scales.append(float(rec[3]))
            assert torch.isfinite(arena.flat).all() and torch.isfinite(tr.optimizer.buf).all(), step
            if kind == 'moco':
                assert torch.isfinite(tr.arena_k.flat).all(), step
            if float(rec[2]) != 0.0:
                skipped += 1
                assert torch.equal(arena.flat, before), step               # skipped: the query encoder is untouched
            else:
                applied += 1
                assert not torch.equal(arena.flat, before), step
            if applied >= 4:
                break
        st = tr.scale_state.cpu()
        assert skipped >= 5 and applied >= 4, (skipped, applied, scales)
        assert float(st[2]) == skipped and float(st[0]) == 2.0 ** 30 / 2.0 ** skipped, (st, skipped)
        assert torch.isfinite(out['loss']).all()
        assert tr._segments[0].graph is not None                           # the later steps were hipGraph replays
        tr.close()
    finally:
        ops.set_conv_math(default)
