"""GPU parity of every kernel family against (a) the committed golden vectors produced by the
reference's own modules and (b) the CPU oracle / ATen fp32 on seeded inputs, through the C ABI.
Tolerance: 1e-3 relative (max|a-b| / max|b|) as BASELINE.json's north_star states; most ops are
checked far tighter (1e-5) because fp32 MFMA is an exact fma chain."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
CONVS = ['stem_s', 'stem_t', 's3d_t_s2', 'c1x3x3', 'c1x3x3_s2', 'c3x1x1', 'c3x1x1_s2', 'c1x1x1_s2', 'c1x1x1',
         'c3x3x3', 'c3x3x3_s2', 'c7x7x7']


@pytest.fixture(scope='module')
def ops(pkg):
    return pkg.engine.ops


@pytest.fixture(params=['f32', 'bf16x6', 'bf16x3'])
def ctol(request, ops):
    """Runs a conv test under every arithmetic mode of the kernels (gca_set_conv_math) and hands it the bar that mode is
    held to: 1e-5 for fp32 MFMA and for bf16x6 (fp32-grade split products; measured 0.3-1e-6 rms against fp32 MFMA),
    5e-5 for bf16x3 (measured 4-6e-6) -- all far inside north_star's 1e-3."""
    default = ops.get_conv_math()
    ops.set_conv_math(request.param)
    yield 5e-5 if request.param == 'bf16x3' else 1e-5
    ops.set_conv_math(default)


def _conv_all(ops, x, w, dy, k, s, p):
    plan = ops.conv_plan(tuple(x.shape), w.shape[0], k, s, p, x.device)
    y, (ss, sq) = ops.conv_fwd(plan, x, ops.conv_pack(plan, 0, w), None, stats=True)
    dx = ops.conv_dgrad(plan, dy, ops.conv_pack(plan, 1, w))
    dw = torch.zeros_like(w)
    ops.conv_wgrad(plan, x, dy, dw, accumulate=True)
    return y, dx, dw, ss.sum(1), sq.sum(1)


@pytest.mark.parametrize('name', CONVS)
def test_conv_golden(ops, golden, ctol, name):
    g = golden('ops')
    cfg = g.t(name + ':cfg').tolist()
    x, w, dy = (g.t(name + ':' + t).to(DEV) for t in ('x', 'w', 'dy'))
    y, dx, dw, s1, s2 = _conv_all(ops, x, w, dy, cfg[0:3], cfg[3:6], cfg[6:9])
    yr = g.t(name + ':y')
    assert rel_err(y, yr) < ctol
    assert rel_err(dx, g.t(name + ':dx')) < ctol
    assert rel_err(dw, g.t(name + ':dw')) < ctol
    # fused BN statistics from the conv epilogue
    assert rel_err(s1, yr.sum((0, 2, 3, 4))) < 1e-4
    assert rel_err(s2, (yr * yr).sum((0, 2, 3, 4))) < 1e-4


@pytest.mark.parametrize('shape,K,k,s,p', [
    ((3, 5, 6, 17, 19), 70, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # ragged N tile, M tail
    ((2, 130, 4, 9, 9), 200, (3, 1, 1), (2, 1, 1), (1, 0, 0)),      # BM=128 path, K tail, temporal stride
    ((1, 64, 2, 33, 33), 144, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # strided dgrad (divide mode)
    ((4, 16, 1, 1, 1), 24, (1, 1, 1), (1, 1, 1), (0, 0, 0)),        # a Linear layer
    ((2, 7, 5, 6, 7), 9, (3, 3, 3), (1, 2, 1), (1, 0, 2)),          # mixed stride / asymmetric padding
])
def test_conv_random_vs_aten(ops, ctol, shape, K, k, s, p):
    torch.manual_seed(0)
    x = torch.randn(shape)
    w = torch.randn((K, shape[1]) + tuple(k)) * 0.1
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    y, dx, dw, _, _ = _conv_all(ops, x.to(DEV), w.to(DEV), dy.to(DEV), k, s, p)
    assert rel_err(y, yr) < ctol
    assert rel_err(dx, xr.grad) < ctol
    assert rel_err(dw, wr.grad) < ctol


@pytest.mark.parametrize('shape,K,k,s,p', [
    ((3, 40, 5, 8, 8), 150, (3, 1, 1), (1, 1, 1), (1, 0, 0)),       # pointwise in space: float4-gather variants
    ((2, 24, 6, 8, 8), 70, (3, 1, 1), (2, 1, 1), (1, 0, 0)),        # ... with temporal stride (dgrad classes)
    ((2, 20, 3, 12, 12), 100, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # spatial window, strided (4 dgrad classes)
    ((2, 33, 1, 1, 1), 170, (1, 1, 1), (1, 1, 1), (0, 0, 0)),       # Linear
])
def test_conv_every_launch_configuration(ops, ctol, shape, K, k, s, p):
    """Force every tile height (32..160), the 256-column float4 variant, and split-K factors through the
    tune_* fields: all must give the same convolution (the autotuner may pick any of them)."""
    torch.manual_seed(0)
    x = torch.randn(shape)
    w = torch.randn((K, shape[1]) + tuple(k)) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    plan = ops.ConvPlan(*shape, K, k, s, p, DEV)
    plan.tuned = [True, True, True]
    seen = set()
    for code in (32, 64, 96, 128, 160, 1024 + 32, 1024 + 64, 1024 + 128):
        for sp in (1, 3):
            plan.g.tune_fwd_bm = plan.g.tune_dgrad_bm = code
            plan.g.tune_fwd_splits = plan.g.tune_dgrad_splits = sp
            plan.g.tune_wgrad_splits = sp
            plan.refresh()
            seen.add((plan.cfg(0), plan.cfg(1)))
            y, (ss, sq) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None, stats=True)
            dx = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd))
            dw = torch.zeros_like(wd)
            ops.conv_wgrad(plan, xd, dyd, dw, accumulate=True)
            assert rel_err(y, yr) < ctol, (code, sp, plan.cfg(0))
            assert rel_err(dx, xr.grad) < ctol, (code, sp, plan.cfg(1))
            assert rel_err(dw, wr.grad) < ctol, (code, sp)
            assert rel_err(ss.sum(1), yr.detach().sum((0, 2, 3, 4))) < 1e-4
            assert rel_err(sq.sum(1), (yr.detach() ** 2).sum((0, 2, 3, 4))) < 1e-4
    assert len(seen) >= 8
    # two-phase launches: tall tiles over the first column tiles, short tiles over the rest (unit-stride passes only)
    for which, Ntot in ((0, yr.numel() // K), (1, x.numel() // shape[1])):
        tilesN = -(-Ntot // 128)
        if tilesN < 2 or (which == 1 and tuple(s) != (1, 1, 1)):
            continue
        for bm, tail in ((64, 1), (160, 2), (96, 1)):
            for mc in sorted({1, tilesN // 2, tilesN - 1}):
                if mc < 1:
                    continue
                plan.g.tune_fwd_bm = plan.g.tune_dgrad_bm = bm
                plan.g.tune_fwd_splits = plan.g.tune_dgrad_splits = 1
                plan.g.tune_fwd_tail = plan.g.tune_dgrad_tail = tail | (mc << 8)
                plan.refresh()
                if which == 0:
                    y, (ss, sq) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None, stats=True)
                    assert rel_err(y, yr) < ctol, (bm, tail, mc)
                    assert rel_err(ss.sum(1), yr.detach().sum((0, 2, 3, 4))) < 1e-4 and rel_err(sq.sum(1), (yr.detach() ** 2).sum((0, 2, 3, 4))) < 1e-4
                else:
                    assert rel_err(ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd)), xr.grad) < ctol, (bm, tail, mc)
    plan.g.tune_fwd_tail = plan.g.tune_dgrad_tail = 0
    # every wgrad tile shape (1..10; shapes not built for this tap count fall back to the heuristic one)
    seen_w = set()
    for idx in range(1, 11):
        for sp in (1, 5):
            plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = idx, sp
            plan.refresh()
            seen_w.add(plan.cfg(2)[:2])
            dw = torch.zeros_like(wd)
            ops.conv_wgrad(plan, xd, dyd, dw, accumulate=True)
            assert rel_err(dw, wr.grad) < ctol, (idx, sp, plan.cfg(2))
    assert len(seen_w) >= 6


def _boxes(bn, q, m, k):
    """Every power-of-two box of bn positions whose halo fits the LDS-halo kernels (<= 384 positions)."""
    out = []
    d = 1
    while d <= bn:
        h = 1
        while d * h <= bn:
            b = (d, h, bn // (d * h))
            P = 1
            for i in range(3):
                P *= (b[i] - 1) * m[i] + k[i]
            if P <= 384 and all(b[i] <= 4 * q[i] for i in range(3)):
                out.append(b)
            h *= 2
        d *= 2
    return out


@pytest.mark.parametrize('shape,K,k,s,p', [
    ((2, 40, 5, 12, 13), 70, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # spatial window; C = 2.5 chunks; ragged boxes
    ((2, 24, 6, 8, 8), 70, (3, 1, 1), (2, 1, 1), (1, 0, 0)),        # temporal window with temporal stride (2 dgrad classes)
    ((1, 32, 4, 9, 10), 33, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # 27 taps, all three axes padded
    ((2, 110, 9, 6, 6), 64, (7, 1, 1), (1, 1, 1), (3, 0, 0)),       # the R(2+1)D stem's temporal conv shape (110 -> 64)
    ((2, 20, 3, 12, 12), 100, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # strided spatial window: 4 dgrad classes
    ((3, 17, 2, 5, 5), 160, (1, 1, 1), (1, 1, 1), (0, 0, 0)),       # one tap: halo == box
])
def test_conv_halo_kernels_every_configuration(ops, ctol, shape, K, k, s, p):
    """conv3d_halo.hip (input window staged once per 16 channels in LDS, taps formed from LDS, pre-split packed weights):
    every tile height, 128- and 256-position boxes of every shape that fits, split-K, BatchNorm statistics, dgrad classes
    -- forced through tune_*_bm | 2048 and tune_*_box; all must reproduce ATen's convolution."""
    torch.manual_seed(5)
    x = torch.randn(shape)
    w = torch.randn((K, shape[1]) + tuple(k)) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    plan = ops.ConvPlan(*shape, K, k, s, p, DEV)
    plan.tuned = [True, True, True]
    ran = [0, 0]
    unit = tuple(s) == (1, 1, 1)
    qf, qd = tuple(yr.shape[2:]), tuple(shape[2:])
    kd_cls = tuple(-(-k[i] // s[i]) for i in range(3))              # taps per dgrad class (at most)
    for bn, rows_list in ((128, (32, 64, 96, 128, 160)), (256, (32, 64, 96, 128))):
        boxes_f = _boxes(bn, qf, s, k)
        boxes_d = _boxes(bn, qd if unit else tuple(-(-qd[i] // s[i]) for i in range(3)), (1, 1, 1), kd_cls)
        for bi, box in enumerate(sorted(set(boxes_f) | set(boxes_d))):
            code = box[0] | (box[1] << 8) | (box[2] << 16)
            for rows in (rows_list if bi % 3 == 0 else rows_list[bi % len(rows_list):][:1]):
                for sp in ((1, 2) if bi % 2 == 0 else (1,)):
                    plan.g.tune_fwd_bm = plan.g.tune_dgrad_bm = rows | 2048
                    plan.g.tune_fwd_box = plan.g.tune_dgrad_box = code
                    plan.g.tune_fwd_splits = plan.g.tune_dgrad_splits = sp
                    plan.refresh()
                    if (plan.cfg(0)[3] >> 14) & 1:
                        assert plan.cfg(0)[:2] == (rows, bn)
                        y, (ss, sq) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None, stats=True)
                        assert rel_err(y, yr) < ctol, ('fwd', rows, box, sp)
                        assert rel_err(ss.sum(1), yr.detach().sum((0, 2, 3, 4))) < 1e-4, ('sum', rows, box, sp)
                        assert rel_err(sq.sum(1), (yr.detach() ** 2).sum((0, 2, 3, 4))) < 1e-4, ('sq', rows, box, sp)
                        ran[0] += 1
                    if (plan.cfg(1)[3] >> 14) & 1:
                        dx = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd))
                        assert rel_err(dx, xr.grad) < ctol, ('dgrad', rows, box, sp)
                        acc = torch.ones_like(dx)
                        ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd), acc, accumulate=True)
                        assert rel_err(acc - 1, xr.grad) < 10 * ctol, ('dgrad+=', rows, box, sp)
                        ran[1] += 1
    # (a window strided in two dimensions has no box whose halo fits: such forward passes stay on the gather kernels)
    assert (ran[0] >= 8 or not _boxes(128, qf, s, k)) and ran[1] >= 8, ran
    # bias in the epilogue, and the two views of a (b, 6, T, H, W) batch read in place through the batch stride
    both = torch.randn((shape[0], 2 * shape[1]) + tuple(shape[2:]))
    bd = both.to(DEV)
    bias = torch.randn(K)
    for view in (0, 1):
        xv = torch.chunk(bd, 2, dim=1)[view]
        pv = ops.ConvPlan(*shape, K, k, s, p, DEV, x_batch_stride=xv.stride(0))
        pv.tuned = [True, True, True]
        if not _boxes(128, qf, s, k):
            break
        box = _boxes(128, qf, s, k)[0]
        pv.g.tune_fwd_bm, pv.g.tune_fwd_box = 64 | 2048, box[0] | (box[1] << 8) | (box[2] << 16)
        pv.refresh()
        assert (pv.cfg(0)[3] >> 14) & 1
        yv = ops.conv_fwd(pv, xv, ops.conv_pack(pv, 0, wd), bias.to(DEV))
        want = F.conv3d(torch.chunk(both, 2, dim=1)[view], w, bias, s, p)
        assert rel_err(yv, want) < ctol, ('view', view)


@pytest.mark.parametrize('shape,K,k,s,p,bias', [
    ((4, 3, 8, 56, 56), 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), False),     # R(2+1)D / S3D stem (resnet2p1d.py:162, s3d_1.py:35)
    ((4, 3, 8, 56, 56), 110, (1, 7, 7), (1, 2, 2), (0, 3, 3), False),    # R(2+1)D-18's 110 mid planes: the 128-row tile
    ((2, 3, 8, 48, 48), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3), False),     # 3D-ResNet stem (resnet.py:120)
    ((4, 3, 5, 37, 45), 20, (3, 5, 5), (2, 1, 2), (1, 2, 2), True),      # ragged boxes, 5 taps along W, unit stride in H, bias
    ((6, 4, 6, 30, 34), 70, (1, 3, 8), (1, 2, 2), (0, 1, 3), False),     # 4 channels, 8 W taps (no pad tap), two row tiles
    ((4, 1, 6, 33, 64), 33, (2, 2, 2), (1, 2, 2), (0, 0, 0), False),     # one channel, even kernel, no padding
])
def test_conv_stem_kernel_vs_gather_and_aten(ops, ctol, shape, K, k, s, p, bias):
    """conv3d_stem.hip (phase-split LDS halo, C <= 4, stride-2 W windows) against ATen and against the gather kernels on the
    same operands, BatchNorm partial sums included; the f32 mode has no stem kernel and must stay on the gather kernels."""
    torch.manual_seed(7)
    x = torch.randn(shape)
    w = torch.randn((K, shape[1]) + tuple(k)) * 0.1
    b = torch.randn(K) if bias else None
    yr = F.conv3d(x.double(), w.double(), None if b is None else b.double(), s, p)
    plan = ops.ConvPlan(*shape, K, k, s, p, torch.device(DEV))
    plan.tuned = [True, True, True]
    xd, wd = x.to(DEV), w.to(DEV)
    bd = None if b is None else b.to(DEV)
    outs = {}
    plan.g.tune_fwd_bm = 4096 | 64
    plan.refresh()
    runnable = (plan.cfg(0)[3] >> 16) & 1          # (the 7x7x7 halo of a 256-position box does not fit LDS in three bf16 parts)
    assert runnable == (0 if ops.get_conv_math() == 'f32' or (k == (7, 7, 7) and ops.get_conv_math() == 'bf16x6') else 1)
    for name, code in (('default', 0), ('stem', 4096 | 64), ('gather', 64)):
        plan.g.tune_fwd_bm = code
        plan.refresh()
        stem = (plan.cfg(0)[3] >> 16) & 1
        assert stem == (0 if name == 'gather' else runnable), (name, stem)      # the un-tuned default takes it whenever it can run
        y, (ss, sq) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), bd, stats=True)
        assert ss.shape[1] == plan.parts
        outs[name] = (y, ss.sum(1), sq.sum(1))
        assert rel_err(y, yr) < ctol, name
    plan.g.tune_fwd_bm = 0
    plan.refresh()
    y0 = F.conv3d(x.double(), w.double(), None, s, p)          # statistics are those of the conv output proper (no bias)
    for name in ('default', 'stem'):
        assert rel_err(outs[name][0], outs['gather'][0]) < ctol
        assert rel_err(outs[name][1], y0.sum((0, 2, 3, 4))) < 1e-4 * max(1.0, float(y0.abs().sum((0, 2, 3, 4)).max() / y0.sum((0, 2, 3, 4)).abs().max()))
        assert rel_err(outs[name][2], (y0 * y0).sum((0, 2, 3, 4))) < 1e-4


@pytest.mark.parametrize('shape,K,k,s,p', [
    ((2, 3, 2, 20, 20), 40, (1, 7, 7), (1, 2, 2), (0, 3, 3)),       # 49 taps: 64-bit tap mask (the R(2+1)D stem)
    ((1, 2, 9, 9, 9), 5, (7, 7, 7), (1, 2, 2), (3, 3, 3)),          # 343 taps: per-element window tests (3D-ResNet stem)
    ((2, 6, 5, 7, 7), 10, (3, 3, 3), (1, 1, 1), (1, 1, 1)),         # 27 taps, all three axes padded
])
def test_conv_wgrad_tap_mask_kinds(ops, ctol, shape, K, k, s, p):
    torch.manual_seed(3)
    x = torch.randn(shape)
    w = torch.randn((K, shape[1]) + tuple(k)) * 0.1
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(x, wr, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    plan = ops.ConvPlan(*shape, K, k, s, p, DEV)
    plan.tuned = [True, True, True]
    for idx in range(0, 11):
        plan.g.tune_wgrad_tile = idx
        plan.refresh()
        dw = torch.zeros_like(w).to(DEV)
        ops.conv_wgrad(plan, x.to(DEV), dy.to(DEV), dw, accumulate=True)
        assert rel_err(dw, wr.grad) < ctol, (idx, plan.cfg(2))


def test_conv_math_mode_is_a_floor_on_accuracy(ops):
    """tune_*_math = 1 + arithmetic overrides a pass only towards MORE accurate kernels (f32 > bf16x6 > bf16x3): in bf16x3
    mode a pass pinned to the fp32 MFMA kernel is fp32-accurate; in f32 mode a bf16x3 pin is ignored."""
    torch.manual_seed(1)
    shape, K, k, s, p = (2, 48, 4, 10, 10), 80, (3, 3, 3), (1, 1, 1), (1, 1, 1)
    x, w = torch.randn(shape, device=DEV), torch.randn((K, shape[1]) + k, device=DEV) * 0.05
    default = ops.get_conv_math()
    try:
        ops.set_conv_math('f32')
        plan = ops.ConvPlan(*shape, K, k, s, p, DEV)
        plan.tuned = [True, True, True]
        wp = ops.conv_pack(plan, 0, w)
        y_f32 = ops.conv_fwd(plan, x, wp).clone()
        dy = torch.randn_like(y_f32)
        dw_f32 = torch.zeros_like(w); ops.conv_wgrad(plan, x, dy, dw_f32, accumulate=True)
        plan.g.tune_fwd_math = plan.g.tune_wgrad_math = 2          # bf16x3 asked for, f32 in force: ignored
        plan.refresh()
        assert (plan.cfg(0)[3] >> 12) & 3 == 0 and (plan.cfg(2)[3] >> 12) & 3 == 0
        assert torch.equal(ops.conv_fwd(plan, x, wp), y_f32)
        ops.set_conv_math('bf16x3')
        plan.g.tune_fwd_math = plan.g.tune_wgrad_math = 0
        plan.refresh()
        assert (plan.cfg(0)[3] >> 12) & 3 == 1 and (plan.cfg(2)[3] >> 12) & 3 == 1
        e3 = rel_err(ops.conv_fwd(plan, x, wp), y_f32)
        assert 1e-7 < e3 < 5e-5
        plan.g.tune_fwd_math = plan.g.tune_wgrad_math = 1          # fp32 MFMA pinned inside bf16x3 mode
        plan.refresh()
        assert (plan.cfg(0)[3] >> 12) & 3 == 0 and (plan.cfg(2)[3] >> 12) & 3 == 0
        assert torch.equal(ops.conv_fwd(plan, x, wp), y_f32)
        dw = torch.zeros_like(w); ops.conv_wgrad(plan, x, dy, dw, accumulate=True)
        assert torch.equal(dw, dw_f32)
        plan.g.tune_fwd_math = 3                                    # bf16x6 inside bf16x3 mode
        plan.refresh()
        assert (plan.cfg(0)[3] >> 12) & 3 == 2
        assert rel_err(ops.conv_fwd(plan, x, wp), y_f32) < e3
    finally:
        ops.set_conv_math(default)


def test_conv_batch_stride_views(ops):
    """The two views of a (b,6,T,H,W) batch are read in place (tools/...dis.py:404)."""
    torch.manual_seed(1)
    img = torch.randn(3, 6, 4, 12, 12)
    w = torch.randn(10, 3, 1, 7, 7) * 0.1
    imgd = img.to(DEV)
    for half in (0, 1):
        xv = torch.chunk(imgd, 2, dim=1)[half]
        plan = ops.conv_plan(tuple(xv.shape), 10, (1, 7, 7), (1, 2, 2), (0, 3, 3), DEV, xv.stride(0))
        y = ops.conv_fwd(plan, xv, ops.conv_pack(plan, 0, w.to(DEV)))
        xr = torch.chunk(img, 2, dim=1)[half]
        yr = F.conv3d(xr, w, None, (1, 2, 2), (0, 3, 3))
        assert rel_err(y, yr) < 1e-5
        dy = torch.randn_like(yr)
        dw = torch.zeros_like(w).to(DEV)
        ops.conv_wgrad(plan, xv, dy.to(DEV), dw, accumulate=True)
        wr = w.clone().requires_grad_(True)
        F.conv3d(xr, wr, None, (1, 2, 2), (0, 3, 3)).backward(dy)
        assert rel_err(dw, wr.grad) < 1e-5


def test_conv_bias_and_accumulate(ops):
    torch.manual_seed(2)
    x, w, b = torch.randn(5, 12, 1, 1, 1), torch.randn(20, 12, 1, 1, 1), torch.randn(20)
    plan = ops.conv_plan(tuple(x.shape), 20, 1, 1, 0, DEV)
    y = ops.conv_fwd(plan, x.to(DEV), ops.conv_pack(plan, 0, w.to(DEV)), b.to(DEV))
    assert rel_err(y, F.conv3d(x, w, b)) < 1e-5
    dy = torch.randn(5, 20, 1, 1, 1)
    base = torch.randn_like(x)
    dx = base.clone().to(DEV)
    ops.conv_dgrad(plan, dy.to(DEV), ops.conv_pack(plan, 1, w.to(DEV)), dx, accumulate=True)
    ref = base + torch.einsum('nk,kc->nc', dy.flatten(1), w.flatten(1)).view_as(x)
    assert rel_err(dx, ref) < 1e-5
    db = torch.zeros(20, device=DEV)
    ops.bias_grad(dy.to(DEV), 5, 20, 1, db, True)
    assert rel_err(db, dy.sum((0, 2, 3, 4))) < 1e-5


@pytest.mark.parametrize('name', ['bn_s3d', 'bn_def'])
def test_bn_relu_golden(ops, golden, name):
    g = golden('ops')
    x = g.t(name + ':x').to(DEV)
    N, Cc = x.shape[:2]
    SP = x[0, 0].numel()
    eps, mom = [float(v) for v in g.t(name + ':hp')]
    gam, bet = g.t(name + ':g').to(DEV), g.t(name + ':b').to(DEV)
    rm, rv = g.t(name + ':rm0').to(DEV), g.t(name + ':rv0').to(DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    ss, sq = ops.bn_stats(x, N, Cc, SP)
    mean, invstd, scale, shift = ops.bn_finalize(ss, sq, N * SP, gam, bet, eps, mom, rm, rv, nbt)
    z = ops.bn_apply(x, scale, shift, None, True, N, Cc, SP)
    assert rel_err(z, g.t(name + ':y_relu')) < 1e-5
    assert rel_err(rm, g.t(name + ':rm1')) < 1e-5 and rel_err(rv, g.t(name + ':rv1')) < 1e-5
    assert int(nbt) == 1
    dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    dx = ops.bn_bwd(g.t(name + ':dy').to(DEV), z, x, gam, mean, invstd, True, N, Cc, SP, dg, db)
    assert rel_err(dx, g.t(name + ':dx')) < 1e-4
    assert rel_err(dg, g.t(name + ':dg')) < 1e-4 and rel_err(db, g.t(name + ':db')) < 1e-4
    # relu mode 2: the mask is recomputed from x with the forward's scale/shift instead of being read from z
    dg2, db2 = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    dx2 = ops.bn_bwd(g.t(name + ':dy').to(DEV), None, x, gam, mean, invstd, 2, N, Cc, SP, dg2, db2, None, False, scale, shift)
    assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)


def test_bn_residual_slice_and_odd_sizes(ops):
    """residual add, channel-slice output (concat buffer), SP not a multiple of 4, BatchNorm1d (SP=1)."""
    torch.manual_seed(3)
    for shape in [(3, 6, 2, 5, 7), (4, 10, 1, 1, 1), (2, 8, 2, 4, 4)]:
        x = torch.randn(shape) * 1.5 + 0.3
        res = torch.randn(shape)
        N, Cc = shape[:2]
        SP = x[0, 0].numel()
        bn = torch.nn.BatchNorm3d(Cc)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_()
        xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        zr = F.relu(bn(xr) + rr)
        dz = torch.randn_like(zr)
        zr.backward(dz)
        xd, resd = x.to(DEV), res.to(DEV)
        gam, bet = bn.weight.data.to(DEV), bn.bias.data.to(DEV)
        ss, sq = ops.bn_stats(xd, N, Cc, SP)
        mean, invstd, scale, shift = ops.bn_finalize(ss, sq, N * SP, gam, bet, bn.eps, 0.1, None, None, None)
        wide = torch.zeros((N, Cc + 5) + shape[2:], device=DEV)
        zs = wide[:, 3:3 + Cc]
        ops.bn_apply(xd, scale, shift, resd, True, N, Cc, SP, out=zs)
        assert rel_err(zs, zr) < 1e-5
        assert float(wide[:, :3].abs().max()) == 0 and float(wide[:, 3 + Cc:].abs().max()) == 0
        dwide = torch.zeros_like(wide)
        dwide[:, 3:3 + Cc] = dz.to(DEV)
        dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        dres = torch.empty_like(xd)
        dx = ops.bn_bwd(dwide[:, 3:3 + Cc], zs, xd, gam, mean, invstd, True, N, Cc, SP, dg, db, dres, False)
        assert rel_err(dx, xr.grad) < 1e-4
        assert rel_err(dres, rr.grad) < 1e-5
        assert rel_err(dg, bn.weight.grad) < 1e-4 and rel_err(db, bn.bias.grad) < 1e-4


@pytest.mark.parametrize('name', ['mp133', 'mp333s2', 'mp222', 'mp333s1', 'mp122'])
def test_maxpool_golden(ops, golden, name):
    g = golden('ops')
    cfg = g.t(name + ':cfg').tolist()
    x = g.t(name + ':x').to(DEV)
    plan = ops.pool_plan(tuple(x.shape), tuple(cfg[0:3]), tuple(cfg[3:6]), tuple(cfg[6:9]))
    y, am = ops.maxpool_fwd(plan, x)
    assert torch.equal(y.cpu(), g.t(name + ':y'))
    dx = ops.maxpool_bwd(plan, g.t(name + ':dy').to(DEV), am)
    assert rel_err(dx, g.t(name + ':dx')) < 1e-6        # ties (post-ReLU zeros) must break as ATen does


def test_weighted_avgpool(ops):
    torch.manual_seed(4)
    x = torch.randn(3, 7, 4, 5, 6)
    y = ops.wavgpool_fwd(x.to(DEV), None, 1.0 / (4 * 5 * 6))
    assert rel_err(y, x.mean((2, 3, 4))) < 1e-5
    # S3D tail: avg_pool3d((2,H,W), stride 1) then mean over T'  (s3d_1.py:30-33)
    ref = F.avg_pool3d(x, (2, 5, 6), stride=1).flatten(2).mean(2)
    wt = torch.tensor([1., 2., 2., 1.], device=DEV)
    y2 = ops.wavgpool_fwd(x.to(DEV), wt, 1.0 / (2 * 30 * 3))
    assert rel_err(y2, ref) < 1e-5
    dy = torch.randn(3, 7)
    dx = ops.wavgpool_bwd(dy.to(DEV), wt, 1.0 / (2 * 30 * 3), tuple(x.shape))
    xr = x.clone().requires_grad_(True)
    F.avg_pool3d(xr, (2, 5, 6), stride=1).flatten(2).mean(2).backward(dy)
    assert rel_err(dx, xr.grad) < 1e-5


def test_head_pieces(ops):
    torch.manual_seed(5)
    x = torch.randn(6, 40)
    xr = x.clone().requires_grad_(True)
    yr = F.normalize(xr, dim=1)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    y, inv = ops.l2norm_fwd(x.to(DEV))
    assert rel_err(y, yr) < 1e-5
    assert rel_err(ops.l2norm_bwd(dy.to(DEV), y, inv), xr.grad) < 1e-5
    r = ops.relu_fwd(x.to(DEV))
    assert torch.equal(r.cpu(), F.relu(x))
    assert torch.equal(ops.relu_bwd(dy.to(DEV), r).cpu(), dy * (x > 0))
    # SimSiam negative cosine
    p, z = torch.randn(5, 24), torch.randn(5, 24)
    pr = p.clone().requires_grad_(True)
    lr = -F.cosine_similarity(pr, z, dim=-1).mean() * 0.5
    lr.backward()
    buf = torch.zeros(6, device=DEV)
    dp = ops.negcos(p.to(DEV), z.to(DEV), 0.5, buf, False)
    assert rel_err(buf[:1], lr.detach().reshape(1)) < 1e-5 and rel_err(dp, pr.grad) < 1e-5


@pytest.mark.parametrize('tag,K,D,steps', [('k8', 8, 16, 5), ('k256', 256, 128, 3)])
def test_moco_queue_trace_golden(ops, golden, tag, K, D, steps):
    """RGBMoCo + NCESoftmaxLoss trace incl. ring wrap with n not dividing K (mem_moco.py / criterion.py)."""
    g = golden('moco')
    mem = g.t(tag + ':mem0').to(DEV)
    ptr = 0
    for s in range(steps):
        q, k = g.t('%s:q%d' % (tag, s)).to(DEV), g.t('%s:k%d' % (tag, s)).to(DEV)
        logits, lse, rank = ops.moco_logits_fwd(q, k, mem, 1 / 0.07, want_lse=True, want_rank=True)
        assert rel_err(logits, g.t('%s:logits%d' % (tag, s))) < 1e-5
        loss, _ = ops.nce_loss_fwd(logits, lse)
        assert rel_err(loss, g.t('%s:loss%d' % (tag, s)).reshape(1)) < 1e-5
        saved = ops.queue_enqueue(mem, k, ptr, save=True)
        # gradient against the PRE-enqueue snapshot although the rows are already overwritten
        dq = ops.moco_logits_bwd(k, mem, 1 / 0.07, logits=logits, lse=lse, ov_start=ptr, ov_rows=saved)
        assert rel_err(dq, g.t('%s:dq%d' % (tag, s))) < 1e-4
        dl = ops.nce_loss_bwd(logits, lse)
        dq2 = ops.moco_logits_bwd(k, mem, 1 / 0.07, dlogits=dl, ov_start=ptr, ov_rows=saved)
        assert rel_err(dq2, dq) < 1e-5
        ptr = (ptr + q.shape[0]) % K
        assert torch.equal(mem.cpu(), g.t('%s:mem%d' % (tag, s + 1)))
        assert ptr == int(g.t('%s:ptr%d' % (tag, s + 1)))
        prec1 = float((rank < 1).float().mean() * 100)
        assert abs(prec1 - float(g.t('%s:prec1_%d' % (tag, s)))) < 1e-4


def test_moco_allk_wrap_and_device_pointer(ops, golden):
    g = golden('moco')
    mem = g.t('allk:mem0').to(DEV)
    ptr_dev = torch.tensor([6], dtype=torch.long, device=DEV)
    lg, _, _ = ops.moco_logits_fwd(g.t('allk:q').to(DEV), g.t('allk:k').to(DEV), mem, 1 / 0.07)
    assert rel_err(lg, g.t('allk:logits')) < 1e-5
    ops.queue_enqueue(mem, g.t('allk:all_k').to(DEV), 0, ptr_dev=ptr_dev)
    ops.queue_advance(ptr_dev, 4, 8)
    assert torch.equal(mem.cpu(), g.t('allk:mem1')) and int(ptr_dev) == 2


@pytest.mark.parametrize('b,K,D', [(32, 4096, 128), (5, 1000, 128), (40, 65536, 128), (70, 16384, 128), (33, 9000, 64),
                                   (8, 300, 96), (16, 2048, 136)])
def test_infonce_sizes_vs_oracle(ops, b, K, D):
    """BASELINE configs' InfoNCE shapes (b=32 K=4096 / 65536) + ragged sizes, vs the oracle: every workgroup width
    of the fused kernel (1 / 2 / 4 / 8 waves), several batch tiles, narrower features, and D = 136 (two-pass path)."""
    from oracle.moco import RGBMoCo, NCESoftmaxLoss
    torch.manual_seed(6)
    mo = RGBMoCo(D, K=K, T=0.07)
    q = F.normalize(torch.randn(b, D)).requires_grad_(True)
    k = F.normalize(torch.randn(b, D))
    mem0 = mo.memory.clone()
    logits_r, _ = mo(q, k)
    loss_r = NCESoftmaxLoss()(logits_r)
    loss_r.backward()
    mem = mem0.to(DEV)
    logits, lse, rank = ops.moco_logits_fwd(q.detach().to(DEV), k.to(DEV), mem, 1 / 0.07, True, True)
    assert rel_err(logits, logits_r) < 1e-5
    loss, _ = ops.nce_loss_fwd(logits, lse)
    assert rel_err(loss, loss_r.detach().reshape(1)) < 1e-5
    l2, lse2, rank2, loss2 = ops.moco_logits_fwd(q.detach().to(DEV), k.to(DEV), mem, 1 / 0.07, True, True, want_loss=True)
    assert torch.equal(l2, logits) and torch.equal(lse2, lse) and torch.equal(rank2, rank)      # loss fused into the same launches
    assert rel_err(loss2, loss_r.detach().reshape(1)) < 1e-5
    dq = ops.moco_logits_bwd(k.to(DEV), mem, 1 / 0.07, logits=logits, lse=lse)
    assert rel_err(dq, q.grad) < 1e-4
    want_rank = (logits_r.detach()[:, 1:] >= logits_r.detach()[:, :1]).sum(1)
    assert torch.equal(rank.cpu().long(), want_rank)


def test_graph_block_golden(ops, golden):
    """sim adjacency / hop weighting / relaxed-Bernoulli sample / GCN aggregation (temporal_graph.py)."""
    g = golden('graph')
    w = g.group('aug:w:')
    x = g.t('aug:x').to(DEV)
    wq, wk, wg = (w[n].to(DEV) for n in ('g_q.0.weight', 'g_k.0.weight', 'gcns.0.conv.weight'))
    pl = ops.conv_plan(tuple(x.shape), wq.shape[0], 1, 1, 0, DEV)
    pp = ops.pool_plan(pl.out_shape, (1, 2, 2), (1, 2, 2), (0, 0, 0))
    gq, _ = ops.maxpool_fwd(pp, ops.conv_fwd(pl, x, ops.conv_pack(pl, 0, wq)))
    gk, _ = ops.maxpool_fwd(pp, ops.conv_fwd(pl, x, ops.conv_pack(pl, 0, wk)))
    sim, pre, adj = ops.graph_adj_fwd(gq, gk, g.t('aug:u').to(DEV), 3, 0.5, 1.0)
    assert rel_err(sim, g.t('aug:sim')) < 1e-4
    assert rel_err(pre, g.t('aug:pre')) < 1e-4
    assert rel_err(adj, g.t('aug:adj')) < 1e-4
    pg = ops.conv_plan(tuple(x.shape), wg.shape[0], 1, 1, 0, DEV)
    s = ops.conv_fwd(pg, x, ops.conv_pack(pg, 0, wg))
    out = ops.graph_gcn_fwd(g.t('aug:adj').to(DEV), s)
    assert rel_err(out, g.t('aug:y')) < 1e-5


def test_ema_sgd_multitensor(ops):
    torch.manual_seed(7)
    n = 256 * 37
    p, pe, gr = torch.randn(n), torch.randn(n), torch.randn(n)
    ped = pe.to(DEV)
    ops.ema_update(ped, p.to(DEV), 0.999)
    assert rel_err(ped, pe * 0.999 + p * (1 - 0.999)) < 1e-6
    # SGD with two (lr, wd) classes alternating per chunk, two steps (momentum buffer carry)
    lr = torch.where(torch.arange(n // 256) % 2 == 0, 0.06, 0.12).float()
    wd = torch.where(torch.arange(n // 256) % 2 == 0, 5e-4, 0.0).float()
    pr = p.clone()
    groups = [{'params': [torch.nn.Parameter(pr[i * 256:(i + 1) * 256].clone())], 'lr': float(lr[i]),
               'weight_decay': float(wd[i])} for i in range(n // 256)]
    opt = torch.optim.SGD(groups, momentum=0.9)
    pd, buf = p.to(DEV), torch.zeros(n, device=DEV)
    for step in range(2):
        gstep = gr * (step + 1)
        for i, gp in enumerate(groups):
            gp['params'][0].grad = gstep[i * 256:(i + 1) * 256].clone()
        opt.step()
        ops.sgd_step(pd, gstep.to(DEV), buf, lr.to(DEV), wd.to(DEV), 1.0, 0.9, False)
    want = torch.cat([gp['params'][0].data for gp in groups])
    assert rel_err(pd, want) < 1e-6


def test_gather_rows_contiguous_and_batch_strided_view(ops):
    """ShuffleBN gathers the key view of the (b,6,T,H,W) batch in place (tools/...dis.py:404,213-217)."""
    torch.manual_seed(5)
    img = torch.randn(6, 6, 2, 5, 7, device=DEV)
    idx = torch.tensor([4, 0, 5, 5, 2], device=DEV)
    x2 = torch.chunk(img, 2, dim=1)[1]
    assert torch.equal(ops.gather_rows(x2, idx), x2[idx])
    flat = torch.randn(9, 130, device=DEV)                      # float4 rows
    assert torch.equal(ops.gather_rows(flat, idx), flat[idx])
    odd = torch.randn(9, 3, device=DEV)                         # scalar rows
    assert torch.equal(ops.gather_rows(odd, idx), odd[idx])


@pytest.mark.parametrize('shape,k,s,p', [
    ((2, 5, 9, 14, 15), (3, 3, 3), (2, 2, 2), (1, 1, 1)),     # R(2+1)D / 3D-ResNet stem pool (W % 4 != 0: per-element backward)
    ((2, 5, 9, 14, 16), (3, 3, 3), (2, 2, 2), (1, 1, 1)),     # ... W % 4 == 0: paired-output forward, brick backward (odd D)
    ((1, 3, 8, 40, 36), (3, 3, 3), (2, 2, 2), (1, 1, 1)),     # ... many bricks per plane in every dimension
    ((2, 3, 7, 9, 12), (3, 3, 3), (2, 2, 2), (1, 1, 1)),      # ... odd D and odd H: half-empty bricks at both ends
    ((2, 3, 4, 13, 11), (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # S3D spatial pools
    ((2, 3, 5, 7, 9), (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # S3D Mixed_* branch pool (cover 3x3x3)
    ((2, 3, 6, 8, 10), (2, 2, 2), (2, 2, 2), (0, 0, 0)),      # non-overlapping
    ((1, 2, 7, 9, 12), (3, 1, 1), (2, 1, 1), (1, 0, 0)),      # S3D temporal pool -> run-time window path
    ((1, 2, 9, 11, 10), (5, 4, 2), (2, 3, 1), (2, 1, 1)),     # odd generic geometry
])
def test_maxpool_random_vs_aten_with_ties(ops, shape, k, s, p):
    """Values are quantised so windows hold many equal maxima: the argmax tie break (first in d,h,w order) and
    with it the gradient routing must equal ATen's (s3d_1.py / resnet2p1d.py MaxPool3d call sites)."""
    torch.manual_seed(7)
    x = (torch.randn(shape) * 2).round() / 2
    xr = x.clone().requires_grad_(True)
    yr, ir = F.max_pool3d(xr, k, s, p, return_indices=True)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    plan = ops.pool_plan(shape, k, s, p)
    y, am = ops.maxpool_fwd(plan, x.to(DEV))
    assert torch.equal(y.cpu(), yr.detach())
    assert torch.equal(am.cpu().long(), ir)
    dx = ops.maxpool_bwd(plan, dy.to(DEV), am)
    assert rel_err(dx, xr.grad) < 1e-6
    base = torch.randn(shape)
    dx2 = ops.maxpool_bwd(plan, dy.to(DEV), am, base.clone().to(DEV), True)
    assert rel_err(dx2, base + xr.grad) < 1e-6


@pytest.mark.parametrize('shape', [(1, 2, 5, 6, 8), (1, 2, 5, 6, 7)])
def test_maxpool_nan_propagates_like_aten(ops, shape):
    """ATen's rule is `val > max || isnan(val)`: a NaN in the window wins and later NaNs replace it.  Both stem-pool
    kernels (paired outputs when W % 4 == 0, per output otherwise) must route values and indices the same way."""
    torch.manual_seed(17)
    x = torch.randn(shape)
    x.view(-1)[torch.randperm(x.numel())[:x.numel() // 6]] = float('nan')
    yr, ir = F.max_pool3d(x, 3, 2, 1, return_indices=True)
    plan = ops.pool_plan(shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    y, am = ops.maxpool_fwd(plan, x.to(DEV))
    assert torch.equal(torch.isnan(y.cpu()), torch.isnan(yr))
    assert torch.equal(torch.nan_to_num(y.cpu(), nan=0.0), torch.nan_to_num(yr, nan=0.0))
    assert torch.equal(am.cpu().long(), ir)


def test_wgrad_splitk_reduce_vector_and_scalar_forms_give_the_same_bits(ops):
    """The split-K reduce reads the slabs with 16-byte loads when the weight count and the gradient pointer allow it and
    per element otherwise; per output the summation order is the same, so a gradient written at a 16-byte aligned address
    and one written four bytes further must be identical (and accumulate=True must add to what is there)."""
    torch.manual_seed(23)
    shape, K, k, s, p = (4, 24, 6, 14, 14), 40, (3, 1, 1), (1, 1, 1), (1, 0, 0)
    x, dy_shape = torch.randn(shape, device=DEV), None
    plan = ops.conv_plan(shape, K, k, s, p, DEV)
    dy = torch.randn(plan.out_shape, device=DEV)
    n = K * shape[1] * 3
    buf = torch.zeros(n + 8, device=DEV)
    assert buf.data_ptr() % 16 == 0 and n % 4 == 0
    a = buf[0:n].view(K, shape[1], 3, 1, 1)
    ops.conv_wgrad(plan, x, dy, a, False)
    a = a.clone()
    b = buf[1:n + 1].view(K, shape[1], 3, 1, 1)                   # 4 bytes off: the per-element kernel
    ops.conv_wgrad(plan, x, dy, b, False)
    assert torch.equal(a, b)
    base = torch.randn(n + 8, device=DEV)
    c0, c1 = base.clone(), base.clone()
    ops.conv_wgrad(plan, x, dy, c0[0:n].view(K, shape[1], 3, 1, 1), True)
    ops.conv_wgrad(plan, x, dy, c1[1:n + 1].view(K, shape[1], 3, 1, 1), True)
    assert rel_err(c0[0:n], base[0:n] + a.flatten()) < 1e-6 and rel_err(c1[1:n + 1], base[1:n + 1] + a.flatten()) < 1e-6
    ref = torch.nn.grad.conv3d_weight(x.cpu().double(), (K, shape[1]) + k, dy.cpu().double(), s, p).float()
    assert rel_err(a, ref) < 1e-3


def test_batched_weight_pack_equals_per_layer_pack(pkg, ops):
    """gca_conv_pack_batched (one launch per encoder) must write exactly what gca_conv_pack writes per layer."""
    L = pkg.engine.layers
    torch.manual_seed(8)
    net = torch.nn.ModuleList([L.HipConv3d(5, 70, (1, 3, 3), (1, 2, 2), (0, 1, 1)), L.HipConv3d(70, 33, (3, 1, 1), 1, (1, 0, 0)),
                               L.HipConv3d(33, 200, 1, (2, 2, 2), 0), L.HipLinear(40, 130)]).to(DEV)
    x = torch.randn(2, 5, 4, 12, 12, device=DEV)
    refs = []

    def ref_pack(m, plan, w):
        """Per-layer pack into a NaN-filled buffer: what stays NaN is reserved space no layout writes (a packed buffer has
        room for either weight layout of its class, conv_igemm_host.h pack_reserve)."""
        m.packed(plan, w).fill_(float('nan'))
        return m.packed(plan, w).clone()
    for m in list(net)[:3]:
        plan = m.plan(x)
        refs += [ref_pack(m, plan, 0), ref_pack(m, plan, 1)]
        x = ops.conv_fwd(plan, x, m._pack[0])
    lin = net[3]
    lp = lin.plan(6, DEV)
    refs += [ref_pack(lin, lp, 0), ref_pack(lin, lp, 1)]
    same = lambda a, b: bool(((a == b) | (a.isnan() & b.isnan())).all())
    packer = L.BatchedPacker(net, (0, 1))
    assert packer.n > 8               # 4 layers x 2 directions, and the strided (1,3,3) dgrad has 4 problem classes
    for m in net:
        for w in (0, 1):
            m._pack[w].fill_(float('nan'))
    packer.run()
    got = []
    for m in net:
        got += [m._pack[0], m._pack[1]]
    for a, b in zip(got, refs):
        assert same(a, b) and not bool(b.isnan().all())
    # layers skip their own packing while the batch is live, and pack again after release()
    with torch.no_grad():
        net[0].weight.mul_(2.0)
    assert same(net[0].packed(net[0]._pack_plan[0], 0), refs[0])
    packer.release()
    assert same(net[0].packed(net[0]._pack_plan[0], 0), refs[0] * 2.0)


def test_maxpool_with_fused_bn_relu_producer(ops):
    """gca_maxpool3d_fwd(scale, shift) == maxpool(bn_apply(x, relu)) bit for bit, argmax included
    (the stem of resnet2p1d.py:252-255 / resnet.py:176-179 without materialising the normalised tensor)."""
    torch.manual_seed(9)
    x = torch.randn(2, 6, 5, 10, 12, device=DEV)
    sc, sh = torch.randn(6, device=DEV), torch.randn(6, device=DEV)
    z = ops.bn_apply(x, sc, sh, None, True, 2, 6, 5 * 10 * 12)
    for k, s, p in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((2, 3, 1), (1, 2, 1), (1, 0, 0))):
        plan = ops.pool_plan(tuple(x.shape), k, s, p)
        y0, a0 = ops.maxpool_fwd(plan, z)
        y1, a1 = ops.maxpool_fwd(plan, x, True, sc, sh)
        assert torch.equal(y0, y1) and torch.equal(a0, a1)
    # at a size with many blocks: fused producer == pooling the materialised tensor == ATen, bit for bit
    x = torch.randn(2, 4, 12, 44, 52, device=DEV)
    sc, sh = torch.randn(4, device=DEV), torch.randn(4, device=DEV)
    z = ops.bn_apply(x, sc, sh, None, True, 2, 4, 12 * 44 * 52)
    plan = ops.pool_plan(tuple(x.shape), (3, 3, 3), (2, 2, 2), (1, 1, 1))
    y0, a0 = ops.maxpool_fwd(plan, z)
    y1, a1 = ops.maxpool_fwd(plan, x, True, sc, sh)
    yr, ir = F.max_pool3d(z.cpu(), 3, 2, 1, return_indices=True)
    assert torch.equal(y0, y1) and torch.equal(a0, a1) and torch.equal(y0.cpu(), yr) and torch.equal(a0.cpu().long(), ir)


def test_autotuner_pins_a_valid_configuration(ops):
    """ConvPlan.tune measures the candidate launch shapes of each pass on the real operands and pins one through
    gca_conv_geom.tune_* (the role cudnn.benchmark plays in the reference); results must not change."""
    torch.manual_seed(6)
    shape, K, k, s, p = (4, 24, 4, 14, 14), 72, (1, 3, 3), (1, 1, 1), (0, 1, 1)
    x = torch.randn(shape)
    w = torch.randn((K, shape[1]) + k) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    plan = ops.ConvPlan(*shape, K, k, s, p, DEV)
    plan.tuned = [False, False, False]                      # force the measurement (the suite default is heuristic)
    y = ops.conv_fwd(plan, x.to(DEV), ops.conv_pack(plan, 0, w.to(DEV)))
    dx = ops.conv_dgrad(plan, dy.to(DEV), ops.conv_pack(plan, 1, w.to(DEV)))
    dw = torch.zeros_like(w).to(DEV)
    ops.conv_wgrad(plan, x.to(DEV), dy.to(DEV), dw, accumulate=True)
    assert plan.tuned == [True, True, True]
    g = plan.g
    assert g.tune_fwd_bm and g.tune_dgrad_bm and g.tune_wgrad_tile and min(g.tune_fwd_splits, g.tune_dgrad_splits, g.tune_wgrad_splits) >= 1
    assert rel_err(y, yr) < 1e-5 and rel_err(dx, xr.grad) < 1e-5 and rel_err(dw, wr.grad) < 1e-5


@pytest.mark.parametrize('shape,K,k,s,p', [
    ((32, 3, 16, 112, 112), 110, (1, 7, 7), (1, 2, 2), (0, 3, 3)),     # R(2+1)D-18 stem, BASELINE configs[1] size
    ((32, 110, 16, 56, 56), 64, (7, 1, 1), (1, 1, 1), (3, 0, 0)),      # stem temporal conv (158 GFLOP)
    ((32, 64, 8, 28, 28), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # layer1 spatial conv
    ((32, 230, 8, 14, 14), 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),      # layer2 strided temporal conv
    ((32, 512, 1, 4, 4), 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # layer4 (split-K territory)
])
def test_conv_full_size_vs_device_reference_and_linearity(ops, ctol, shape, K, k, s, p):
    """BASELINE-size layers (too big for the CPU oracle in test time): all three passes against ATen's own fp32
    convolution on the same GPU (MIOpen; a different algorithm and summation order, hence 2e-4), plus a
    size-independent property of the product kernels alone: linearity in the input."""
    torch.manual_seed(11)
    x = torch.randn(shape, device=DEV)
    w = torch.randn((K, shape[1]) + tuple(k), device=DEV) * (1.0 / (shape[1] * k[0] * k[1] * k[2]) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    plan = ops.conv_plan(shape, K, k, s, p, DEV)
    wp0, wp1 = ops.conv_pack(plan, 0, w), ops.conv_pack(plan, 1, w)
    y, (ss, sq) = ops.conv_fwd(plan, x, wp0, None, stats=True)
    assert rel_err(y, yr.detach()) < 2e-4
    assert rel_err(ss.sum(1), yr.detach().sum((0, 2, 3, 4))) < 1e-3 and rel_err(sq.sum(1), (yr.detach() ** 2).sum((0, 2, 3, 4))) < 1e-3
    if shape[1] > 3:
        assert rel_err(ops.conv_dgrad(plan, dy, wp1), xr.grad) < 2e-4
    dw = torch.zeros_like(w)
    ops.conv_wgrad(plan, x, dy, dw, accumulate=True)
    assert rel_err(dw, wr.grad) < 2e-4
    x2 = torch.randn(shape, device=DEV)
    lin = ops.conv_fwd(plan, 0.5 * x - 2.0 * x2, wp0)
    assert rel_err(lin, 0.5 * y - 2.0 * ops.conv_fwd(plan, x2, wp0)) < 2 * ctol


@pytest.mark.parametrize('N,C,SP,res', [(4, 10, 48, True), (3, 7, 30, False), (2, 5, 20000, True), (8, 6, 1, False)])
def test_bn_train_fwd_and_small_bwd_equal_the_separate_passes(ops, N, C, SP, res):
    """gca_bn_train_fwd (finalize + apply, ONE launch when N*SP is small) and the one-launch small BatchNorm backward
    against the separate reduce / finalize / apply kernels (N*SP = 40000 takes those), same inputs."""
    torch.manual_seed(12)
    x = torch.randn(N, C, SP, device=DEV) * 2 + 0.3
    r = torch.randn(N, C, SP, device=DEV) if res else None
    gam, bet = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    ss, sq = ops.bn_stats(x, N, C, SP)

    def fresh():
        return torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
    rm0, rv0, nb0 = fresh()
    mean, invstd, scale, shift = ops.bn_finalize(ss, sq, N * SP, gam, bet, 1e-5, 0.1, rm0, rv0, nb0)
    z0 = ops.bn_apply(x, scale, shift, r, True, N, C, SP)
    rm1, rv1, nb1 = fresh()
    z1, mean1, invstd1, scale1, shift1 = ops.bn_train_fwd(ss, sq, N * SP, gam, bet, 1e-5, 0.1, rm1, rv1, nb1, x, r, True, N, C, SP)
    assert torch.equal(z0, z1) and torch.equal(mean, mean1) and torch.equal(invstd, invstd1) and torch.equal(scale, scale1)
    assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1) and int(nb1) == 1
    # backward against an autograd reference of the same op
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    zr = F.relu(y + rr if res else y)
    dz = torch.randn_like(zr)
    zr.backward(dz)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dres = torch.empty_like(x) if res else None
    dx = ops.bn_bwd(dz, z1, x, gam, mean, invstd, 1, N, C, SP, dg, db, dres, False)
    assert rel_err(dx, xr.grad) < 1e-4 and rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    if res:
        assert rel_err(dres, rr.grad) < 1e-6
    else:      # no residual: the mask can come from x instead of z
        dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx2 = ops.bn_bwd(dz, None, x, gam, mean, invstd, 2, N, C, SP, dg2, db2, None, False, scale, shift)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg)


@pytest.mark.parametrize('shape,K,k,p,res,halo', [
    ((4, 96, 2, 7, 7), 80, (3, 1, 1), (1, 0, 0), True, False),        # gather kernel, residual, K tile tail
    ((3, 64, 2, 6, 6), 48, (1, 3, 3), (0, 1, 1), False, True),        # LDS-halo kernel
    ((2, 256, 1, 4, 4), 130, (1, 1, 1), (0, 0, 0), False, False),     # pointwise, 32 positions
])
def test_conv_splitk_slabs_folded_by_the_batchnorm_kernel(ops, shape, K, k, p, res, halo):
    """gca_conv_fwd_slabs + gca_bn_train_fwd_slabs (a split-K conv in front of a small BatchNorm: the BatchNorm kernel folds
    the slabs) against gca_conv_fwd (which finishes its slabs itself) + gca_bn_train_fwd: the same bits for y, the
    statistics and z to fp32 rounding (they skip the fp32 partial sums here), running statistics updated once."""
    torch.manual_seed(K)
    N, C, D, Hh, W = shape
    x = torch.randn(shape, device=DEV)
    w = torch.randn((K, C) + k, device=DEV) * 0.1
    gam, bet = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV)
    plan = ops.ConvPlan(N, C, D, Hh, W, K, k, 1, p, DEV)
    plan.tuned = [True, True, True]
    for sp in (2, 3, 5):
        plan.g.tune_fwd_bm, plan.g.tune_fwd_splits = (2048 + 64 if halo else 64), sp
        plan.refresh()
        if plan.cfg(0)[2] < 2:
            continue
        wp = ops.conv_pack(plan, 0, w)
        SP = plan.out_shape[2] * plan.out_shape[3] * plan.out_shape[4]
        r = torch.randn(plan.out_shape, device=DEV) if res else None

        def fresh():
            return torch.zeros(K, device=DEV), torch.ones(K, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
        rm0, rv0, nb0 = fresh()
        y0, (ss, sq) = ops.conv_fwd(plan, x, wp, None, stats=True)
        z0, mean0, invstd0, scale0, shift0 = ops.bn_train_fwd(ss, sq, N * SP, gam, bet, 1e-5, 0.1, rm0, rv0, nb0, y0, r, True, N, K, SP)
        rm1, rv1, nb1 = fresh()
        y1, z1, mean1, invstd1, scale1, shift1 = ops.conv_bn_small_fwd(plan, x, wp, N * SP, gam, bet, 1e-5, 0.1, rm1, rv1, nb1, r, True)
        assert torch.equal(y0, y1), (sp, plan.cfg(0))
        assert rel_err(mean1, mean0) < 1e-6 and rel_err(invstd1, invstd0) < 1e-6 and rel_err(z1, z0) < 2e-6
        assert rel_err(rm1, rm0) < 1e-6 and rel_err(rv1, rv0) < 1e-6 and int(nb1) == 1
        # a channel slice of a wider buffer as the destination (Inception concat)
        wide = torch.zeros((N, K + 5) + tuple(plan.out_shape[2:]), device=DEV)
        rm2, rv2, nb2 = fresh()
        ops.conv_bn_small_fwd(plan, x, wp, N * SP, gam, bet, 1e-5, 0.1, rm2, rv2, nb2, r, True, out=wide[:, 3:3 + K])
        assert torch.equal(wide[:, 3:3 + K], z1) and float(wide[:, :3].abs().max()) == 0.0 and float(wide[:, 3 + K:].abs().max()) == 0.0
        return
    pytest.fail('no launch shape with a split reduction was available')


@pytest.mark.parametrize('shape,K,kd,pd', [
    ((2, 40, 9, 4, 8), 48, 7, 3),       # stem-temporal kind: C, K not multiples of 32 (tile tails), D > kd, HW = 32
    ((3, 33, 5, 4, 4), 70, 7, 3),       # D < kd: most taps of most planes lie in the padding; HW = 16; odd unit count
    ((2, 64, 1, 4, 4), 32, 7, 3),       # D = 1: only the centre tap ever meets data
    ((2, 50, 8, 4, 12), 64, 3, 1),      # residual-block kind, HW = 48
    ((5, 70, 4, 8, 8), 96, 3, 1),
    ((2, 32, 6, 4, 4), 40, 3, 0),       # "valid" temporal conv: OD = D - 2
    ((2, 32, 3, 4, 4), 40, 3, 2),       # over-padded: OD = D + 2
])
def test_conv_wgrad_streaming_temporal_kernel(ops, shape, K, kd, pd):
    """conv3d_wgrad_ts.hip (tune_wgrad_tile 11 / 12: the register-window weight gradient of (kd,1,1) unit-stride convs) forced
    over both wave tiles, split counts from one workgroup to one unit per wave, both split-product arithmetics, += into a
    live buffer and a batch-strided x view -- against ATen's weight gradient in fp64.  fp32-MFMA mode must refuse it."""
    torch.manual_seed(kd * 100 + shape[1])
    N, C, D, Hh, W = shape
    big = torch.randn(N, 2 * C, D, Hh, W)
    x = big[:, C:]                                           # batch-strided view (the second view of a clip pair)
    w = torch.randn(K, C, kd, 1, 1) * 0.1
    xr, wr = x.double().clone().requires_grad_(True), w.double().clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, 1, (pd, 0, 0))
    dy = torch.randn(yr.shape)
    yr.backward(dy.double())
    units = N * (Hh * W // 16)
    default = ops.get_conv_math()
    bigd, dyd = big.to(DEV), dy.to(DEV)
    xd = bigd[:, C:]
    try:
        for mode, tol in (('bf16x6', 1e-5), ('bf16x3', 5e-5)):
            ops.set_conv_math(mode)
            for view in (False, True):
                xin = xd if view else xd.contiguous()
                plan = ops.ConvPlan(N, C, D, Hh, W, K, (kd, 1, 1), 1, (pd, 0, 0), DEV, x_batch_stride=xin.stride(0) if view else 0)
                plan.tuned = [True, True, True]
                for tile in (11, 12):
                    for sp in sorted({1, 2, max(1, units // 8), max(1, units // 4)}):
                        plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = tile, sp
                        plan.refresh()
                        cfg = plan.cfg(2)
                        assert cfg[3] & 255 == tile, (tile, cfg)                       # really the streaming kernel
                        dw = torch.full_like(w, 0.5).to(DEV)
                        ops.conv_wgrad(plan, xin, dyd, dw, accumulate=True)
                        assert rel_err(dw - 0.5, wr.grad) < tol, (mode, view, tile, sp, cfg)
                        ops.conv_wgrad(plan, xin, dyd, dw, accumulate=False)
                        assert rel_err(dw, wr.grad) < tol, (mode, view, tile, sp, cfg)
        ops.set_conv_math('f32')
        plan = ops.ConvPlan(N, C, D, Hh, W, K, (kd, 1, 1), 1, (pd, 0, 0), DEV)
        plan.tuned = [True, True, True]
        plan.g.tune_wgrad_tile = 11
        plan.refresh()
        assert plan.cfg(2)[3] & 255 != 11                                              # falls back to a conv_wgrad_kernel shape
        dw = torch.zeros_like(w).to(DEV)
        ops.conv_wgrad(plan, xd.contiguous(), dyd, dw, accumulate=True)
        assert rel_err(dw, wr.grad) < 1e-5
    finally:
        ops.set_conv_math(default)


@pytest.mark.parametrize('shape,K', [
    ((2, 40, 3, 8, 28), 48),        # W = 28: two chunks per row, the second one ragged; C, K tile tails
    ((3, 33, 2, 5, 16), 70),        # one full chunk per row, odd unit count
    ((2, 64, 1, 9, 4), 32),         # W = 4: one quad per row
    ((1, 20, 2, 2, 56), 40),        # H = 2 (every row touches the padding), four chunks
    ((4, 32, 4, 28, 28), 144),      # the layer-1 kind of R(2+1)D-18 at small batch
    ((3, 40, 4, 14, 14), 96),       # W = 14: one chunk per row, tail columns masked in registers (layer-2 kind)
    ((2, 33, 2, 7, 7), 64),         # W = 7
    ((2, 32, 2, 6, 13), 32),        # odd W
])
def test_conv_wgrad_streaming_spatial_kernel(ops, shape, K):
    """conv3d_wgrad_ts.hip, tune_wgrad_tile 13: the streaming weight gradient of (1,3,3) / pad 1 / unit-stride convs (three-row
    register window, column shifts cut out of one wide fragment read, zero padding by the DMA's range check) over split counts
    from one workgroup to one unit per wave, both split-product arithmetics, += and a batch-strided x view, vs ATen in fp64."""
    torch.manual_seed(shape[1] * 7 + shape[4])
    N, C, D, Hh, W = shape
    big = torch.randn(N, 2 * C, D, Hh, W)
    x = big[:, C:]
    w = torch.randn(K, C, 1, 3, 3) * 0.1
    xr, wr = x.double().clone().requires_grad_(True), w.double().clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, 1, (0, 1, 1))
    dy = torch.randn(yr.shape)
    yr.backward(dy.double())
    units = N * D * -(-W // 16)
    default = ops.get_conv_math()
    bigd, dyd = big.to(DEV), dy.to(DEV)
    xd = bigd[:, C:]
    try:
        for mode, tol in (('bf16x6', 1e-5), ('bf16x3', 5e-5)):
            ops.set_conv_math(mode)
            for view in (False, True):
                xin = xd if view else xd.contiguous()
                plan = ops.ConvPlan(N, C, D, Hh, W, K, (1, 3, 3), 1, (0, 1, 1), DEV, x_batch_stride=xin.stride(0) if view else 0)
                plan.tuned = [True, True, True]
                for sp in sorted({1, 2, max(1, units // 8), max(1, units // 4)}):
                    plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = 13, sp
                    plan.refresh()
                    cfg = plan.cfg(2)
                    assert cfg[3] & 255 == 13, cfg
                    dw = torch.full_like(w, 0.5).to(DEV)
                    ops.conv_wgrad(plan, xin, dyd, dw, accumulate=True)
                    assert rel_err(dw - 0.5, wr.grad) < tol, (mode, view, sp, cfg)
                    ops.conv_wgrad(plan, xin, dyd, dw, accumulate=False)
                    assert rel_err(dw, wr.grad) < tol, (mode, view, sp, cfg)
        # geometries it must refuse: W % 4 != 0 beyond one chunk, strides
        for shp, s in (((2, 16, 2, 6, 18), 1), ((2, 16, 2, 8, 28), (1, 2, 2))):
            plan = ops.ConvPlan(*shp, 32, (1, 3, 3), s, (0, 1, 1), DEV)
            plan.g.tune_wgrad_tile = 13
            plan.refresh()
            assert plan.cfg(2)[3] & 255 != 13
    finally:
        ops.set_conv_math(default)


@pytest.mark.parametrize('shape,K,k,p', [
    ((3, 3, 4, 28, 48), 110, (1, 7, 7), (0, 3, 3)),     # R(2+1)D-18 stem kind: four row tiles, OW = 24 (second step half valid)
    ((2, 3, 5, 36, 64), 64, (1, 7, 7), (0, 3, 3)),      # S3D stem kind: two row tiles, the two waves of a tile set share the steps
    ((2, 3, 6, 20, 32), 64, (7, 7, 7), (3, 3, 3)),      # 3D-ResNet stem kind: two tap planes per workgroup, planes in the padding
    ((2, 3, 4, 22, 48), 20, (3, 5, 7), (1, 2, 3)),      # K < 32, 5 x 7 taps, odd OH
    ((2, 1, 3, 16, 32), 40, (3, 7, 5), (1, 3, 2)),      # one input channel, 5 taps along W
    ((2, 4, 2, 18, 32), 70, (1, 5, 7), (0, 2, 3)),      # four input channels (140 columns), K tile tail
    ((1, 3, 3, 40, 16), 33, (1, 7, 7), (0, 3, 3)),      # OW = 8: half a step per row
])
def test_conv_wgrad_stem_kernel(ops, shape, K, k, p):
    """conv3d_wgrad_stem.hip (tune_wgrad_tile 14): the weight gradient of the <= 4-channel, stride-(1,2,2) stem convs (parity
    phases of every input row staged once, rows rolling over the output rows, dY by LDS-DMA as fragments) over split counts
    from one workgroup per tap-plane group to chunks of 8 output rows, all three arithmetics it serves (bf16x6, bf16x3, fp16
    storage), += and a batch-strided x view, vs ATen in fp64."""
    torch.manual_seed(shape[3] * 5 + K)
    N, C, D, Hh, W = shape
    big = torch.randn(N, 2 * C, D, Hh, W)
    x = big[:, C:]
    w = torch.randn((K, C) + k) * 0.1
    xr, wr = x.double().clone().requires_grad_(True), w.double().clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, (1, 2, 2), p)
    dy = torch.randn(yr.shape)
    yr.backward(dy.double())
    units = N * yr.shape[2]
    default = ops.get_conv_math()
    try:
        for mode, tol in (('bf16x6', 1e-5), ('bf16x3', 5e-5), ('fp16', 2e-3)):
            ops.set_conv_math(mode)
            f16 = mode == 'fp16'
            bigd = big.to(DEV).half() if f16 else big.to(DEV)
            dyd = dy.to(DEV).half() if f16 else dy.to(DEV)
            ref = wr.grad
            if f16:                                     # the fp16 path differentiates the ROUNDED tensors
                ref = torch.nn.grad.conv3d_weight(bigd[:, C:].double().cpu(), w.shape, dyd.double().cpu(), stride=(1, 2, 2), padding=p)
                tol = 1e-5
            xd = bigd[:, C:]
            for view in ((False,) if f16 else (False, True)):
                xin = xd if view else xd.contiguous()
                plan = ops.ConvPlan(N, C, D, Hh, W, K, k, (1, 2, 2), p, DEV, x_batch_stride=xin.stride(0) if view else 0, act_f16=f16)
                plan.tuned = [True, True, True]
                for sp in sorted({1, 2, units, 4 * units}):
                    plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = 14, sp
                    plan.refresh()
                    cfg = plan.cfg(2)
                    assert cfg[3] & 255 == 14, cfg
                    dw = torch.full_like(w, 0.5).to(DEV)
                    ops.conv_wgrad(plan, xin, dyd, dw, accumulate=True)
                    assert rel_err(dw - 0.5, ref) < tol, (mode, view, sp, cfg)
                    ops.conv_wgrad(plan, xin, dyd, dw, accumulate=False)
                    assert rel_err(dw, ref) < tol, (mode, view, sp, cfg)
        # geometries it must refuse: unit stride, > 4 input channels, 9 taps along W
        ops.set_conv_math('bf16x6')
        for shp, kk, s, pp in (((2, 3, 2, 16, 32), (1, 7, 7), 1, (0, 3, 3)), ((2, 8, 2, 16, 32), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                               ((2, 3, 2, 16, 32), (1, 3, 9), (1, 2, 2), (0, 1, 4))):
            plan = ops.ConvPlan(*shp, 32, kk, s, pp, DEV)
            plan.g.tune_wgrad_tile = 14
            plan.refresh()
            assert plan.cfg(2)[3] & 255 != 14
    finally:
        ops.set_conv_math(default)


@pytest.mark.parametrize('shape,K,kd,sd,pd', [
    ((4, 48, 1, 4, 4), 40, 3, 1, 1),        # D = 1: only the centre tap meets data (layer4 of R(2+1)D-18)
    ((3, 32, 2, 5, 5), 24, 3, 2, 1),        # D = 2, temporal stride 2 -> OD = 1: taps 1, 2
    ((2, 32, 2, 4, 4), 32, 7, 1, 3),        # D = 2 under a 7-tap window: taps 2..4
    ((2, 32, 3, 4, 4), 32, 3, 1, 1),        # nothing to drop
    ((2, 32, 1, 4, 4), 32, 3, 1, 0 + 1),
])
def test_temporal_convs_drop_taps_that_only_meet_padding(ops, ctol, shape, K, kd, sd, pd):
    """Forward / dgrad classes of (kd,1,1) convs leave out the taps that multiply zero padding for every output position
    (conv_igemm_host.h build_classes); results against ATen, packed size shrinks accordingly."""
    torch.manual_seed(kd + shape[2])
    x = torch.randn(shape)
    w = torch.randn(K, shape[1], kd, 1, 1) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, (sd, 1, 1), (pd, 0, 0))
    dy = torch.randn_like(yr)
    yr.backward(dy)
    plan = ops.ConvPlan(*shape, K, (kd, 1, 1), (sd, 1, 1), (pd, 0, 0), DEV)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    y, (ss, sq) = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None, stats=True)
    dx = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd))
    dw = torch.zeros_like(wd)
    ops.conv_wgrad(plan, xd, dyd, dw, accumulate=True)
    assert rel_err(y, yr) < ctol and rel_err(dx, xr.grad) < ctol and rel_err(dw, wr.grad) < ctol
    assert rel_err(ss.sum(1), yr.detach().sum((0, 2, 3, 4))) < 1e-4


@pytest.mark.parametrize('shape,K,kd,pd', [
    ((3, 40, 6, 4, 8), 48, 3, 1),        # residual-block kind; C = 2.5 chunks of 16 (channel tail), K tile tail
    ((2, 110, 9, 4, 4), 64, 7, 3),       # the stem's temporal conv kind
])
def test_conv_consumes_producer_batchnorm_relu_on_the_fly(ops, shape, K, kd, pd):
    """gca_conv_fwd_xf / gca_conv_wgrad_xf: a conv that reads its producer's PRE-activation tensor y and applies
    relu(y * scale[c] + shift[c]) where it stages its input (LDS-halo forward, streaming weight gradient) against the same conv
    on the materialised tensor z (gca_bn_apply) -- same kernels, same products: bit for bit -- and against ATen on z.  Negative
    shifts make whole regions of z zero; the conv's zero padding must pad z (not relu(shift))."""
    torch.manual_seed(31)
    N, C, D, Hh, W = shape
    y_in = torch.randn(shape, device=DEV)
    gam = torch.rand(C, device=DEV) + 0.5
    bet = torch.randn(C, device=DEV) + 0.7                         # mostly positive shifts: relu(shift) != 0 in the padding
    ss, sq = ops.bn_stats(y_in, N, C, D * Hh * W)
    rm, rv, nb = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
    mean, invstd, scale, shift = ops.bn_finalize(ss, sq, N * D * Hh * W, gam, bet, 1e-5, 0.1, rm, rv, nb)
    z = ops.bn_apply(y_in, scale, shift, None, True, N, C, D * Hh * W)
    w = torch.randn(K, C, kd, 1, 1, device=DEV) * 0.1
    zr, wr = z.detach().cpu().double().requires_grad_(True), w.cpu().double().requires_grad_(True)
    outr = F.conv3d(zr, wr, None, 1, (pd, 0, 0))
    dy = torch.randn(outr.shape)
    outr.backward(dy.double())
    default = ops.get_conv_math()
    try:
        for mode, tol in (('bf16x6', 1e-5), ('bf16x3', 5e-5)):
            ops.set_conv_math(mode)
            plan = ops.ConvPlan(N, C, D, Hh, W, K, (kd, 1, 1), 1, (pd, 0, 0), DEV)
            plan.tuned = [True, True, True]
            assert not ops.conv_xf_ok(plan)                                     # heuristic shapes: gather kernels / conv_wgrad_kernel
            # a halo box that fits: bd x bh x bw = 128 positions
            box = None
            for bd in (1, 2, 4, 8):
                for bh in (1, 2, 4, 8):
                    bw = 128 // (bd * bh)
                    if bw >= 1 and bd * bh * bw == 128 and bd <= 2 * D and bh <= 2 * Hh and bw <= 2 * W and (bd + kd - 1) * bh * bw <= 384:
                        box = box or (bd, bh, bw)
            plan.g.tune_fwd_bm, plan.g.tune_fwd_box = 64 | 2048, box[0] | (box[1] << 8) | (box[2] << 16)
            plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = 11, 2
            plan.refresh()
            assert (plan.cfg(0)[3] >> 14) & 1 and plan.cfg(2)[3] & 255 == 11
            assert ops.conv_xf_ok(plan)
            wp = ops.conv_pack(plan, 0, w)
            o1, (s1, q1) = ops.conv_fwd(plan, z, wp, None, stats=True)
            o2, (s2, q2) = ops.conv_fwd_xf(plan, y_in, scale, shift, wp, stats=True)
            assert torch.equal(o1, o2) and torch.equal(s1, s2) and torch.equal(q1, q2)
            assert rel_err(o2, outr) < tol
            dyd = dy.to(DEV)
            g1, g2 = torch.zeros_like(w), torch.zeros_like(w)
            ops.conv_wgrad(plan, z, dyd, g1, accumulate=True)
            ops.conv_wgrad(plan, y_in, dyd, g2, accumulate=True, xf=(scale, shift))
            assert torch.equal(g1, g2)
            assert rel_err(g2, wr.grad) < tol
            # through the collector (one batched reduce) as the trainers run it
            d = ops.DeferredReduce()
            g3 = torch.zeros_like(w)
            ops.DEFER[0] = d
            try:
                ops.conv_wgrad(plan, y_in, dyd, g3, accumulate=True, xf=(scale, shift))
                d.flush()
            finally:
                ops.DEFER[0] = None
            assert torch.equal(g3, g2)
    finally:
        ops.set_conv_math(default)
