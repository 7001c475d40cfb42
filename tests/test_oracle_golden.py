"""Pins the CPU oracle (oracle/) to the reference: every fixture under tests/golden/ was
produced by running the reference's own classes (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import encoders as oenc, graph as ograph, moco as omoco, wrappers as owrap
from conftest import rel_err

TOL = 2e-5   # oracle and reference run the same ATen kernels; only op order may differ


def test_conv_bn_pool_vectors_are_aten(golden):
    """The per-op fixtures are plain ATen calls -- re-derive them to make sure the files are sane."""
    g = golden('ops')
    for name in ['stem_s', 'stem_t', 's3d_t_s2', 'c1x3x3', 'c1x3x3_s2', 'c3x1x1', 'c3x1x1_s2', 'c1x1x1_s2',
                 'c1x1x1', 'c3x3x3', 'c3x3x3_s2', 'c7x7x7']:
        cfg = g.t(name + ':cfg').tolist()
        y = F.conv3d(g.t(name + ':x'), g.t(name + ':w'), None, cfg[3:6], cfg[6:9])
        assert rel_err(y, g.t(name + ':y')) < TOL, name
    for name in ['mp133', 'mp333s2', 'mp222', 'mp333s1', 'mp122']:
        cfg = g.t(name + ':cfg').tolist()
        y = F.max_pool3d(g.t(name + ':x'), cfg[0:3], cfg[3:6], cfg[6:9])
        assert torch.equal(y, g.t(name + ':y')), name


def test_r2plus1d_blocks(golden):
    g = golden('blocks')
    blk = oenc.R2BasicBlock(16, 16)
    blk.load_state_dict(g.group('bb:w:'))
    blk.train()
    assert rel_err(blk(g.t('bb:x')), g.t('bb:y')) < TOL
    after = g.group('bb:after:')
    for k, v in blk.state_dict().items():
        assert rel_err(v.float(), after[k].float()) < TOL, k
    ds = nn.Sequential(nn.Conv3d(16, 32, 1, 2, bias=False), nn.BatchNorm3d(32))
    b2 = oenc.R2BasicBlock(16, 32, 2, ds)
    b2.load_state_dict(g.group('bbs:w:'))
    assert rel_err(b2(g.t('bbs:x')), g.t('bbs:y')) < TOL


def test_s3d_blocks(golden):
    g = golden('blocks')
    sep = oenc.S3DSep(3, 64, 7, 2, 3)
    sep.load_state_dict(g.group('sep:w:'))
    assert rel_err(sep(g.t('sep:x')), g.t('sep:y')) < TOL
    mix = oenc.S3DMixed(*oenc.S3D_MIXED['3b'])
    mix.load_state_dict(g.group('m3b:w:'))
    assert rel_err(mix(g.t('m3b:x')), g.t('m3b:y')) < TOL


def test_r3d_bottleneck(golden):
    g = golden('blocks')
    ds = nn.Sequential(nn.Conv3d(16, 16, 1, stride=2, bias=False), nn.BatchNorm3d(16))
    bt = oenc.R3Bottleneck(16, 4, 2, ds)
    bt.load_state_dict(g.group('r3b:w:'))
    assert rel_err(bt(g.t('r3b:x')), g.t('r3b:y')) < TOL


def test_r2plus1d_tiny_model_fwd_bwd(golden):
    g = golden('r2p1d_tiny')
    m = oenc.R2Plus1D(10, widen_factor=0.125)
    m.load_state_dict(g.group('r2t:w:'))
    x = g.x('r2t:xspec').requires_grad_(True)
    m.train()
    y = m(x)
    assert rel_err(y, g.t('r2t:y_train')) < TOL
    y.square().sum().backward()
    assert rel_err(x.grad, g.t('r2t:dx')) < 1e-4
    assert rel_err(m.conv1_s.weight.grad, g.t('r2t:dw_conv1_s')) < 1e-4
    assert rel_err(m.layer4[0].conv2_t.weight.grad, g.t('r2t:dw_l4_conv2_t')) < 1e-4
    assert rel_err(m.fc.weight.grad, g.t('r2t:dw_fc')) < 1e-4
    assert rel_err(m.bn1_s.weight.grad, g.t('r2t:dg_bn1_s')) < 1e-4
    after = g.group('r2t:after:')
    for k, v in m.state_dict().items():
        assert rel_err(v.float(), after[k].float()) < 1e-4, k
    m.eval()
    assert rel_err(m(x), g.t('r2t:y_eval')) < TOL


@pytest.mark.parametrize('tag,ctor,strip_fc', [
    ('s3d', lambda: oenc.S3D(), True), ('r18', lambda: oenc.R2Plus1D(18), True),
    ('r3d18', lambda: oenc.R3D(18, 96, 16), False)])
def test_full_size_encoders_from_seed(golden, tag, ctor, strip_fc):
    """make_golden.py asserted that the same seed gives the reference's exact weights; here the
    seeded oracle must reproduce the reference's output on the stored input."""
    g = golden('encoders_seeded')
    torch.manual_seed(int(g.t(tag + ':seed')))
    m = ctor()
    m.train()
    if strip_fc:
        m.fc = nn.Identity()
    with torch.no_grad():
        y = m(g.x(tag + ':xspec'))
    assert rel_err(y, g.t(tag + ':y_train')) < 1e-4
    if tag == 's3d':
        assert rel_err(m.base[0].bn_s.running_mean, g.t('s3d:rm_base0_bn_s')) < 1e-4


def test_project_head(golden):
    g = golden('moco')
    h = owrap.ProjectHead(24, 16, 'mlp')
    h.load_state_dict(g.group('head:w:'))
    assert rel_err(h(g.t('head:x')), g.t('head:y')) < TOL


@pytest.mark.parametrize('tag,K,D,steps', [('k8', 8, 16, 5), ('k256', 256, 128, 3)])
def test_moco_queue_trace(golden, tag, K, D, steps):
    g = golden('moco')
    mo = omoco.RGBMoCo(D, K=K, T=0.07)
    mo.memory.copy_(g.t(tag + ':mem0'))
    crit = omoco.NCESoftmaxLoss()
    for s in range(steps):
        q = g.t('%s:q%d' % (tag, s)).clone().requires_grad_(True)
        logits, labels = mo(q, g.t('%s:k%d' % (tag, s)))
        loss = crit(logits)
        loss.backward()
        assert rel_err(logits, g.t('%s:logits%d' % (tag, s))) < TOL
        assert torch.equal(labels, g.t('%s:labels%d' % (tag, s)))
        assert rel_err(loss, g.t('%s:loss%d' % (tag, s))) < TOL
        assert rel_err(q.grad, g.t('%s:dq%d' % (tag, s))) < TOL
        assert torch.equal(mo.memory, g.t('%s:mem%d' % (tag, s + 1)))          # byte copy of k rows
        assert mo.index == int(g.t('%s:ptr%d' % (tag, s + 1)))
        p1, = omoco.accuracy(logits.detach(), labels, topk=(1,))
        assert torch.equal(p1, g.t('%s:prec1_%d' % (tag, s)))


def test_moco_allk_wraparound(golden):
    g = golden('moco')
    mo = omoco.RGBMoCo(16, K=8, T=0.07)
    mo.memory.copy_(g.t('allk:mem0'))
    mo.index = 6
    lg, _ = mo(g.t('allk:q'), g.t('allk:k'), all_k=g.t('allk:all_k'))
    assert rel_err(lg, g.t('allk:logits')) < TOL
    assert torch.equal(mo.memory, g.t('allk:mem1'))
    assert mo.index == int(g.t('allk:ptr1')) == 2


def test_hop_distance(golden):
    g = golden('graph')
    for T in (2, 4, 8, 16):
        assert torch.equal(ograph.hop_distance(T, 3), g.t('hop:T%d' % T))


def test_temporal_graph_aug(golden):
    g = golden('graph')
    aug = ograph.TemporalGraphAug(32)
    aug.load_state_dict(g.group('aug:w:'))
    x = g.t('aug:x').clone().requires_grad_(True)
    sim = aug.sim_adj(x)
    assert rel_err(sim, g.t('aug:sim')) < TOL
    pre = aug.hop_weighted(sim, ograph.hop_distance(8, 3))
    assert rel_err(pre, g.t('aug:pre')) < TOL
    adj = ograph.relaxed_bernoulli_rsample(pre, g.t('aug:u'), 1.0)
    assert rel_err(adj, g.t('aug:adj')) < TOL
    with torch.no_grad():
        assert rel_err(aug(g.t('aug:x'), adj=g.t('aug:adj')), g.t('aug:y')) < TOL
    y = aug(x, u=g.t('aug:u'))          # gradient also flows through sim -> adj (rsample is reparameterised)
    assert rel_err(y, g.t('aug:y')) < TOL
    y.backward(g.t('aug:dy'))
    assert rel_err(x.grad, g.t('aug:dx')) < 1e-4
    assert rel_err(aug.gcns[0].conv.weight.grad, g.t('aug:dw_gcn')) < 1e-4
    assert rel_err(aug.g_q[0].weight.grad, g.t('aug:dw_gq')) < 1e-4
    assert rel_err(aug.g_k[0].weight.grad, g.t('aug:dw_gk')) < 1e-4
    with torch.no_grad():
        yf = aug(g.t('aug:x'), u=g.t('aug:u_full_seed53'))
    assert rel_err(yf, g.t('aug:y_full_seed53')) < TOL


def test_simsiam_loss_and_grads(golden):
    g = golden('steps')
    oenc.BACKBONES['R2P1D10T'] = lambda: oenc.R2Plus1D(10, widen_factor=0.125)
    model, ema = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'simsiam')
    assert ema is None
    model.load_state_dict(g.group('ss:w:'))
    model.train()
    loss = model(g.x('ss:xspec'))
    assert rel_err(loss, g.t('ss:loss')) < 1e-4
    loss.backward()
    sm = model.model
    assert rel_err(sm.prediction.l2.weight.grad, g.t('ss:dw_pred_l2')) < 1e-3
    assert rel_err(sm.projection.l1[0].weight.grad, g.t('ss:dw_proj_l1')) < 1e-3
    assert rel_err(sm.encoder.base_model.conv1_s.weight.grad, g.t('ss:dw_conv1_s')) < 1e-3


def test_moco_two_step_trace(golden):
    """Three full _train_moco iterations (queue wraps: K=20, 8 keys per step) (tools/train_video_contrast_dis.py:395-454) against the
    trace produced with the reference's model/queue/criterion/optimiser classes."""
    g = golden('steps')
    oenc.BACKBONES['R2P1D10T'] = lambda: oenc.R2Plus1D(10, widen_factor=0.125)
    model, ema = owrap.create_visual_model('R2P1D10T', 8, 32, 'mlp', 'moco')
    model.load_state_dict(g.group('mo:w:'))
    ema.load_state_dict(g.group('mo:w:'))
    contrast = omoco.RGBMoCo(32, K=20, T=0.07)
    contrast.memory.copy_(g.t('mo:mem0'))
    opt = omoco.make_optimizer(model, 0.06, 0.9, 5e-4)
    # make_lr_scheduler runs before the first iteration (tools/...dis.py:117): epoch-0 warm-up factor applies
    f0 = omoco.warmup_multistep_factor(0)
    for gr in opt.param_groups:
        gr['lr'] = gr['lr'] * f0
    names = [n for n, _ in model.named_parameters()]
    assert names == [str(s) for s in g.z['mo:group_names']]
    assert np.allclose([gr['lr'] for gr in opt.param_groups], g.z['mo:group_lr'])
    assert np.allclose([gr['weight_decay'] for gr in opt.param_groups], g.z['mo:group_wd'])
    model.train()
    omoco.set_key_encoder_mode(ema)
    crit = omoco.NCESoftmaxLoss()
    for it in range(3):
        r = omoco.moco_train_step(model, ema, contrast, crit, opt, g.x('mo:xspec%d' % it), 0.999,
                                  shuffle_ids=g.t('mo:shuffle%d' % it))
        assert rel_err(r['loss'], g.t('mo:loss%d' % it)) < 1e-4
        assert rel_err(r['logits'], g.t('mo:logits%d' % it)) < 1e-3
        assert rel_err(r['q'], g.t('mo:q%d' % it)) < 1e-3
        assert rel_err(r['k'], g.t('mo:k%d' % it)) < 1e-3
    after = g.group('mo:after:')
    for k, v in model.state_dict().items():
        assert rel_err(v.float(), after[k].float()) < 1e-3, k
    ek = ema.state_dict()
    for k, v in g.group('mo:afterk:').items():
        assert rel_err(ek[k].float(), v.float()) < 1e-3, k
    assert rel_err(contrast.memory, g.t('mo:mem3')) < 1e-3
    assert contrast.index == int(g.t('mo:ptr3')) == 4


def test_lr_schedule(golden):
    g = golden('steps')
    want = g.z['lr:weights_epoch0_199']
    got = [0.06 * omoco.warmup_multistep_factor(e) for e in range(200)]
    assert np.allclose(got, want, rtol=1e-6)


def test_input_stage_normalize_and_to_tensor(golden):
    """oracle.input against the reference's VideoNormalize / VideoToTensor (consistency_transforms.py:11-65): bit for bit --
    both are the same fp32 numpy statements."""
    from oracle import input as oinput
    g = golden('input')
    for tag in ('a', 'b'):
        frames = g.z[tag + ':frames']
        mean, std = tuple(g.z[tag + ':mean']), tuple(g.z[tag + ':std'])
        n0 = oinput.video_normalize(frames[0], mean, std)
        assert n0.dtype == np.float32 and np.array_equal(n0, g.z[tag + ':norm0'])
        ten = oinput.video_to_tensor([oinput.video_normalize(f, mean, std) for f in frames])
        assert torch.equal(ten, g.t(tag + ':tensor'))
        # the composed sample builder with an identity crop / no flip is the same thing, twice, on the channel axis
        T, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        s = oinput.make_sample(np.stack([frames, frames]), [(0, 0, 0), (0, 0, 0)], H, W, mean, std)
        assert tuple(s.shape) == (6, T, H, W) and torch.equal(s[:3], ten) and torch.equal(s[3:], ten)
    # index maps restated from albumentations / cv2 (third party, not in the reference tree): self-consistency only
    img = np.arange(2 * 5 * 3, dtype=np.uint8).reshape(2, 5, 3)
    assert np.array_equal(oinput.hflip(img)[:, 0], img[:, 4]) and np.array_equal(oinput.hflip(oinput.hflip(img)), img)
    assert oinput.random_crop_coords(10, 20, 4, 8, 0.0, 0.999) == (0, 11) and oinput.random_crop_coords(10, 20, 4, 8, 0.5, 0.5) == (3, 6)
