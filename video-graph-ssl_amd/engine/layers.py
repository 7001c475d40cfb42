"""Leaf modules (same parameter/buffer names as torch.nn so reference checkpoints load by key)
and the fused functional units the encoders are written in.

Every ``f_*`` function takes the tape + Vars, launches HIP kernels for the forward and records
ONE closure for the backward.  Parameter gradients are accumulated (+=) straight into
``param.grad`` (views of the flat gradient arena, engine/arena.py).
"""
import math

import torch
import torch.nn as nn

from . import ops
from . import tape as _tape
from .tape import LazyVar, Var


GRAD_LOG = None        # a list while a trainer maps parameters to backward closures: (closure index, parameter)


def _grad_of(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p.data)
    if GRAD_LOG is not None:
        GRAD_LOG.append((_tape.CURRENT[0], p))
    return p.grad


# ----------------------------------------------------------------------------- leaf modules
class _PackedWeight(object):
    """Weights are consumed in the MFMA-friendly packed layout (gca_conv_pack).  A layer packs on use, unless a
    BatchedPacker has already refreshed this step's copy for the same plan (one launch for the whole encoder)."""

    def _init_pack(self):
        self._pack, self._pack_plan, self._prepacked = [None, None], [None, None], [False, False]
        self._pack_layout = [None, None]      # gca_conv_pack_layout() the buffer was last written in

    def packed(self, plan, which):
        if self._prepacked[which] and self._pack_plan[which] is plan:
            # plans are shared per geometry: if a launch-tuning decision moved this class to another packed layout after
            # the batched pack was planned, the buffer no longer holds what the kernel will read
            if ops.H.lib.gca_conv_pack_layout(plan.gp, which) != self._pack_layout[which]:
                raise RuntimeError('packed weights are in a stale layout for this plan; rebuild the BatchedPacker')
            return self._pack[which]
        self._pack[which] = ops.conv_pack(plan, which, self.weight.data, self._pack[which])
        self._pack_plan[which] = plan
        self._pack_layout[which] = ops.H.lib.gca_conv_pack_layout(plan.gp, which)
        return self._pack[which]


class BatchedPacker(object):
    """One gca_conv_pack_batched launch for every (layer, direction) of `module` that has been used at least
    once (so its plan and packed buffer exist).  run() at the start of a step, release() at its end."""

    def __init__(self, module, directions):
        import ctypes
        H = ops.H
        recs, self.layers = [], []
        dev = None
        for m in module.modules():
            if not isinstance(m, _PackedWeight):
                continue
            for which in directions:
                plan, buf = m._pack_plan[which], m._pack[which]
                if plan is None or buf is None:
                    continue
                n = H.lib.gca_conv_pack_jobs_host(plan.gp, which, None, None, None)
                if n <= 0:
                    raise RuntimeError('gca_conv_pack_jobs_host: %d' % n)
                raw = ctypes.create_string_buffer(int(n) * H.PACK_JOB_BYTES)
                if H.lib.gca_conv_pack_jobs_host(plan.gp, which, m.weight.data_ptr(), buf.data_ptr(), raw) != n:
                    raise RuntimeError('gca_conv_pack_jobs_host failed')
                recs.append(raw.raw)
                m._pack_layout[which] = H.lib.gca_conv_pack_layout(plan.gp, which)     # what these jobs write
                self.layers.append((m, which, plan, buf))
                dev = buf.device
        blob = ctypes.create_string_buffer(b''.join(recs))
        self.n = (len(blob) - 1) // H.PACK_JOB_BYTES
        self.jobs, self.blocks = None, 0
        if self.n:
            self.blocks = H.lib.gca_conv_pack_jobs_finalize_host(blob, self.n)
            if self.blocks <= 0:
                raise RuntimeError('gca_conv_pack_jobs_finalize_host: %d' % self.blocks)
            self.jobs = torch.frombuffer(bytearray(blob.raw[:self.n * H.PACK_JOB_BYTES]), dtype=torch.uint8).to(dev)

    def run(self):
        if not self.n:
            return
        for m, which, plan, buf in self.layers:
            if m._pack[which] is not buf or m._pack_plan[which] is not plan:
                raise RuntimeError('BatchedPacker is stale (a layer was re-planned); rebuild it')
            m._prepacked[which] = True
        ops.H.call('gca_conv_pack_batched', ops.ptr(self.jobs), self.n, self.blocks, ops.stream())

    def release(self):
        for m, which, _, _ in self.layers:
            m._prepacked[which] = False


class HipConv3d(nn.Module, _PackedWeight):
    """nn.Conv3d(bias=False|True) parameters; default init = torch's (kaiming_uniform a=sqrt 5)."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        t3 = lambda v: (v, v, v) if isinstance(v, int) else tuple(v)
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding = t3(kernel_size), t3(stride), t3(padding)
        self.weight = nn.Parameter(torch.empty(cout, cin, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2])
            nn.init.uniform_(self.bias, -bound, bound)
        self._init_pack()

    def plan(self, x):
        xs = 0
        if not x.is_contiguous():
            N, Cc, D, Hh, W = x.shape
            if x.stride()[1:] != (D * Hh * W, Hh * W, W, 1):
                raise RuntimeError('conv input must be NCDHW-contiguous inside each clip')
            xs = x.stride(0)
        return ops.conv_plan(tuple(x.shape), self.out_channels, self.kernel_size, self.stride, self.padding,
                             x.device, xs, act_f16=x.dtype is torch.float16)

    def forward(self, x):      # plain inference use of the leaf (no BN fusion)
        return ops.conv_fwd(self.plan(x), x, self.packed(self.plan(x), 0), None if self.bias is None else self.bias.data)


class _HipBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))


class HipBatchNorm3d(_HipBatchNorm):
    pass


class HipBatchNorm1d(_HipBatchNorm):
    pass


class HipLinear(nn.Module, _PackedWeight):
    def __init__(self, fin, fout, bias=True):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(fin)
            nn.init.uniform_(self.bias, -bound, bound)
        self._init_pack()

    def plan(self, b, device):
        return ops.conv_plan((b, self.in_features, 1, 1, 1), self.out_features, 1, 1, 0, device)


class HipMaxPool3d(nn.Module):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        t3 = lambda v: (v, v, v) if isinstance(v, int) else tuple(v)
        self.kernel_size = t3(kernel_size)
        self.stride = t3(stride if stride is not None else kernel_size)
        self.padding = t3(padding)


class HipAvgPool3d(nn.Module):
    """nn.AvgPool3d(kernel_size) with its default stride (= kernel size) and no padding."""

    def __init__(self, kernel_size):
        super().__init__()
        t3 = lambda v: (v, v, v) if isinstance(v, int) else tuple(v)
        self.kernel_size = self.stride = t3(kernel_size)
        self.padding = (0, 0, 0)


class HipReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()


class HipIdentity(nn.Module):
    pass


class HipNormalize(nn.Module):
    def __init__(self, p=2):
        super().__init__()
        self.p = p


# ----------------------------------------------------------------------------- fused functional units
def _bn_scale_shift(bn, ss, sq, count):
    return ops.bn_finalize(ss, sq, count, bn.weight.data, bn.bias.data, bn.eps, bn.momentum,
                           bn.running_mean, bn.running_var, bn.num_batches_tracked)


def _stored(x):
    """fp16-storage path (ops.set_conv_math('fp16')): the fp32 clip is cast once, where it enters the first conv."""
    if ops.ACT_F16[0] and x.dtype is torch.float32 and x.dim() == 5:
        return ops.cast_f16(x)
    return x


# GCA_LAZY_BN=0: every BatchNorm writes its normalised tensor (A/B runs)
import os as _os
LAZY_BN = _os.environ.get('GCA_LAZY_BN', '1') != '0'
BN_SMALL_ELEMS = 32768      # bn.hip: N*SP at or below this runs finalize + apply as ONE launch


def _conv_input(conv, xv):
    """-> (plan, x, xf): what a conv reads.  A LazyVar producer (conv -> BN -> ReLU not yet written) is consumed as (y, (scale,
    shift)) when this conv's pinned launch shapes can apply the BatchNorm + ReLU while staging (ops.conv_xf_ok: forward on an
    LDS-halo kernel, weight gradient on a streaming kernel; resnet2p1d.py:66-85's bn1_s -> conv1_t, bn1_t -> conv2_s,
    bn2_s -> conv2_t and the stem, s3d_1.py:61-68's conv_s -> conv_t); otherwise its `.t` materialises z."""
    if isinstance(xv, LazyVar) and not xv.materialized and LAZY_BN:
        plan = conv.plan(xv.y)
        if ops.conv_xf_ok(plan):
            return plan, xv.y, (xv.scale, xv.shift)
    x = _stored(xv.t)
    return conv.plan(x), x, None


def f_seq(tape, module, xv):
    """Run a container: modules with their own ``fwd`` (blocks, graph-wrapped modules), max pools and
    plain nn.Sequential chains of those."""
    if hasattr(module, 'fwd'):
        return module.fwd(tape, xv)
    if isinstance(module, HipMaxPool3d):
        return f_maxpool(tape, module, xv)
    if isinstance(module, HipAvgPool3d):
        return f_avgpool(tape, module, xv)
    if isinstance(module, HipConv3d):
        return f_conv(tape, module, xv)
    if isinstance(module, nn.Sequential):
        for m in module:
            xv = f_seq(tape, m, xv)
        return xv
    raise TypeError('engine cannot run %r' % type(module))


def f_concat(tape, buf, branch_vars, offsets):
    """Channel concat without a copy: the branches already wrote their outputs into channel slices of
    `buf` (f_conv_bn_act(out=...)); on the way back each branch reads its slice of d(buf)."""
    outv = Var(buf, tape.recording)

    def back():
        g = outv.grad
        for bv, off in zip(branch_vars, offsets):
            bv.grad = g[:, off:off + bv.t.shape[1]]
        outv.grad = None
    tape.record(back)
    return outv


def f_conv_bn_act(tape, conv, bn, xv, relu=True, residual=None, out=None):
    """z = [relu]( BN(conv(x)) [+ residual] ).  Training-mode BN takes its batch statistics from the
    conv epilogue (no extra pass over y).  Follows resnet2p1d.py:66-85 / s3d_1.py:43-47,61-68.
    `out`: optional channel-slice view of a wider buffer to receive z (Inception concat).
    A plain conv -> BN -> ReLU unit (no residual, no slice) returns a LazyVar: z is written only if some consumer asks for it."""
    plan, x, xf = _conv_input(conv, xv)
    N, K, OD, OH, OW = plan.out_shape
    SP = OD * OH * OW
    wp = conv.packed(plan, 0)
    train = bn.training
    # (small maps keep the one-launch finalize + apply kernel of gca_bn_train_fwd: deferring their apply would ADD a launch,
    # and their consumers run on the gather kernels anyway)
    lazy = (LAZY_BN and train and relu and residual is None and out is None and x.dtype is torch.float32
            and N * SP > BN_SMALL_ELEMS)
    z = None
    if train:
        fused = (ops.FUSE_SPLITK_BN and xf is None and not lazy and plan.tuned[0] and x.dtype is torch.float32
                 and N * SP <= BN_SMALL_ELEMS and conv.bias is None and plan.cfg(0)[2] > 1 and plan.fwd_ws > 0)
        if fused:
            # the conv's split-K slabs go straight into the BatchNorm kernel: no finishing launch
            y, z, mean, invstd, scale, shift = ops.conv_bn_small_fwd(
                plan, x, wp, N * SP, bn.weight.data, bn.bias.data, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                bn.num_batches_tracked, None if residual is None else residual.t, relu, out=out)
        elif xf is not None:
            y, (ss, sq) = ops.conv_fwd_xf(plan, x, xf[0], xf[1], wp, stats=True)
        else:
            y, (ss, sq) = ops.conv_fwd(plan, x, wp, None, stats=True, w_raw=conv.weight.data)
        if fused:
            pass
        elif lazy:
            mean, invstd, scale, shift = _bn_scale_shift(bn, ss, sq, N * SP)
        else:
            z, mean, invstd, scale, shift = ops.bn_train_fwd(ss, sq, N * SP, bn.weight.data, bn.bias.data, bn.eps, bn.momentum,
                                                             bn.running_mean, bn.running_var, bn.num_batches_tracked, y,
                                                             None if residual is None else residual.t, relu, N, K, SP, out=out)
    else:
        if xf is not None:                   # (a training-mode producer in front of an eval-mode unit: take its tensor)
            x, xf = _stored(xv.t), None
        y = ops.conv_fwd(plan, x, wp, None, w_raw=conv.weight.data)
        mean = invstd = None
        scale, shift = ops.bn_fold_eval(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps)
        z = ops.bn_apply(y, scale, shift, None if residual is None else residual.t, relu, N, K, SP, out=out)
    zv = LazyVar(y, scale, shift, tape.recording) if lazy else Var(z, tape.recording)
    if tape.recording:
        if not train:
            raise NotImplementedError('backward through eval-mode BatchNorm is not on the pre-training path')

        def back():
            dres = racc = None
            if residual is not None and residual.needs_grad:
                dres, racc = residual.grad_buffer()
            # ReLU mask: with no residual in front of the ReLU it is recomputed from y (the forward's own scale/shift),
            # so z is not streamed again; with a residual the saved output is the only record
            mode = (2 if residual is None else 1) if relu else 0
            dy = ops.bn_bwd(zv.grad, z, y, bn.weight.data, mean, invstd, mode, N, K, SP,
                            _grad_of(bn.weight), _grad_of(bn.bias), dres, bool(racc), scale, shift)
            if DEBUG_GRADS is not None:          # diagnostics only: gradient wrt the conv output, per BN module
                DEBUG_GRADS[id(bn)] = (zv.grad.clone(), dy.clone())
            zv.grad = None
            ops.conv_wgrad(plan, x, dy, _grad_of(conv.weight), accumulate=True, xf=xf)
            if xv.needs_grad:
                buf, acc = xv.grad_buffer()
                ops.conv_dgrad(plan, dy, conv.packed(plan, 1), buf, acc, w_raw=conv.weight.data)
        tape.record(back)
    return zv


DEBUG_GRADS = None


def f_conv(tape, conv, xv):
    """Plain convolution (optional bias), no normalisation: temporal_graph.py:46,119-122."""
    x = _stored(xv.t)
    plan = conv.plan(x)
    y = ops.conv_fwd(plan, x, conv.packed(plan, 0), None if conv.bias is None else conv.bias.data, w_raw=conv.weight.data)
    yv = Var(y, tape.recording)

    def back():
        dy = yv.grad
        yv.grad = None
        ops.conv_wgrad(plan, x, dy, _grad_of(conv.weight), accumulate=True)
        if conv.bias is not None:
            N, K, OD, OH, OW = plan.out_shape
            ops.bias_grad(dy, N, K, OD * OH * OW, _grad_of(conv.bias), True)
        if xv.needs_grad:
            buf, acc = xv.grad_buffer()
            ops.conv_dgrad(plan, dy, conv.packed(plan, 1), buf, acc, w_raw=conv.weight.data)
    tape.record(back)
    return yv


def f_linear(tape, lin, xv):
    """y = x W^T + b through the conv kernels as a 1x1x1 conv on (b, C, 1, 1, 1)."""
    x = xv.t
    b = x.shape[0]
    plan = lin.plan(b, x.device)
    y = ops.conv_fwd(plan, x, lin.packed(plan, 0), None if lin.bias is None else lin.bias.data,
                     w_raw=lin.weight.data).view(b, lin.out_features)
    yv = Var(y, tape.recording)

    def back():
        dy = yv.grad
        yv.grad = None
        ops.conv_wgrad(plan, x, dy, _grad_of(lin.weight), accumulate=True)
        if lin.bias is not None:
            ops.bias_grad(dy, b, lin.out_features, 1, _grad_of(lin.bias), True)
        if xv.needs_grad:
            buf, acc = xv.grad_buffer()
            ops.conv_dgrad(plan, dy, lin.packed(plan, 1), buf, acc, w_raw=lin.weight.data)
    tape.record(back)
    return yv


def f_bn1d_act(tape, bn, xv, relu):
    """BatchNorm1d (+ReLU) on (b, C): project_head.py:39-50,64-68."""
    x = xv.t
    b, Cc = x.shape
    if bn.training:
        ss, sq = ops.bn_stats(x, b, Cc, 1)
        z, mean, invstd, scale, shift = ops.bn_train_fwd(ss, sq, b, bn.weight.data, bn.bias.data, bn.eps, bn.momentum,
                                                         bn.running_mean, bn.running_var, bn.num_batches_tracked, x, None,
                                                         relu, b, Cc, 1)
    else:
        mean = invstd = None
        scale, shift = ops.bn_fold_eval(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps)
        z = ops.bn_apply(x, scale, shift, None, relu, b, Cc, 1)
    zv = Var(z, tape.recording)

    def back():
        if mean is None:
            raise NotImplementedError('backward through eval-mode BatchNorm1d')
        dx = ops.bn_bwd(zv.grad, z, x, bn.weight.data, mean, invstd, 2 if relu else 0, b, Cc, 1,
                        _grad_of(bn.weight), _grad_of(bn.bias), None, False, scale, shift)
        zv.grad = None
        if xv.needs_grad:
            xv.add_grad(dx)
    tape.record(back)
    return zv


def f_relu(tape, xv):
    y = ops.relu_fwd(xv.t)
    yv = Var(y, tape.recording)

    def back():
        if xv.needs_grad:
            xv.add_grad(ops.relu_bwd(yv.grad, y))
        yv.grad = None
    tape.record(back)
    return yv


def f_conv_bn_relu_maxpool(tape, conv, bn, pool, xv):
    """maxpool(relu(BN(conv(x)))) with the BN+ReLU evaluated inside the pooling kernel: the normalised tensor (the
    largest activation of the R(2+1)D / 3D-ResNet stems, resnet2p1d.py:252-255, resnet.py:176-179) is never written.
    Backward: pool gather -> BN backward with the ReLU mask recomputed from the conv output -> wgrad / dgrad."""
    plan, x, xf = _conv_input(conv, xv)
    N, K, OD, OH, OW = plan.out_shape
    SP = OD * OH * OW
    wp = conv.packed(plan, 0)
    if bn.training:
        if xf is not None:
            y, (ss, sq) = ops.conv_fwd_xf(plan, x, xf[0], xf[1], wp, stats=True)
        else:
            y, (ss, sq) = ops.conv_fwd(plan, x, wp, None, stats=True, w_raw=conv.weight.data)
        mean, invstd, scale, shift = _bn_scale_shift(bn, ss, sq, N * SP)
    else:
        if xf is not None:
            x, xf = _stored(xv.t), None
        y = ops.conv_fwd(plan, x, wp, None, w_raw=conv.weight.data)
        mean = invstd = None
        scale, shift = ops.bn_fold_eval(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps)
    pplan = ops.pool_plan(tuple(y.shape), pool.kernel_size, pool.stride, pool.padding)
    p, am = ops.maxpool_fwd(pplan, y, want_argmax=tape.recording, scale=scale, shift=shift)
    pv = Var(p, tape.recording)
    if tape.recording:
        if mean is None:
            raise NotImplementedError('backward through eval-mode BatchNorm is not on the pre-training path')

        def back():
            dz = ops.maxpool_bwd(pplan, pv.grad, am)
            pv.grad = None
            dy = ops.bn_bwd(dz, None, y, bn.weight.data, mean, invstd, 2, N, K, SP, _grad_of(bn.weight), _grad_of(bn.bias),
                            None, False, scale, shift)
            del dz
            ops.conv_wgrad(plan, x, dy, _grad_of(conv.weight), accumulate=True, xf=xf)
            if xv.needs_grad:
                buf, acc = xv.grad_buffer()
                ops.conv_dgrad(plan, dy, conv.packed(plan, 1), buf, acc, w_raw=conv.weight.data)
        tape.record(back)
    return pv


def f_maxpool(tape, pool, xv):
    x = xv.t
    plan = ops.pool_plan(tuple(x.shape), pool.kernel_size, pool.stride, pool.padding)
    y, am = ops.maxpool_fwd(plan, x, want_argmax=tape.recording and xv.needs_grad)
    yv = Var(y, tape.recording)

    def back():
        if xv.needs_grad:
            buf, acc = xv.grad_buffer()
            ops.maxpool_bwd(plan, yv.grad, am, buf, acc)
        yv.grad = None
    tape.record(back)
    return yv


def f_avgpool(tape, pool, xv):
    x = xv.t
    plan = ops.pool_plan(tuple(x.shape), pool.kernel_size, pool.stride, pool.padding)
    yv = Var(ops.avgpool_fwd(plan, x), tape.recording)

    def back():
        if xv.needs_grad:
            buf, acc = xv.grad_buffer()
            ops.avgpool_bwd(plan, yv.grad, buf, acc)
        yv.grad = None
    tape.record(back)
    return yv


def f_dropout(tape, drop, xv):
    """nn.Dropout on the pooled (N, C) features (visual_wrappers.py:109-110, MODEL.DROPOUT > 0).  The keep mask is drawn
    by torch's generator on the device (as nn.Dropout does: reproducible under torch.manual_seed); mask and 1/(1-p)
    scaling are one elementwise product -- a (N, C) tensor of a few KB, not a hot-path kernel."""
    if drop.p == 0 or not drop.training:
        return xv
    x = xv.t
    if drop.p >= 1:
        mask = torch.zeros_like(x)
    else:
        mask = torch.empty_like(x).bernoulli_(1.0 - drop.p).mul_(1.0 / (1.0 - drop.p))
    yv = Var(x * mask, tape.recording)

    def back():
        if xv.needs_grad:
            xv.add_grad(yv.grad * mask)
        yv.grad = None
    tape.record(back)
    return yv


def f_s3d_tail_with_dropout(tape, drop, xv):
    """S3D's tail when its `fc` has been replaced by nn.Dropout (MODEL.DROPOUT > 0, visual_wrappers.py:109-110):
    avg_pool3d((2,H,W), stride 1) -> dropout on the (B,C,T-1,1,1) map -> mean over time (s3d_1.py:30-33).  The spatial
    means come from the pooling kernel; the (B,C,T) remainder is a few KB of elementwise work."""
    x = xv.t
    N, C, T, Hh, W = x.shape
    m = ops.wavgpool_fwd(x.view(N, C * T, 1, Hh, W), None, 1.0 / (Hh * W)).view(N, C, T)
    pooled = 0.5 * (m[:, :, :-1] + m[:, :, 1:])
    keep = 1.0 - drop.p
    mask = torch.empty_like(pooled).bernoulli_(keep).mul_(1.0 / keep) if keep > 0 else torch.zeros_like(pooled)
    yv = Var((pooled * mask).mean(2), tape.recording)

    def back():
        if xv.needs_grad:
            dp = yv.grad[:, :, None] * mask * (0.5 / (T - 1))
            dm = torch.zeros((N, C, T), dtype=torch.float32, device=x.device)
            dm[:, :, :-1] += dp
            dm[:, :, 1:] += dp
            xv.add_grad(ops.wavgpool_bwd(dm.view(N, C * T), None, 1.0 / (Hh * W), (N, C * T, 1, Hh, W), x.dtype).view(x.shape))
        yv.grad = None
    tape.record(back)
    return yv


def f_wavgpool(tape, xv, wt=None, norm=None):
    """(N,C,D,H,W) -> (N,C): norm * sum_d wt[d] * sum_hw x.  Global mean when wt is None."""
    x = xv.t
    if norm is None:
        norm = 1.0 / (x.shape[2] * x.shape[3] * x.shape[4])
    y = ops.wavgpool_fwd(x, wt, norm)
    yv = Var(y, tape.recording)

    def back():
        if xv.needs_grad:
            xv.add_grad(ops.wavgpool_bwd(yv.grad, wt, norm, tuple(x.shape), x.dtype))
        yv.grad = None
    tape.record(back)
    return yv


def f_l2norm(tape, xv):
    y, inv = ops.l2norm_fwd(xv.t)
    yv = Var(y, tape.recording)

    def back():
        if xv.needs_grad:
            xv.add_grad(ops.l2norm_bwd(yv.grad, y, inv))
        yv.grad = None
    tape.record(back)
    return yv


def f_head_fc(tape, fc, xv):
    """The backbone's last layer after VisualModelWrapper replaced it (visual_wrappers.py:107-110):
    Identity for DROPOUT == 0, otherwise whatever was left there."""
    if isinstance(fc, HipIdentity):
        return xv
    if isinstance(fc, HipLinear):
        return f_linear(tape, fc, xv)
    if isinstance(fc, nn.Dropout):
        return f_dropout(tape, fc, xv)
    raise TypeError('unsupported head module %r' % type(fc))
