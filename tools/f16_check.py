#!/usr/bin/env python3
"""fp16-storage conv kernels, one geometry at a time: forward / dgrad / wgrad under the heuristic configuration, forced
gather tiles and forced halo boxes against fp64 ATen on the same fp16-representable operands; dgrad errors are also broken
down by stride residue of the position (= problem class).

    python tools/f16_check.py
"""
import importlib
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('GCA_AUTOTUNE', '0')

CASES = [
    ((2, 3, 8, 32, 32), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),
    ((2, 64, 4, 28, 28), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ((2, 64, 6, 28, 28), 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((2, 256, 4, 14, 14), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ((2, 128, 4, 14, 14), 512, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
    ((3, 40, 3, 9, 11), 50, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ((2, 24, 5, 10, 10), 36, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ((4, 64, 1, 1, 1), 24, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
]


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def main():
    pkg = importlib.import_module('video-graph-ssl_amd')
    ops = pkg.engine.ops
    dev = torch.device('cuda:0')
    for shape, K, k, s, p in CASES:
        torch.manual_seed(0)
        x = torch.randn(shape).half().float()
        w = (torch.randn((K, shape[1]) + tuple(k)) * (2.0 / (shape[1] * k[0] * k[1] * k[2])) ** 0.5).half().float()
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv3d(xr.double(), wr.double(), None, s, p)
        dy = torch.randn(yr.shape).half().float()
        yr.backward(dy.double())
        yr = yr.detach()
        print('case %s K=%d k=%s s=%s p=%s' % (shape, K, k, s, p), flush=True)
        for label, env in (('heuristic', None), ('gather', '0')):
            if env is None:
                os.environ.pop('GCA_HALO', None)
            else:
                os.environ['GCA_HALO'] = env
            plan = ops.ConvPlan(*shape, K, k, s, p, dev, act_f16=True)
            plan.tuned = [True, True, True]
            xd, dyd, wd = x.to(dev).half(), dy.to(dev).half(), w.to(dev)
            y = ops.conv_fwd(plan, xd, ops.conv_pack(plan, 0, wd), None)
            dx = ops.conv_dgrad(plan, dyd, ops.conv_pack(plan, 1, wd))
            dw = torch.zeros_like(wd)
            ops.conv_wgrad(plan, xd, dyd, dw, accumulate=True)
            torch.cuda.synchronize()
            lay = [oct(H) for H in (pkg._hip.lib.gca_conv_pack_layout(plan.gp, 0), pkg._hip.lib.gca_conv_pack_layout(plan.gp, 1))]
            print('  %-9s layouts fwd %s dgrad %s | y %.2e dx %.2e dw %.2e' % (label, lay[0], lay[1], rel(y, yr), rel(dx, xr.grad), rel(dw, wr.grad)), flush=True)
            e = (dx.double().cpu() - xr.grad).abs() / xr.grad.abs().max()
            for rd in range(s[0]):
                for rh in range(s[1]):
                    for rw in range(s[2]):
                        print('      residue (%d,%d,%d): %.2e' % (rd, rh, rw, float(e[:, :, rd::s[0], rh::s[1], rw::s[2]].max())))
    os.environ.pop('GCA_HALO', None)
    # determinism of one split-K fp16 forward: fresh plans, repeated launches
    shape, K, k, s, p = CASES[4]
    torch.manual_seed(0)
    x = torch.randn(shape).half().to(dev)
    w = (torch.randn((K, shape[1]) + tuple(k)) * 0.1).half().float().to(dev)
    yr = F.conv3d(x.double(), w.double(), None, s, p)
    for rep in range(6):
        plan = ops.ConvPlan(*shape, K, k, s, p, dev, act_f16=True)
        plan.tuned = [True, True, True]
        wp = ops.conv_pack(plan, 0, w)
        ys = [ops.conv_fwd(plan, x, wp, None) for _ in range(3)]
        torch.cuda.synchronize()
        bad = (ys[0].double() - yr).abs() > 5e-3 * yr.abs().max()
        print('rep %d cfg %s ws %d: err %s  wrong %d of %d, rows with wrong %s, zeros among wrong %d' % (
            rep, plan.cfg(0), plan.fwd_ws, ['%.1e' % rel(y, yr) for y in ys], int(bad.sum()), bad.numel(),
            sorted(set(bad.nonzero()[:, 1].tolist()))[:8], int((ys[0][bad] == 0).sum())), flush=True)


if __name__ == '__main__':
    main()
