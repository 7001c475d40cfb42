#!/usr/bin/env python3
"""Exact-fp32 vs bf16x3 conv kernels: time per tile shape and the error of bf16x3 against the exact result."""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GCA_AUTOTUNE'] = '0'
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_micro import LAYERS, ev

ap = argparse.ArgumentParser()
ap.add_argument('--layers', default='L00,L01,L02,L03,L12,L14,L21,L30')
ap.add_argument('--tiles', default='64,96,128,160,1088,1152')
ap.add_argument('--wgrad', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda:0')


def relerr(a_, b_):
    return float((a_ - b_).abs().max() / b_.abs().max()), float((a_ - b_).norm() / b_.norm())


for name in a.layers.split(','):
    C, D, H, W, K, k, s, p = LAYERS[name]
    shp = (32, C, D, H, W)
    torch.manual_seed(0)
    x = torch.randn(shp, device=dev)
    w = torch.randn((K, C) + k, device=dev) * 0.05
    ref = {}
    for mode in ('f32', 'bf16x6', 'bf16x3'):
        ops.set_conv_math(mode)
        plan = ops.ConvPlan(32, C, D, H, W, K, k, s, p, dev)
        N, _, OD, OH, OW = plan.out_shape
        fl = 2.0 * N * K * OD * OH * OW * C * k[0] * k[1] * k[2]
        dy = torch.randn(plan.out_shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        wp0, wp1 = ops.conv_pack(plan, 0, w), ops.conv_pack(plan, 1, w)
        dx = torch.empty(shp, device=dev)
        for bm in [int(v) for v in a.tiles.split(',')]:
            plan.g.tune_fwd_bm, plan.g.tune_fwd_splits = bm, 1
            plan.g.tune_dgrad_bm, plan.g.tune_dgrad_splits = bm, 1
            plan.refresh()
            try:
                y = ops.conv_fwd(plan, x, wp0, None, stats=True)[0]
                ops.conv_dgrad(plan, dy, wp1, dx, False)
                tf = ev(lambda: ops.conv_fwd(plan, x, wp0, None, stats=True), 5)
                td = ev(lambda: ops.conv_dgrad(plan, dy, wp1, dx, False), 5)
            except RuntimeError as e:
                print(name, mode, bm, 'ERR', e); continue
            if mode == 'f32' and 'y' not in ref:
                ref['y'], ref['dx'] = y.clone(), dx.clone()
            ey, edx = relerr(y, ref['y']), relerr(dx, ref['dx'])
            print('%s %-6s tile %4d | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | err vs f32: y max %.1e rms %.1e  dx max %.1e rms %.1e' %
                  (name, mode, bm, tf, fl / 1e9 / tf, td, fl / 1e9 / td, ey[0], ey[1], edx[0], edx[1]), flush=True)
        if a.wgrad:
            dw = torch.empty_like(w)
            ops.conv_wgrad(plan, x, dy, dw, False)
            tw = ev(lambda: ops.conv_wgrad(plan, x, dy, dw, False), 5)
            if mode == 'f32':
                ref['dw'] = dw.clone()
            ew = relerr(dw, ref['dw'])
            print('%s %-6s wgrad cfg %s %7.3f ms %6.1f TF | err max %.1e rms %.1e' % (name, mode, plan.cfg(2), tw, fl / 1e9 / tw, ew[0], ew[1]), flush=True)
ops.set_conv_math('f32')
