#!/usr/bin/env python3
"""Check and time the stem weight-gradient kernel (conv3d_wgrad_stem.hip, tune_wgrad_tile 14) against conv_wgrad_kernel
(heuristic shape) and an fp64 host reference on a reduced batch:
    python tools/wgrad_stem_bench.py [bf16x6|bf16x3|fp16]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops
DEV = torch.device('cuda:0')
# name, x shape, K, kernel, stride, pad
STEMS = [('R2P1D L00', (32, 3, 16, 112, 112), 110, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
         ('S3D stem', (4, 3, 16, 224, 224), 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
         ('R3D50 stem', (16, 3, 32, 224, 224), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),
         ('odd small', (2, 3, 5, 38, 48), 40, (3, 5, 7), (1, 2, 2), (1, 2, 3))]


def time_ms(fn, reps=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
    ops.set_conv_math(mode)
    f16 = mode == 'fp16'
    for name, shape, K, k, s, p in STEMS:
        N, C, D, H, W = shape
        torch.manual_seed(1)
        x = torch.randn(shape, device=DEV)
        plan = ops.ConvPlan(N, C, D, H, W, K, k, s, p, DEV, act_f16=f16)
        dy = torch.randn(plan.out_shape, device=DEV)
        if f16:
            x, dy = x.half(), dy.half()
        flops = 2.0 * dy.numel() * C * k[0] * k[1] * k[2]
        plan.tuned = [True, True, True]
        outs = {}
        for label, tile, sp in (('conv_wgrad_kernel (heuristic)', 0, 0), ('stem tile 14 / 128', 14, 128), ('stem tile 14 / 256', 14, 256),
                                ('stem tile 14 / 512', 14, 512), ('stem tile 14 / 1024', 14, 1024), ('stem tile 14 / default', 14, 0)):
            plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = tile, sp
            plan.refresh()
            cfg = plan.cfg(2)
            if tile and cfg[3] & 255 != tile:
                print('%-11s %-30s refused' % (name, label))
                continue
            dw = torch.zeros((K, C) + k, device=DEV)
            ms = time_ms(lambda: ops._conv_wgrad_launch(plan, x, dy, dw, False))
            outs[label] = dw.clone()
            print('%-11s %-30s splits %4d  %8.4f ms %7.1f TF/s' % (name, label, cfg[2], ms, flops / 1e9 / ms), flush=True)
        # reference on two clips, fp64 on the host
        nn = min(N, 2)
        xr, dyr = x[:nn].double().cpu(), dy[:nn].double().cpu()
        ref = torch.nn.grad.conv3d_weight(xr, (K, C) + k, dyr, stride=s, padding=p)
        plan2 = ops.ConvPlan(nn, C, D, H, W, K, k, s, p, DEV, act_f16=f16)
        plan2.tuned = [True, True, True]
        for tile in (0, 14):
            plan2.g.tune_wgrad_tile, plan2.g.tune_wgrad_splits = tile, 7
            plan2.refresh()
            if tile and plan2.cfg(2)[3] & 255 != tile:
                continue
            dw = torch.zeros((K, C) + k, device=DEV)
            ops._conv_wgrad_launch(plan2, x[:nn].contiguous(), dy[:nn].contiguous(), dw, False)
            err = (dw.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
            print('%-11s tile %2d vs fp64 reference (2 clips): max err / max |dw| = %.3e' % (name, tile, err), flush=True)
        labs = [l for l in outs if l.startswith('stem')]
        if labs:
            a, b = outs['conv_wgrad_kernel (heuristic)'], outs[labs[0]]
            print('%-11s stem vs conv_wgrad_kernel: max diff / max = %.3e' % (name, (a - b).abs().max().item() / a.abs().max().item()))


if __name__ == '__main__':
    main()
