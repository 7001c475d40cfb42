#!/usr/bin/env python3
"""Every conv layer of an encoder at benchmark size: forward (+ BatchNorm statistics), dgrad (plain and accumulating) and
wgrad under the launch configuration the AUTOTUNER pins (measured on the spot, or read from GCA_TUNE_CACHE) against the
un-tuned gather-kernel configuration, on the same operands.  Prints the pinned configuration and the differences.

    python tools/tuned_check.py [--backbone R2P1D18] [--batch 32] [--math f32]
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['GCA_HALO'] = '0'          # the un-tuned reference runs on the gather kernels


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', default='R2P1D18')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--size', type=int, default=112)
    ap.add_argument('--math', default='f32')
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module('video-graph-ssl_amd')
    ops = pkg.engine.ops
    ops.set_conv_math(args.math)
    bb = pkg.lib.modeling.backbone.backbone_3d
    torch.manual_seed(0)
    enc = getattr(bb, args.backbone)().cuda()
    enc.fc = pkg.engine.layers.HipIdentity()
    ops.AUTOTUNE = False
    layers = bench.conv_layers_of(enc, (args.batch, 3, args.frames, args.size, args.size), pkg)
    dev = torch.device('cuda:0')
    worst = 0.0
    for i, (m, shp, xs) in enumerate(layers):
        x = torch.randn(shp, device=dev)
        w = m.weight.data
        outs = []
        for tuned in (False, True):
            ops.AUTOTUNE = tuned
            plan = ops.ConvPlan(*shp, m.out_channels, m.kernel_size, m.stride, m.padding, dev)
            if not tuned:
                dy = torch.randn(plan.out_shape, device=dev)
            wp0, wp1 = ops.conv_pack(plan, 0, w), ops.conv_pack(plan, 1, w)
            y, (ss, sq) = ops.conv_fwd(plan, x, wp0, None, stats=True, w_raw=w)
            o = [y, ss.sum(1), sq.sum(1)]
            if i > 0:
                dx = ops.conv_dgrad(plan, dy, wp1, w_raw=w)
                acc = torch.ones_like(dx)
                ops.conv_dgrad(plan, dy, wp1, acc, accumulate=True, w_raw=w)
                o += [dx, acc - 1]
            dw = torch.zeros_like(w)
            ops.conv_wgrad(plan, x, dy, dw, accumulate=True)
            o.append(dw)
            outs.append(o)
            if tuned:
                g = plan.g
                desc = 'fwd bm %d sp %d tail %d math %d box %06x | dgrad bm %d sp %d tail %d math %d box %06x | wgrad tile %d sp %d' % (
                    g.tune_fwd_bm, g.tune_fwd_splits, g.tune_fwd_tail, g.tune_fwd_math, g.tune_fwd_box, g.tune_dgrad_bm,
                    g.tune_dgrad_splits, g.tune_dgrad_tail, g.tune_dgrad_math, g.tune_dgrad_box, g.tune_wgrad_tile, g.tune_wgrad_splits)
        errs = [rel(a, b) for a, b in zip(outs[1], outs[0])]
        worst = max(worst, max(errs))
        print('L%02d in%-24s K=%-4d k=%s s=%s | %s | diffs %s' % (i, shp, m.out_channels, m.kernel_size, m.stride, desc,
                                                                 ' '.join('%.1e' % e for e in errs)), flush=True)
    print('worst tuned-vs-heuristic difference: %.2e' % worst)


if __name__ == '__main__':
    main()
