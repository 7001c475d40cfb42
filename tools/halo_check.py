#!/usr/bin/env python3
"""Every conv layer of an encoder at benchmark size: the LDS-halo kernels (heuristic box, and every candidate the tuner
would measure) against the gather kernels on the same operands -- forward, statistics, dgrad -- with timings.

    python tools/halo_check.py [--backbone R2P1D18] [--batch 32] [--math bf16x6] [--cands]
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('GCA_AUTOTUNE', '0')


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', default='R2P1D18')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--size', type=int, default=112)
    ap.add_argument('--math', default='bf16x6')
    ap.add_argument('--cands', action='store_true', help='time every halo candidate of the tuner')
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module('video-graph-ssl_amd')
    ops = pkg.engine.ops
    ops.set_conv_math(args.math)
    bb = pkg.lib.modeling.backbone.backbone_3d
    torch.manual_seed(0)
    enc = getattr(bb, args.backbone)().cuda()
    enc.fc = pkg.engine.layers.HipIdentity()
    layers = bench.conv_layers_of(enc, (args.batch, 3, args.frames, args.size, args.size), pkg)
    dev = torch.device('cuda:0')
    worst = 0.0
    for i, (m, shp, xs) in enumerate(layers):
        plan = ops.ConvPlan(*shp, m.out_channels, m.kernel_size, m.stride, m.padding, dev)
        plan.tuned = [True, True, True]
        x = torch.randn(shp, device=dev)
        w = m.weight.data
        dy = torch.randn(plan.out_shape, device=dev)
        base = torch.randn(shp, device=dev)
        flops = 2.0 * dy.numel() * shp[1] * plan.taps

        def run(which):
            if which == 0:
                wp = ops.conv_pack(plan, 0, w)
                y, (ss, sq) = ops.conv_fwd(plan, x, wp, None, stats=True)
                t = bench.ev_time_ms(lambda: ops.conv_fwd(plan, x, wp, None, stats=True), 5, 1)
                return (y, ss.sum(1), sq.sum(1)), t
            wp = ops.conv_pack(plan, 1, w)
            dx = ops.conv_dgrad(plan, dy, wp)
            acc = base.clone()
            ops.conv_dgrad(plan, dy, wp, acc, accumulate=True)
            t = bench.ev_time_ms(lambda: ops.conv_dgrad(plan, dy, wp), 5, 1)
            return (dx, acc - base), t
        for which in ((0, 1) if i > 0 else (0,)):
            name = ('fwd', 'dgrad')[which]
            cfg_h = plan.cfg(which)
            halo_default = (cfg_h[3] >> 14) & 1
            out_h, t_h = run(which)
            # the gather kernels on the same problem (tile code without the halo flag = conv3d.hip)
            setattr(plan.g, 'tune_%s_bm' % name, 64)
            plan.refresh()
            out_g, t_g = run(which)
            setattr(plan.g, 'tune_%s_bm' % name, 0)
            plan.refresh()
            errs = [rel(a, b) for a, b in zip(out_h, out_g)]
            worst = max(worst, max(errs)) if halo_default else worst
            line = 'L%02d %-5s in%-24s K=%-4d k=%s s=%s %7.2f GF | default %s %s %.3f ms %6.1f TF | gather<64> %.3f ms %6.1f TF | diff %s' % (
                i, name, shp, m.out_channels, m.kernel_size, m.stride, flops / 1e9, 'HALO' if halo_default else 'gath',
                cfg_h[:3], t_h, flops / 1e9 / t_h, t_g, flops / 1e9 / t_g, ' '.join('%.1e' % e for e in errs))
            print(line, flush=True)
            if args.cands:
                M = m.out_channels if which == 0 else shp[1]
                for c in plan._halo_candidates(which, M):
                    setattr(plan.g, 'tune_%s_bm' % name, c[0])
                    setattr(plan.g, 'tune_%s_box' % name, c[2])
                    plan.refresh()
                    if (plan.cfg(which)[3] >> 14) & 1:
                        out_c, t_c = run(which)
                        e = max(rel(a, b) for a, b in zip(out_c, out_g))
                        print('      halo rows %3d box %s: %.3f ms %6.1f TF  diff %.1e' % (
                            c[0] & 1023, (c[2] & 255, (c[2] >> 8) & 255, c[2] >> 16), t_c, flops / 1e9 / t_c, e), flush=True)
                setattr(plan.g, 'tune_%s_bm' % name, 0)
                setattr(plan.g, 'tune_%s_box' % name, 0)
                plan.refresh()
        del x, dy
    print('worst halo-vs-gather difference: %.2e' % worst)


if __name__ == '__main__':
    main()
