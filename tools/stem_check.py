#!/usr/bin/env python3
"""Stem convs at benchmark size: the stem kernel (conv3d_stem.hip) against the gather kernels, timings and differences.

    python tools/stem_check.py [--math bf16x6|bf16x3|fp16]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--math', default='bf16x6')
    ap.add_argument('--only', default='', help='run only the cases whose name contains this, stem kernel only (profiling)')
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module('video-graph-ssl_amd')
    ops = pkg.engine.ops
    half = args.math == 'fp16'
    ops.set_conv_math(args.math)
    dev = torch.device('cuda:0')
    cases = [('R(2+1)D-18 stem b32', (32, 3, 16, 112, 112), 110, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
             ('R(2+1)D K=45 b32', (32, 3, 16, 112, 112), 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
             ('S3D stem b4 224', (4, 3, 16, 224, 224), 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
             ('R3D stem b16 32x224', (16, 3, 32, 224, 224), 64, (7, 7, 7), (1, 2, 2), (3, 3, 3))]
    for name, shape, K, k, s, p in cases:
        if args.only and args.only not in name:
            continue
        plan = ops.ConvPlan(*shape, K, k, s, p, dev, act_f16=half)
        plan.tuned = [True, True, True]
        x = torch.randn(shape, device=dev).to(plan.act_dtype)
        w = torch.randn((K, 3) + k, device=dev) * 0.05
        flops = 2.0 * torch.tensor(plan.out_shape).prod().item() * 3 * k[0] * k[1] * k[2]
        res = {}
        for label, code in ((('stem', 4096 | 64),) if args.only else (('stem', 4096 | 64), ('gather64', 64), ('gather128', 128))):
            plan.g.tune_fwd_bm = code
            plan.refresh()
            wp = ops.conv_pack(plan, 0, w)
            y, (ss, sq) = ops.conv_fwd(plan, x, wp, None, stats=True)
            t = bench.ev_time_ms(lambda: ops.conv_fwd(plan, x, wp, None, stats=True), 5, 2)
            res[label] = (y.float(), t, plan.cfg(0))
        d = float((res['stem'][0] - res['gather64'][0]).abs().max() / res['gather64'][0].abs().max()) if 'gather64' in res else 0.0
        print('%-22s %7.1f GF | ' % (name, flops / 1e9) + ' | '.join('%s %.3f ms %6.1f TF cfg %s' % (l, r[1], flops / 1e9 / r[1], r[2]) for l, r in res.items()) + ' | diff %.1e' % d, flush=True)


if __name__ == '__main__':
    main()
