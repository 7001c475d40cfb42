#!/usr/bin/env python3
"""Time the streaming temporal weight-gradient kernel (conv3d_wgrad_ts.hip, tune_wgrad_tile 11 / 12) against the tuned
conv_wgrad_kernel shapes on the temporal convs of BASELINE configs[1] (R(2+1)D-18, 32 x 16 x 112 x 112).  HIP events on the
launch stream, split-K reduce included."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops
DEV = torch.device('cuda:0')
LAYERS = [('L01', (32, 110, 16, 56, 56), 64, 7, 3), ('L03', (32, 144, 8, 28, 28), 64, 3, 1)]
SPATIAL = [('L02', (32, 64, 8, 28, 28), 144), ('L12', (32, 128, 4, 14, 14), 288), ('L21', (32, 256, 2, 7, 7), 576)]


def time_ms(fn, reps=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ops.set_conv_math(sys.argv[1] if len(sys.argv) > 1 else 'bf16x6')
    ops.load_tune_cache(os.path.join(ROOT, 'profiles', 'tune_cache.json'))
    for name, shape, K, kd, pd in LAYERS:
        N, C, D, H, W = shape
        x = torch.randn(shape, device=DEV)
        plan = ops.ConvPlan(N, C, D, H, W, K, (kd, 1, 1), 1, (pd, 0, 0), DEV)
        dy = torch.randn(plan.out_shape, device=DEV)
        dw = torch.zeros(K, C, kd, 1, 1, device=DEV)
        flops = 2.0 * N * K * D * H * W * C * kd
        plan.tuned = [True, True, False]
        key = [k for k in ops._TUNE_CACHE if k.startswith('v%d%s:2:' % (ops.H.lib.gca_version(), 'c'))
               and k.endswith(','.join(str(v) for v in (N, C, D, H, W, K, kd, 1, 1, 1, 1, 1, pd, 0, 0, 0)))]
        rows = []
        if key:
            hit = ops._TUNE_CACHE[key[0]]
            plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = hit[0], hit[1]
            plan.g.tune_wgrad_math = hit[2] if len(hit) > 2 else 0
            plan.refresh()
            rows.append(('tuned conv_wgrad_kernel %s' % (plan.cfg(2)[:3],), time_ms(lambda: ops._conv_wgrad_launch(plan, x, dy, dw, False))))
        plan.g.tune_wgrad_math = 0
        units = N * (H * W // 16)
        for tile in (11, 12):
            tm = 2 if tile == 12 else 1
            tiles = -(-K // (32 * tm)) * -(-C // 32)
            for nb in (128, 256, 512, 1024, 2048):
                sp = max(1, min(units // 4, -(-nb // tiles)))
                plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = tile, sp
                plan.refresh()
                if plan.cfg(2)[3] & 255 != tile:
                    continue
                rows.append(('ts tile %d splits %d (blocks %d)' % (tile, plan.cfg(2)[2], plan.cfg(2)[2] * tiles),
                             time_ms(lambda: ops._conv_wgrad_launch(plan, x, dy, dw, False))))
        for what, ms in rows:
            print('%s %-52s %8.4f ms %7.1f TF/s' % (name, what, ms, flops / 1e9 / ms))


def spatial():
    for name, shape, K in SPATIAL:
        N, C, D, H, W = shape
        x = torch.randn(shape, device=DEV)
        plan = ops.ConvPlan(N, C, D, H, W, K, (1, 3, 3), 1, (0, 1, 1), DEV)
        dy = torch.randn(plan.out_shape, device=DEV)
        dw = torch.zeros(K, C, 1, 3, 3, device=DEV)
        flops = 2.0 * N * K * D * H * W * C * 9
        plan.tuned = [True, True, True]
        rows = []
        for tile, sp in ((6, 102), (6, 28), (5, 9), (4, 16)):
            plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = tile, sp
            plan.refresh()
            rows.append(('conv_wgrad_kernel %s' % (plan.cfg(2)[:3],), time_ms(lambda: ops._conv_wgrad_launch(plan, x, dy, dw, False))))
        units = N * D * -(-W // 16)
        tiles = -(-K // 32) * -(-C // 32)
        for nb in (128, 256, 512, 768, 1024):
            sp = max(1, min(units // 4, nb // tiles))
            plan.g.tune_wgrad_tile, plan.g.tune_wgrad_splits = 13, sp
            plan.refresh()
            if plan.cfg(2)[3] & 255 != 13:
                continue
            rows.append(('ss tile 13 splits %d (blocks %d)' % (plan.cfg(2)[2], plan.cfg(2)[2] * tiles),
                         time_ms(lambda: ops._conv_wgrad_launch(plan, x, dy, dw, False))))
        for what, ms in rows:
            print('%s %-52s %8.4f ms %7.1f TF/s' % (name, what, ms, flops / 1e9 / ms))


if __name__ == '__main__':
    spatial()
    main()
