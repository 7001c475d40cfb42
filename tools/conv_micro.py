#!/usr/bin/env python3
"""Micro-benchmark / profiling target: run selected R(2+1)D-18 conv layers (fwd / dgrad / wgrad) alone.
    python tools/conv_micro.py [--layers L01,L02,L12,L30] [--what fwd,dgrad,wgrad] [--reps 5] [--batch 32]
Prints ms and TFLOP/s per (layer, pass); under rocprofv3 --pmc it is the per-kernel counter target."""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops

# (Cin, D, H, W, K, kernel, stride, pad) at batch 1 of R(2+1)D-18, 16 x 112 x 112
LAYERS = {
    'L00': (3, 16, 112, 112, 110, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
    'L01': (110, 16, 56, 56, 64, (7, 1, 1), (1, 1, 1), (3, 0, 0)),
    'L02': (64, 8, 28, 28, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'L03': (144, 8, 28, 28, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'L10': (64, 8, 28, 28, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    'L11': (230, 8, 14, 14, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),
    'L12': (128, 4, 14, 14, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'L14': (288, 4, 14, 14, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'L19': (128, 4, 14, 14, 460, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    'L21': (256, 2, 7, 7, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'L23': (576, 2, 7, 7, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    'L28': (256, 2, 7, 7, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    'L30': (512, 1, 4, 4, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    'L32': (1152, 1, 4, 4, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
}


def ev(fn, reps):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='L00,L01,L02,L03,L12,L14,L21,L23,L30,L32')
    ap.add_argument('--what', default='fwd,dgrad,wgrad')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    tot = {}
    for name in a.layers.split(','):
        C, D, H, W, K, k, s, p = LAYERS[name]
        shp = (a.batch, C, D, H, W)
        plan = ops.conv_plan(shp, K, k, s, p, dev)
        N, _, OD, OH, OW = plan.out_shape
        fl = 2.0 * N * K * OD * OH * OW * C * k[0] * k[1] * k[2]
        x = torch.randn(shp, device=dev)
        dy = torch.randn(plan.out_shape, device=dev)
        w = torch.randn((K, C) + k, device=dev) * 0.05
        wp0, wp1 = ops.conv_pack(plan, 0, w), ops.conv_pack(plan, 1, w)
        dw, dx = torch.zeros_like(w), torch.empty(shp, device=dev)
        line = '%s in%-24s K=%-4d k=%s s=%s GF %7.2f |' % (name, shp, K, k, s, fl / 1e9)
        for what in a.what.split(','):
            if what == 'fwd':
                t = ev(lambda: ops.conv_fwd(plan, x, wp0, None, stats=True), a.reps)
            elif what == 'dgrad':
                t = ev(lambda: ops.conv_dgrad(plan, dy, wp1, dx, False), a.reps)
            else:
                t = ev(lambda: ops.conv_wgrad(plan, x, dy, dw, True), a.reps)
            line += ' %s %7.3f ms %6.1f TF |' % (what, t, fl / 1e9 / t)
            tot[what] = tot.get(what, 0.0) + t
        print(line, flush=True)
    print('total ms:', {k: round(v, 3) for k, v in tot.items()})


if __name__ == '__main__':
    main()
