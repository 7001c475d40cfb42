#!/usr/bin/env python3
"""Time every (tile, split) launch configuration of selected layers' forward / dgrad kernels."""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GCA_AUTOTUNE'] = '0'
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_micro import LAYERS, ev

ap = argparse.ArgumentParser()
ap.add_argument('--layers', default='L01,L03,L14')
ap.add_argument('--splits', default='1')
ap.add_argument('--tails', default='', help='two-phase: slots guesses, e.g. 512,768,1024,1280 (tail rows 32 and 64 tried)')
ap.add_argument('--tiles', default='32,64,96,128,160,1088,1152')
a = ap.parse_args()
dev = torch.device('cuda:0')
for name in a.layers.split(','):
    C, D, H, W, K, k, s, p = LAYERS[name]
    shp = (32, C, D, H, W)
    plan = ops.ConvPlan(32, C, D, H, W, K, k, s, p, dev)
    N, _, OD, OH, OW = plan.out_shape
    fl = 2.0 * N * K * OD * OH * OW * C * k[0] * k[1] * k[2]
    x = torch.randn(shp, device=dev); dy = torch.randn(plan.out_shape, device=dev)
    w = torch.randn((K, C) + k, device=dev) * 0.05
    wp0, wp1 = ops.conv_pack(plan, 0, w), ops.conv_pack(plan, 1, w)
    dx = torch.empty(shp, device=dev)
    for bm in [int(v) for v in a.tiles.split(',')]:
        for sp in [int(v) for v in a.splits.split(',')]:
            plan.g.tune_fwd_bm, plan.g.tune_fwd_splits = bm, sp
            plan.g.tune_dgrad_bm, plan.g.tune_dgrad_splits = bm, sp
            plan.refresh()
            try:
                tf = ev(lambda: ops.conv_fwd(plan, x, wp0, None, stats=True), 5)
                td = ev(lambda: ops.conv_dgrad(plan, dy, wp1, dx, False), 5)
            except RuntimeError as e:
                print(name, bm, sp, 'ERR', e); continue
            print('%s tile %3d split %2d  cfg f%s d%s | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF' %
                  (name, bm, sp, plan.cfg(0), plan.cfg(1), tf, fl / 1e9 / tf, td, fl / 1e9 / td), flush=True)
            if a.tails and sp == 1 and bm < 1024 and bm > 32:
                for which, M, Nt in ((0, K, N * OD * OH * OW), (1, C, 32 * D * H * W)):
                    tilesM, tilesN = -(-M // bm), -(-Nt // 128)
                    for slots in [int(v) for v in a.tails.split(',')]:
                        mc = (tilesM * tilesN // slots) * slots // tilesM
                        if mc <= 0 or mc >= tilesN:
                            continue
                        for tr in (32, 64):
                            if tr >= bm:
                                continue
                            code = (tr // 32) | (mc << 8)
                            if which == 0:
                                plan.g.tune_fwd_tail = code
                            else:
                                plan.g.tune_dgrad_tail = code
                            plan.refresh()
                            t = ev(lambda: ops.conv_fwd(plan, x, wp0, None, stats=True), 5) if which == 0 else ev(lambda: ops.conv_dgrad(plan, dy, wp1, dx, False), 5)
                            print('      %s two-phase: slots %4d main cols %5d/%d tail rows %2d  %7.3f ms %6.1f TF' %
                                  ('fwd  ' if which == 0 else 'dgrad', slots, mc, tilesN, tr, t, fl / 1e9 / t), flush=True)
                    plan.g.tune_fwd_tail = plan.g.tune_dgrad_tail = 0
                    plan.refresh()
