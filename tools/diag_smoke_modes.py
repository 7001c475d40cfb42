#!/usr/bin/env python3
"""Per-quantity errors of the smoke() step (tiny MoCo iteration vs the fp64 oracle) under each conv arithmetic mode."""
import importlib, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
pkg = importlib.import_module('video-graph-ssl_amd')
import parity
ops = pkg.engine.ops
ops.AUTOTUNE = False
g = torch.Generator().manual_seed(99)
imgs = [torch.randn(8, 6, 8, 48, 48, generator=g)]
shs = [torch.randperm(8, generator=g)]
parity.register_tiny(pkg)
for mode in ('f32', 'bf16x6', 'bf16x3'):
    ops.set_conv_math(mode)
    rec = parity.run_moco_parity(pkg, torch.device('cuda:0'), 'R2P1D10T', imgs, shs, feat_dim=32, K=20, T=8, with_cpu32=False)[0]
    gm = sorted(rec['grad_hip'].values())
    print(mode, {k: '%.2e' % v for k, v in rec['fwd'].items()}, 'queue %.2e' % rec['post']['queue'],
          'grad median %.2e max %.2e' % (gm[len(gm) // 2], gm[-1]))
