#!/usr/bin/env python3
"""Gradients of one configs[1] MoCo iteration under the TUNED launch configuration (profiles/tune_cache.json + autotuner on:
two-phase tiles, split-K factors, LDS-halo boxes, per-pass arithmetic pins) against the same iteration under the heuristic
configuration the test suite runs on -- same weights, same clips, same arithmetic mode.  Any per-tensor difference beyond
rounding (+ the odd ReLU flip) would be a kernel that is wrong in one launch shape.

    python tools/grad_tuned_vs_heuristic.py [--math f32] [--batch 32]
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['GCA_AUTOTUNE'] = '0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--math', default='f32')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--perturb', type=float, default=0.0, help='relative noise added to the clips of the SECOND run')
    ap.add_argument('--perturb-queue', type=float, default=0.0, help='relative noise on the queue rows of the SECOND run: the encoder forward (and so every backward operator) stays bit-identical, only dL/dq moves')
    ap.add_argument('--warm-queue', action='store_true', help='queue rows near the key features (training regime) instead of random unit vectors')
    ap.add_argument('--pre-steps', type=int, default=0, help='train this many iterations first (heuristic shapes) and compare from THAT state')
    ap.add_argument('--layerwise', action='store_true', help='compare the gradient at every BatchNorm (d out, d conv-out) between the two runs, in backward order')
    ap.add_argument('--save', default='')
    ap.add_argument('--against', default='')
    ap.add_argument('--modes', default='heuristic,tuned', help='two of heuristic,tuned (GCA_HALO=0 in the environment turns the halo heuristic off)')
    ap.add_argument('--cache', default=os.path.join(ROOT, 'profiles', 'tune_cache.json'))
    args = ap.parse_args()
    import parity
    pkg = importlib.import_module('video-graph-ssl_amd')
    ops = pkg.engine.ops
    ops.set_conv_math(args.math)
    dev = torch.device('cuda:0')
    cfg = parity.make_cfg(pkg, 'R2P1D18', 'moco', 128, 4096, 16)
    torch.manual_seed(3)
    images = torch.randn(args.batch, 6, 16, 112, 112).to(dev)
    sh = torch.randperm(args.batch)
    res = {}
    state = None
    if args.pre_steps:
        ops.AUTOTUNE = False
        tr = pkg.MoCoTrainer(cfg, dev, use_graph=True, seed=1)
        g = torch.Generator().manual_seed(17)
        for i in range(args.pre_steps):
            out = tr.train_step(torch.randn(args.batch, 6, 16, 112, 112, generator=g).to(dev))
        print('pre-trained %d steps, loss %.4f' % (args.pre_steps, float(out['loss'])), flush=True)
        state = tr.state_dict()
        tr.close()
        del tr
    for idx, mode in enumerate(args.modes.split(',')):
        ops._conv_plan.cache_clear()
        ops.AUTOTUNE = mode == 'tuned'
        if mode == 'tuned' and os.path.exists(args.cache):
            ops.load_tune_cache(args.cache)
        tr = pkg.MoCoTrainer(cfg, dev, use_graph=False, seed=1)
        if state is not None:
            tr.load_state_dict(state)
        if args.warm_queue:
            g = torch.Generator().manual_seed(11)
            kmean = torch.nn.functional.normalize(torch.randn(1, 128, generator=g))
            tr.contrast.memory.copy_(torch.nn.functional.normalize(kmean + 0.03 * torch.randn(4096, 128, generator=g)))
        if args.perturb_queue and len(res) == 1:
            m_ = tr.contrast.memory
            m_.mul_(1.0 + args.perturb_queue * torch.randn(m_.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7)))
        x_in = images
        if args.perturb and len(res) == 1:
            x_in = images * (1.0 + args.perturb * torch.randn(images.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)))
        if args.layerwise:
            pkg.engine.layers.DEBUG_GRADS = {}
        out = tr.train_step(x_in, shuffle_ids=sh)
        torch.cuda.synchronize()
        a = tr.arena_q
        grads = {n: a.grad[o:o + s].detach().clone() for n, o, s in zip(a.names, a.offsets, a.sizes)}
        halo = sum(1 for m in tr.model.modules() if isinstance(m, pkg.engine.layers.HipConv3d) and m._pack_plan[0] is not None
                   and (m._pack_plan[0].cfg(0)[3] >> 14) & 1)
        dbg = None
        if args.layerwise:
            names = {id(m): n for n, m in tr.model.named_modules()}
            dbg = [(names.get(k, '?'), v[0], v[1]) for k, v in pkg.engine.layers.DEBUG_GRADS.items()]     # insertion = backward order
            pkg.engine.layers.DEBUG_GRADS = None
        res[idx] = dict(dbg=dbg, loss=float(out['loss']), q=out['q'].detach().clone(), grads=grads, halo_fwd_layers=halo)
        print('%s: loss %.6f, %d conv layers forward on the LDS-halo kernels' % (mode, res[idx]['loss'], halo), flush=True)
        tr.close()
        del tr
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / y.double().abs().max().clamp_min(1e-30))
    A, B = 0, 1
    if args.layerwise:
        print('backward order: d(BN out) diff, d(conv out) diff')
        for (n1, dz1, dy1), (n2, dz2, dy2) in zip(res[0]['dbg'], res[1]['dbg']):
            l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
            frac = float(((dz2 - dz1).abs() > 1e-2 * dz1.abs().max()).float().mean())
            print('   %-46s dz max %.2e L2 %.2e frac>1%% %.1e | dy max %.2e L2 %.2e' % (n1[26:], rel(dz2, dz1), l2(dz2, dz1), frac, rel(dy2, dy1), l2(dy2, dy1)))
    print('q rel diff %.2e' % rel(res[B]['q'], res[A]['q']))
    errs = sorted(((rel(res[B]['grads'][n], g), n) for n, g in res[A]['grads'].items() if float(g.abs().max()) > 0),
                  reverse=True)
    vals = [e for e, _ in errs]
    if args.save:
        torch.save({n: g.cpu() for n, g in res[A]['grads'].items()}, args.save)
    if args.against:
        ref = torch.load(args.against)
        e2 = sorted(((rel(res[A]['grads'][n], g.cuda()), n) for n, g in ref.items() if float(g.abs().max()) > 0), reverse=True)
        v2 = [e for e, _ in e2]
        print('vs saved gradients %s: median %.2e p90 %.2e max %.2e' % (args.against, v2[len(v2) // 2], v2[len(v2) // 10], v2[0]))
        for e, n in e2[:6]:
            print('   %.3e  %s' % (e, n))
    print('per-tensor gradient difference, run 2 (%s) vs run 1 (%s):' % tuple(reversed(args.modes.split(','))))
    print('per-tensor gradient difference: median %.2e  p90 %.2e  max %.2e  (%d tensors)'
          % (vals[len(vals) // 2], vals[len(vals) // 10], vals[0], len(vals)))
    for e, n in errs[:12]:
        print('   %.3e  %s' % (e, n))


if __name__ == '__main__':
    main()
