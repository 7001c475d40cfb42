#!/usr/bin/env python3
"""Is the fp16-storage path's distance from fp32 what fp16 rounding of the feature maps explains?

One MoCo iteration of a 3D-ResNet, same weights / clips, three ways on the HIP path:
   A  fp32 storage (f32 MFMA arithmetic)
   B  fp16 storage
   C  fp32 storage with every conv INPUT perturbed by relative noise 2^-11 u, u ~ U(-.5,.5)... (approximated by perturbing
      the clip only: --perturb, default 2^-12) -- the conditioning reference: what a single storage-rounding-sized
      perturbation at the input does to the same quantities.
Prints feature / loss differences, the per-tensor gradient error distribution of B and C against A, and (--layerwise) the
gradient at every BatchNorm in backward order for B against A: rounding grows gradually, a broken kernel jumps.

    python tools/f16_model_check.py [--backbone R3D18] [--batch 8] [--frames 16] [--size 112] [--layerwise]
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
    ap.add_argument('--backbone', default='R3D18')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--size', type=int, default=112)
    ap.add_argument('--perturb', type=float, default=2.0 ** -12)
    ap.add_argument('--layerwise', action='store_true')
    ap.add_argument('--fixed-dq', action='store_true', help='backward from one fixed d(loss)/d(features) instead of the InfoNCE gradient (which, at initialisation, is itself ill-conditioned in the features)')
    args = ap.parse_args()
    import parity
    pkg = importlib.import_module('video-graph-ssl_amd')
    ops = pkg.engine.ops
    bb = pkg.lib.modeling.backbone.backbone_3d
    depth = int(args.backbone[3:])
    bb.register('R3DX', lambda: getattr(bb.resnet, 'resnet%d' % depth)(sample_size=args.size, sample_duration=args.frames))
    dev = torch.device('cuda:0')
    cfg = parity.make_cfg(pkg, 'R3DX', 'moco', 128, 4096, args.frames)
    torch.manual_seed(3)
    images = torch.randn(args.batch, 6, args.frames, args.size, args.size).to(dev)
    sh = torch.randperm(args.batch)
    res = {}
    for tag, mode, pert in (('A fp32', 'f32', 0.0), ('B fp16-storage', 'fp16', 0.0), ('C fp32 perturbed', 'f32', args.perturb)):
        ops._conv_plan.cache_clear()
        ops.set_conv_math(mode)
        tr = pkg.MoCoTrainer(cfg, dev, use_graph=False, seed=1)
        x_in = images
        if pert:
            x_in = images * (1.0 + pert * (torch.rand(images.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) - 0.5) * 2)
        if args.layerwise:
            pkg.engine.layers.DEBUG_GRADS = {}
        if args.fixed_dq:
            from importlib import import_module
            T = import_module('video-graph-ssl_amd.engine.tape')
            tape = T.Tape(True)
            tr.optimizer.zero_grad()
            qv = tr.model.fwd(tape, T.Var(torch.chunk(x_in, 2, dim=1)[0]))
            dq = torch.randn(qv.t.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(9)) / args.batch
            qv.grad = dq * tr.loss_scale
            tape.backward()
            if tr.loss_scale != 1.0:
                ops.scale_(tr.arena_q.grad, 1.0 / tr.loss_scale)
            out = dict(loss=(qv.t * dq).sum(), q=qv.t)
        else:
            out = tr.train_step(x_in, shuffle_ids=sh)
        torch.cuda.synchronize()
        a = tr.arena_q
        grads = {n: a.grad[o:o + s].detach().clone() for n, o, s in zip(a.names, a.offsets, a.sizes)}
        dbg = None
        if args.layerwise:
            names = {id(m): n for n, m in tr.model.named_modules()}
            ls = tr.loss_scale                  # activation gradients of the fp16 path carry the static loss scale
            dbg = [(names.get(k, '?'), v[0].float() / ls, v[1].float() / ls) for k, v in pkg.engine.layers.DEBUG_GRADS.items()]
            pkg.engine.layers.DEBUG_GRADS = None
        res[tag] = dict(loss=float(out['loss']), q=out['q'].detach().clone(), grads=grads, dbg=dbg)
        print('%-18s loss %.6f' % (tag, res[tag]['loss']), flush=True)
        tr.close()
        del tr
    ops.set_conv_math('bf16x6')
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / y.double().abs().max().clamp_min(1e-30))
    A = res['A fp32']
    for tag in ('B fp16-storage', 'C fp32 perturbed'):
        R = res[tag]
        errs = sorted(((rel(R['grads'][n], g), n) for n, g in A['grads'].items() if float(g.abs().max()) > 0), reverse=True)
        v = [e for e, _ in errs]
        print('%s vs A: q %.2e  loss %.2e | gradients per tensor: median %.2e p90 %.2e max %.2e (%d tensors)' % (
            tag, rel(R['q'], A['q']), abs(R['loss'] - A['loss']) / abs(A['loss']), v[len(v) // 2], v[len(v) // 10], v[0], len(v)))
        for e, n in errs[:5]:
            print('      %.3e  %s' % (e, n))
    if args.layerwise:
        l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
        for tag in ('B fp16-storage', 'C fp32 perturbed'):
            print('backward order, %s vs A: d(BN out) L2 | d(conv out) L2' % tag)
            for (n1, dz1, dy1), (n2, dz2, dy2) in zip(A['dbg'], res[tag]['dbg']):
                print('   %-40s %.2e | %.2e' % (n1[-40:], l2(dz2, dz1), l2(dy2, dy1)))


if __name__ == '__main__':
    main()
