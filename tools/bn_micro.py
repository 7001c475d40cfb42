#!/usr/bin/env python3
"""Micro-benchmark: BatchNorm backward (reduce + finalize + apply) on a layer-sized map.
    python tools/bn_micro.py [--shape 32,144,8,28,28] [--reps 20]"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops

ap = argparse.ArgumentParser()
ap.add_argument('--shapes', default='32,144,8,28,28;32,64,8,28,28;32,64,16,56,56')
ap.add_argument('--reps', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
for sh in a.shapes.split(';'):
    N, C, D, H, W = [int(v) for v in sh.split(',')]
    SP = D * H * W
    x = torch.randn(N, C, D, H, W, device=dev)
    dz = torch.randn_like(x)
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev) * 0.1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros(1, dtype=torch.long, device=dev)
    ss, sq = ops.bn_stats(x, N, C, SP)
    mean, invstd, scale, shift = ops.bn_finalize(ss, sq, N * SP, gamma, beta, 1e-5, 0.1, rm, rv, nbt)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    def run():
        return ops.bn_bwd(dz, None, x, gamma, mean, invstd, 2, N, C, SP, dg, db, scale=scale, shift=shift)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    mb = x.numel() * 4 / 1e6
    print('%s  %.1f MB/tensor  bn_bwd %.3f ms  (5 passes -> %.2f TB/s)' % (sh, mb, ms, 5 * mb / 1e6 / (ms / 1e3)), flush=True)
