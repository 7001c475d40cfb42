"""One rank of the multi-process MoCo rehearsal (started by tests/test_gpu_dist.py, never collected by
pytest).  Every rank drives the SAME GPU; the collectives run over gloo through host staging
(parallel.DistCtx(host_staged=True)), so the N>1 control flow of MoCoTrainer -- shared-seed permutation,
row all-to-all ShuffleBN exchange, key all-gather, un-shuffle, gradient all-reduce, segment graphs --
is exactly the one RCCL runs in production.

usage: dist_worker.py RANK WORLD PORT OUTDIR USE_GRAPH STEPS [BACKEND]

BACKEND "nccl" (world 1 only on a one-GPU box): the same flow with the production branches -- RCCL all_to_all_single /
all_gather_into_tensor / asynchronous bucketed all_reduce on the process group's stream, no host staging --
DistCtx(force_active=True)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

B, T, HW, K, FEAT = 8, 8, 48, 40, 128      # K % (B*world) != 0 at world 2 -> queue wrap inside a step


def node_batch(step, world):
    g = torch.Generator().manual_seed(1000 + step)
    return torch.randn(world * B, 6, T, HW, HW, generator=g)


def main():
    rank, world, port, outdir, use_graph, steps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
                                                   bool(int(sys.argv[5])), int(sys.argv[6]))
    backend = sys.argv[7] if len(sys.argv) > 7 else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world,
                                device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    pkg = importlib.import_module('video-graph-ssl_amd')
    import parity
    parity.register_tiny(pkg)
    cfg = parity.make_cfg(pkg, 'R2P1D10T', 'moco', FEAT, K, T)
    ctx = pkg.parallel.DistCtx(rank, world, host_staged=backend != 'nccl',
                               force_active=backend == 'nccl' or os.environ.get('GCA_DIST_FORCE_ACTIVE', '0') == '1')
    mode = sys.argv[8] if len(sys.argv) > 8 else 'moco'
    if mode == 'simsiam':
        # SimSiam (configs[3] trainer): per-rank batches, staged backward with bucketed all-reduce, replicas stay identical
        cfg = parity.make_cfg(pkg, 'R2P1D10T', 'simsiam', 32, 16, T)
        tr = pkg.SimSiamTrainer(cfg, 'cuda:0', ctx=ctx, use_graph=use_graph, seed=123 + rank)
        tr.bucket_elems = int(os.environ.get('GCA_BUCKET_ELEMS', '60000'))
        if tr.bucket_elems <= 0:
            tr.bucket_elems = 1 << 40
        out = {}
        for s in range(steps):
            x = node_batch(s, world)[rank * B:(rank + 1) * B].to('cuda:0')
            o = tr.train_step(x)
            torch.cuda.synchronize()
            out['loss%d' % s] = o['loss'].detach().cpu().numpy()
        for k, v in tr.model.state_dict().items():
            out['final/' + k] = v.detach().cpu().numpy()
        out['n_buckets'] = np.array(len(tr._buckets))
        np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **out)
        dist.barrier()
        dist.destroy_process_group()
        return
    # different seeds per rank: the initial broadcast from rank 0 must make the replicas identical
    tr = pkg.MoCoTrainer(cfg, 'cuda:0', ctx=ctx, use_graph=use_graph, seed=123 + rank)
    tr.bucket_elems = int(os.environ.get('GCA_BUCKET_ELEMS', '60000'))     # tiny model: ~8 gradient buckets (0 = one all-reduce)
    if tr.bucket_elems <= 0:
        tr.bucket_elems = 1 << 40
    out = {}
    if rank == 0:
        for k, v in tr.model.state_dict().items():
            out['init/' + k] = v.detach().cpu().numpy()
        out['mem0'] = tr.contrast.memory.detach().cpu().numpy()
    for s in range(steps):
        x = node_batch(s, world)[rank * B:(rank + 1) * B].to('cuda:0')
        o = tr.train_step(x)
        torch.cuda.synchronize()
        out['loss%d' % s] = o['loss'].detach().cpu().numpy()
        out['logits%d' % s] = o['logits'].detach().cpu().numpy()
        out['q%d' % s] = o['q'].detach().cpu().numpy()
    for k, v in tr.model.state_dict().items():
        out['final/' + k] = v.detach().cpu().numpy()
    for k, v in tr.model_ema.state_dict().items():
        out['ema/' + k] = v.detach().cpu().numpy()
    out['mem'] = tr.contrast.memory.detach().cpu().numpy()
    out['index'] = np.array(tr.contrast.index)
    out['ptr_dev'] = tr.ptr_dev.cpu().numpy()
    out['n_buckets'] = np.array(len(tr._buckets))
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
