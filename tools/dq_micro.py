import importlib, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
pkg = importlib.import_module('video-graph-ssl_amd')
ops = pkg.engine.ops
b = 32
for K in (4096, 65536):
    q = torch.nn.functional.normalize(torch.randn(b, 128, device='cuda'))
    k = torch.nn.functional.normalize(torch.randn(b, 128, device='cuda'))
    mem = torch.nn.functional.normalize(torch.randn(K, 128, device='cuda'))
    logits, lse, rank, loss = ops.moco_logits_fwd(q, k, mem, 1 / 0.07, want_lse=True, want_rank=True, want_loss=True)
    t_f = bench.ev_time_ms(lambda: ops.moco_logits_fwd(q, k, mem, 1 / 0.07, want_lse=True, want_rank=True, want_loss=True), 20, 3)
    t_b = bench.ev_time_ms(lambda: ops.moco_logits_bwd(k, mem, 1 / 0.07, logits=logits, lse=lse), 20, 3)
    allk = torch.randn(256, 128, device='cuda')
    ptr = torch.zeros(1, dtype=torch.long, device='cuda')
    t_e = bench.ev_time_ms(lambda: ops.queue_enqueue(mem, allk, 0, save=True, ptr_dev=ptr), 20, 3)
    print('K=%d fwd %.4f ms  dq %.4f ms  enqueue(256 rows) %.4f ms' % (K, t_f, t_b, t_e), flush=True)
