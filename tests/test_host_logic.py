"""CPU-only tests of the host side: the C ABI library loads and exports every symbol include/gca_hip.h
declares, the product modules reproduce the reference's parameter trees / initial weights (via the oracle,
which make_golden.py proved seed-equivalent to the reference), config loading, LR schedule, flat arenas,
and the 2-rank (gloo) data-parallel exchange logic.  No kernel is launched here."""
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, 'include', 'gca_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(gca_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) > 40
    bound = set(pkg._hip.SIGNATURES)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    for name in declared:
        assert getattr(pkg._hip.lib, name) is not None
    assert pkg._hip.lib.gca_version() >= 4


def test_product_path_refuses_cpu_tensors(pkg):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pkg.engine.ops.relu_fwd(torch.randn(4))


def test_geometry_validation_without_gpu(pkg):
    H = pkg._hip
    import ctypes as C
    good = H.ConvGeom(2, 3, 4, 8, 8, 5, 1, 3, 3, 1, 1, 1, 0, 1, 1, 4, 8, 8, 0)
    assert H.lib.gca_conv_pack_elems(C.byref(good), 0) == (32 + 128) * 32   # K = 27 -> 32; M = 5 -> 32 rows + 128 zero rows (tallest tile may overhang)
    assert H.lib.gca_conv_table_rows(C.byref(good), 1) == 48 + 32          # 5 * 9 = 45 -> 48 rows + tap-delta table
    bad = H.ConvGeom(2, 3, 4, 8, 8, 5, 1, 3, 3, 1, 1, 1, 0, 1, 1, 4, 8, 7, 0)   # wrong OW
    assert H.lib.gca_conv_pack_elems(C.byref(bad), 0) == -1
    rows = H.lib.gca_conv_table_rows(C.byref(good), 0)
    tab = torch.zeros(rows * 2, dtype=torch.int32)
    assert H.lib.gca_conv_table_build_host(C.byref(good), 0, tab.data_ptr()) == 0
    t = tab.view(rows, 2)
    # row k = (c, kh, kw): BYTE offset 4*(c*D*H*W + kh*W + kw) ; padded rows are invalid with tap id 63
    assert int(t[0, 0]) == 0 and int(t[4, 0]) == 4 * (1 * 8 + 1) and int(t[9, 0]) == 4 * (4 * 8 * 8)
    # meta word: tap id in bits 0-5, valid in bit 6, (dd, dh, dw) signed bytes above
    assert (int(t[26, 1]) >> 6) & 1 == 1 and int(t[26, 1]) & 63 == 8 and (int(t[27, 1]) >> 6) & 1 == 0 and int(t[27, 1]) & 63 == 63


def test_conv_arithmetic_mode_is_an_accuracy_floor_host_side(pkg):
    """gca_set_conv_math / tune_*_math (host logic only, no launch): the launch-configuration queries report the arithmetic
    a pass would run with -- the mode in force, or a pinned one when that is at least as accurate (f32 > bf16x6 > bf16x3)."""
    H = pkg._hip
    import ctypes as C
    g = H.ConvGeom(2, 16, 4, 8, 8, 40, 3, 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 8, 0)
    out = (C.c_int32 * 4)()
    math_of = lambda which: (H.lib.gca_conv_kernel_cfg(C.byref(g), which, out), (out[3] >> 12) & 3)[1] if which < 2 else \
        (H.lib.gca_conv_wgrad_cfg(C.byref(g), out), (out[3] >> 12) & 3)[1]
    default = H.lib.gca_get_conv_math()
    try:
        assert H.lib.gca_set_conv_math(3) == -1 and H.lib.gca_set_conv_math(-1) == -1
        for mode in (0, 1, 2):
            assert H.lib.gca_set_conv_math(mode) == 0 and H.lib.gca_get_conv_math() == mode
            g.tune_fwd_math = g.tune_dgrad_math = g.tune_wgrad_math = 0
            assert [math_of(w) for w in (0, 1, 2)] == [mode] * 3
            for pin in (0, 1, 2):
                g.tune_fwd_math = g.tune_dgrad_math = g.tune_wgrad_math = 1 + pin
                rank = {0: 2, 2: 1, 1: 0}
                want = pin if rank[pin] >= rank[mode] else mode
                assert [math_of(w) for w in (0, 1, 2)] == [want] * 3, (mode, pin)
        g.tune_fwd_math = 4
        assert H.lib.gca_conv_kernel_cfg(C.byref(g), 0, out) == -1          # out of range: invalid geometry
    finally:
        H.lib.gca_set_conv_math(default)


def test_weight_gradient_kernel_selection_host_side(pkg):
    """tune_wgrad_tile (host logic only, no launch): gca_conv_wgrad_cfg reports the kernel a weight gradient would run on --
    the streaming temporal / (1,3,3) kernels (11, 12, 13) and the stem kernel (14) only where geometry and arithmetic admit
    them, the gather kernel's shape otherwise; split counts follow the unit counts of each kernel; the work-space size
    follows the split count."""
    H = pkg._hip
    import ctypes as C
    out = (C.c_int32 * 4)()

    def geom(N, Cin, D, Hh, W, K, k, s, p, f16=0):
        od, oh, ow = [(d + 2 * pp - kk) // ss + 1 for d, pp, kk, ss in zip((D, Hh, W), p, k, s)]
        g = H.ConvGeom(N, Cin, D, Hh, W, K, *k, *s, *p, od, oh, ow, 0)
        g.act_f16 = f16
        return g

    def tile(g, t, sp=0):
        g.tune_wgrad_tile, g.tune_wgrad_splits = t, sp
        assert H.lib.gca_conv_wgrad_cfg(C.byref(g), out) == 0
        return out[3] & 255, out[2]

    default = H.lib.gca_get_conv_math()
    try:
        H.lib.gca_set_conv_math(2)                                              # bf16x6
        stem = geom(32, 3, 16, 112, 112, 110, (1, 7, 7), (1, 2, 2), (0, 3, 3))
        assert tile(stem, 14, 256) == (14, 256)
        assert tile(stem, 14, 1024) == (14, 1024)                               # 512 (clip, od) units x 2 row chunks
        assert tile(stem, 14, 0)[0] == 14 and 1 <= tile(stem, 14, 0)[1] <= 1024
        ws = [H.lib.gca_conv_wgrad_ws_bytes(C.byref(stem)) for _ in [tile(stem, 14, 128)]][0]
        assert ws == 128 * 110 * 3 * 49 * 4
        assert tile(stem, 11)[0] not in (11, 14) and tile(stem, 13)[0] not in (13, 14)      # not a temporal / (1,3,3) conv
        r3d = geom(16, 3, 32, 224, 224, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3), f16=1)
        assert tile(r3d, 14, 512) == (14, 512)                                  # fp16 storage: the f16 instantiation
        r3d32 = geom(16, 3, 32, 224, 224, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3))
        assert tile(r3d32, 14, 512)[0] != 14                                    # fp32 tensors, bf16x6 parts: the rows do not fit the LDS
        for bad in (geom(4, 3, 8, 56, 56, 64, (1, 7, 7), (1, 1, 1), (0, 3, 3)),          # unit stride
                    geom(4, 8, 8, 56, 56, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)),          # 8 input channels
                    geom(4, 3, 8, 56, 56, 200, (1, 7, 7), (1, 2, 2), (0, 3, 3)),         # > 128 output channels
                    geom(4, 3, 8, 56, 54, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3))):         # OW = 27: not a multiple of 8
            assert tile(bad, 14)[0] != 14
        temporal = geom(32, 144, 8, 28, 28, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))
        assert tile(temporal, 11, 64)[0] == 11 and tile(temporal, 12, 64)[0] == 12 and tile(temporal, 14)[0] != 14
        spatial = geom(32, 64, 8, 28, 28, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        assert tile(spatial, 13, 64)[0] == 13 and tile(spatial, 11)[0] != 11
        H.lib.gca_set_conv_math(0)                                              # fp32 MFMA: none of the split-product kernels
        assert tile(stem, 14)[0] != 14 and tile(temporal, 11)[0] != 11 and tile(spatial, 13)[0] != 13
    finally:
        H.lib.gca_set_conv_math(default)


@pytest.mark.parametrize('name,octor', [('R2P1D18', lambda o: o.R2Plus1D(18)), ('S3D', lambda o: o.S3D()),
                                        ('R3D18', lambda o: o.R3D(18, 112, 16))])
def test_backbones_match_oracle_state_dict(pkg, name, octor):
    from oracle import encoders as oenc
    torch.manual_seed(11)
    a = octor(oenc)
    torch.manual_seed(11)
    b = getattr(pkg.lib.modeling.backbone.backbone_3d, name)()
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


@pytest.mark.parametrize('mem_type', ['moco', 'simsiam'])
def test_model_factory_keys_match_oracle(pkg, mem_type):
    """create_visual_model: same state-dict keys as the reference wrappers (model.encoder.base_model.*, ...)."""
    from oracle import wrappers as owrap
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity
    cfg = parity.make_cfg(pkg, 'R2P1D18', mem_type, 1024 if mem_type == 'simsiam' else 128, 256, 16)
    torch.manual_seed(3)
    model, ema = pkg.create_visual_model(cfg)
    torch.manual_seed(3)
    om, oe = owrap.create_visual_model('R2P1D18', 16, cfg.CROSS.FEAT_DIM, 'mlp', mem_type)
    assert (ema is None) == (oe is None) == (mem_type == 'simsiam')
    sa, sb = om.state_dict(), model.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert model.model.encoder.feature_dim == 512


def test_graph_block_insertion_keys(pkg):
    """MODEL.AUG_FLAG inserts Sequential(TemporalGraphAug, module) at base.5/9/14 (visual_wrappers.py:121-124)."""
    from oracle import wrappers as owrap
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity
    cfg = parity.make_cfg(pkg, 'S3D', 'simsiam', 1024, 256, 16, aug=True)
    torch.manual_seed(5)
    model, _ = pkg.create_visual_model(cfg)
    torch.manual_seed(5)
    om, _ = owrap.create_visual_model('S3D', 16, 1024, 'mlp', 'simsiam', aug_flag=True)
    sa, sb = om.state_dict(), model.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    assert any(k.endswith('base.5.0.gcns.0.conv.weight') for k in sb)
    assert sb['model.encoder.base_model.base.5.0.g_q.0.weight'].shape == (96, 192, 1, 1, 1)


YAML_MOCO = """
MODEL:
  BACKBONE_TYPE: '3D'
  BACKBONE: 'S3D'
  PRETRAINED: False
  DROPOUT: 0.
INPUT:
  BASE_SIZE: [112, 112]
  VIDEO_LENGTH: 16
DATALOADER:
  BATCH_SIZE: 32
SOLVER:
  BASE_LR: 0.06
  LR_SCHEDULER: 'step'
  STEPS: [80, 120, 160]
  WARMUP_FACTOR: 0.01
  WARMUP_ITERS: 10
  MAX_EPOCHS: 200
CONTRAST:
  MEM_TYPE: 'moco'
  NCE_K: 16384
  NCE_T: 0.07
  ALPHA: 0.999
"""


def test_config_yaml_and_overrides(pkg, tmp_path):
    p = tmp_path / 'visual_moco.yaml'
    p.write_text(YAML_MOCO)
    cfg = pkg.get_defaults()
    cfg.merge_from_file(str(p))
    cfg.merge_from_list(['CONTRAST.NCE_K', '4096', 'MODEL.BACKBONE', 'R2P1D18'])
    cfg.freeze()
    assert cfg.CONTRAST.NCE_K == 4096 and cfg.MODEL.BACKBONE == 'R2P1D18' and cfg.SOLVER.STEPS == [80, 120, 160]
    assert cfg.CROSS.FEAT_DIM == 128 and cfg.SOLVER.BIAS_LR_FACTOR == 2 and cfg.MODEL.DROPOUT == 0.0
    with pytest.raises(AttributeError):
        cfg.MODEL.BACKBONE = 'S3D'
    with pytest.raises(NotImplementedError):
        c2 = cfg.clone()
        c2.CONTRAST.MEM_TYPE = 'bank'
        pkg.create_contrast(c2, 10)


def test_lr_schedule_golden(pkg, golden):
    g = golden('steps')
    want = g.z['lr:weights_epoch0_199']

    class Opt:
        param_groups = [{'lr': 0.06}, {'lr': 0.12}]
    sch = pkg.lib.solver.WarmupMultiStepLR(Opt, [80, 120, 160], 0.1, 0.01, 10, 'linear', 'step', max_epochs=200)
    got = []
    for e in range(200):
        got.append(Opt.param_groups[0]['lr'])
        assert abs(Opt.param_groups[1]['lr'] - 2 * Opt.param_groups[0]['lr']) < 1e-12
        sch.step()
    assert np.allclose(got, want, rtol=1e-6)


def test_param_arena_layout(pkg):
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 300))
    ar = pkg.engine.arena.ParamArena(m)
    assert ar.total % 256 == 0 and ar.offsets == [0, 256, 512, 512 + 2304]
    w0 = m[0].weight.data.clone()
    ar.flat.mul_(2)
    assert torch.equal(m[0].weight.data, 2 * w0)                 # parameters are views of the arena
    assert m[1].weight.grad.data_ptr() == ar.grad[512:].data_ptr()
    tab = ar.chunk_table([1, 2, 3, 4])
    assert tab.tolist()[:2] == [1, 2] and tab[2:11].eq(3).all() and tab[11:].eq(4).all()


# ------------------------------------------------------------------ 2-rank gloo: exchange logic
def _gather(src, idx):
    return src[idx]


def _worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    par = importlib.import_module('video-graph-ssl_amd.parallel')
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ctx = par.DistCtx(rank, world, None)
    b = 5
    g = torch.Generator().manual_seed(0)
    node_x = torch.randn(world * b, 3, 2, generator=g)           # what an all_gather of the clips would hold
    x = node_x[rank * b:(rank + 1) * b].clone()
    out = {}
    for step in range(3):
        ids = par.shared_permutation(world * b, 7, step)         # identical on every rank, no broadcast
        this_x = par.shuffle_exchange(x, ids, ctx, _gather)
        # reference semantics (tools/train_video_contrast_dis.py:213-215): this_x = node_x[ids[rank slice]]
        assert torch.equal(this_x, node_x[ids[rank * b:(rank + 1) * b]])
        k_shuf = this_x.flatten(1).sum(1, keepdim=True) * torch.ones(1, 4)          # stand-in "key encoder"
        all_k = par.gather_keys(k_shuf, ctx)
        assert torch.equal(all_k, (node_x[ids].flatten(1).sum(1, keepdim=True) * torch.ones(1, 4)))   # rank-major
        k = par.unshuffle_keys(all_k, ids, b, ctx, _gather)
        assert torch.allclose(k, x.flatten(1).sum(1, keepdim=True) * torch.ones(1, 4))               # original order
        out['ids%d' % step] = ids
    grad = torch.full((8,), float(rank + 1))
    par.allreduce_sum_(grad, ctx)
    assert torch.equal(grad, torch.full((8,), float(sum(range(1, world + 1)))))
    # the look-ahead plans (device-side indices, here on the CPU) give the same exchange as the on-the-fly ones
    plans = par.ExchangePlans(b, ctx, 7, torch.device('cpu'))
    plans.prefetch(1)
    for step in (0, 1):
        pl = plans.get(step)
        ids = par.shared_permutation(world * b, 7, step)
        assert torch.equal(pl['ids'], ids)
        assert torch.equal(par.shuffle_exchange_planned(x, pl, ctx, _gather), node_x[ids[rank * b:(rank + 1) * b]])
        assert torch.equal(pl['unshuffle_idx'], torch.argsort(ids)[rank * b:(rank + 1) * b])
    # bucketed all-reduce (async handles on gloo) == one all-reduce over the whole arena, bit for bit at two ranks
    gen = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1000, generator=gen)
    whole = flat.clone()
    par.allreduce_sum_(whole, ctx)
    red = par.BucketReducer(flat, ctx)
    for lo, hi in ((700, 1000), (300, 700), (0, 300)):
        red.launch(lo, hi)
    red.wait()
    assert torch.equal(flat, whole)
    t = torch.full((3,), float(rank))
    par.broadcast_(t, ctx)
    assert torch.equal(t, torch.zeros(3))
    q.put((rank, [v.tolist() for v in out.values()]))
    dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    world = 2
    ctxm = mp.get_context('spawn')
    q = ctxm.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]            # the shared permutation is identical on both ranks


def test_gradient_bucket_plan_and_staged_tape(pkg):
    """parallel.plan_buckets: contiguous buckets covering the arena, each tagged with the closure that completes it, in the
    order a reverse sweep completes them; Tape.backward(upto) runs the closures in stages."""
    par = pkg.parallel
    from importlib import import_module
    tape_mod = import_module('video-graph-ssl_amd.engine.tape')
    # 6 parameters of 256 elements; parameter i is written by closure i (a chain), parameter 4 is never written
    offsets, sizes, total = [0, 256, 512, 768, 1024, 1280], [256] * 6, 1536
    first = [0, 1, 2, 3, None, 5]
    bk = par.plan_buckets(first, offsets, sizes, total, 512)
    assert sorted((lo, hi) for _, lo, hi in bk) == [(0, 512), (512, 1024), (1024, 1536)]
    assert [c for c, _, _ in bk] == sorted((c for c, _, _ in bk), reverse=True)
    assert dict(((lo, hi), c) for c, lo, hi in bk) == {(1024, 1536): 5, (512, 1024): 2, (0, 512): 0}
    ran = []
    tp = tape_mod.Tape(True)
    for i in range(6):
        tp.record(lambda i=i: ran.append((i, tape_mod.CURRENT[0])))
    tp.backward(upto=5)
    assert ran == [(5, 5)]
    tp.backward(upto=2)
    assert ran == [(5, 5), (4, 4), (3, 3), (2, 2)]
    tp.backward()
    assert [i for i, _ in ran] == [5, 4, 3, 2, 1, 0] and tape_mod.CURRENT[0] == -1


def test_exchange_plan_bookkeeping(pkg):
    par = pkg.parallel
    b, world = 4, 3
    ids = par.shared_permutation(b * world, 1, 0)
    sends = [par.exchange_plan(ids, b, r, world) for r in range(world)]
    for r in range(world):
        send_idx, send_counts, recv_counts, place = sends[r]
        assert sum(send_counts) == b == sum(recv_counts) and sorted(place.tolist()) == list(range(b))
        for d in range(world):
            assert send_counts[d] == sends[d][2][r]          # what r sends to d is what d expects from r


def test_bench_stdout_carries_only_the_result_line():
    """bench.py's contract is ONE JSON line on stdout.  RCCL writes a version banner to file descriptor 1 from native code when
    the first communicator is created (seen on the GPU box), so bench.py points descriptor 1 at stderr and emits the result
    through a private duplicate of the original stdout: native writes and stray prints must land on stderr."""
    import subprocess
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import bench
        bench.claim_stdout()
        os.write(1, b"native banner\\n")
        print("python print")
        bench.emit({"metric": "m", "value": 1})
    ''' % root)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    assert r.stdout == '{"metric": "m", "value": 1}\n'
    assert 'native banner' in r.stderr and 'python print' in r.stderr
