#!/usr/bin/env python3
"""Headline benchmark: MoCo pre-training clips/sec (R(2+1)D-18, 16 x 112 x 112, 32 clips per GPU) on
N MI355X of one node + InfoNCE forward ms, with the dominant kernel's roofline fraction and the CPU
oracle timed beside it.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pre-training iteration on one synthetic batch already resident in HBM: key
encoder forward, query encoder forward + backward, InfoNCE logits / loss / top-k rank, enqueue,
(gradient all-reduce, ShuffleBN exchange and negatives all-gather for N > 1), SGD step and EMA
update -- nothing skipped.  One JSON line is printed by rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (spec)
PEAK_HBM_GBPS = 8000.0            # HBM3E spec
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak; a bf16x3 / bf16x6 product costs 3 / 6 of them
FWD_GFLOP_PER_VIEW = 16.953       # R(2+1)D-18 @16x112x112, conv+linear, 2*MAC
# --math fp16 = BASELINE.json configs[4]: 3D-ResNet-50, 32-frame 224x224 clips, fp16 storage, 16 clips per GPU (global 128 on 8)
FP16_DEFAULTS = dict(backbone='R3D50', frames=32, size=224, batch=16)


def log(msg):
    sys.stderr.write('[bench %7.1fs] %s\n' % (time.time() - T_START, msg))
    sys.stderr.flush()


# stdout carries exactly ONE line, the JSON result of rank 0.  Native libraries write to file descriptor 1 behind
# Python's back (RCCL prints a five-line version banner there when the first communicator is created), so descriptor 1 is
# pointed at stderr for the life of the process and the result goes to a private duplicate of the original stdout.
_RESULT_FD = [None]


def claim_stdout():
    if _RESULT_FD[0] is None:
        sys.stdout.flush()
        _RESULT_FD[0] = os.dup(1)
        os.dup2(2, 1)


def emit(res):
    line = (json.dumps(res) + '\n').encode()
    if _RESULT_FD[0] is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD[0], line)


T_START = time.time()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=0, help='clips per GPU (default 32; 16 with --math fp16)')
    ap.add_argument('--frames', type=int, default=0, help='default 16 (32 with --math fp16)')
    ap.add_argument('--size', type=int, default=0, help='default 112 (224 with --math fp16)')
    ap.add_argument('--backbone', default='', help='default R2P1D18 (R3D50 with --math fp16)')
    ap.add_argument('--queue', type=int, default=0, help='0 = 4096 at N=1, 65536 at N>1 (BASELINE configs 2/3)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--math', default='bf16x6', choices=['bf16x3', 'bf16x6', 'f32', 'fp16'],
                    help="conv arithmetic (include/gca_hip.h gca_set_conv_math): bf16x3 = fp32 operands split into bf16 "
                         "hi+lo, three bf16 MFMAs per product, fp32 accumulate (4.5e-6 rel. error vs fp32 MFMA, measured); "
                         "bf16x6 = hi+mid+lo, six products, fp32-grade; f32 = fp32 MFMA.  The other modes are timed too and "
                         "reported beside the headline value.  fp16 = the fp16-STORAGE path on BASELINE configs[4] (3D-ResNet-50, 32x224x224, "
                         "16 clips/GPU): feature maps and their gradients fp16 in HBM, v_mfma_f32_32x32x16_f16 with fp32 accumulate, "
                         "fp32 master weights / statistics / head -- an HBM-bound workload, reported against the HBM roofline")
    ap.add_argument('--no-other-math', action='store_true', help='skip timing the other arithmetic modes')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sub-workloads', action='store_true',
                    help='skip the configs[3] (simsiam), configs[4] (fp16) and input-stage sub-records of the default N=1 run')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--layer-table', action='store_true', help='log per-layer conv kernel timings to stderr')
    ap.add_argument('--workload', default='moco', choices=['moco', 'simsiam'],
                    help="moco: BASELINE configs[1]/[2] (default, the headline line); simsiam: configs[3] -- "
                         "configs/visual_simsiam.yaml shape: S3D, 16x224x224, FEAT_DIM 1024, temporal-graph blocks on, 4 clips/GPU")
    args = ap.parse_args()
    base = FP16_DEFAULTS if args.math == 'fp16' else dict(backbone='R2P1D18', frames=16, size=112, batch=32)
    if args.workload == 'simsiam':
        base = dict(backbone='S3D', frames=16, size=224, batch=4)
    for k, v in base.items():
        if not getattr(args, k):
            setattr(args, k, v)
    return args


def make_cfg(pkg, args, K):
    cfg = pkg.get_defaults()
    cfg.merge_from_list(['MODEL.BACKBONE', args.backbone, 'MODEL.BACKBONE_TYPE', '3D', 'MODEL.DROPOUT', 0.0,
                         'MODEL.PRETRAINED', False, 'INPUT.VIDEO_LENGTH', args.frames, 'CONTRAST.MEM_TYPE', 'moco',
                         'CONTRAST.NCE_K', K, 'CONTRAST.NCE_T', 0.07, 'CONTRAST.ALPHA', 0.999, 'CROSS.FEAT_DIM', 128,
                         'SOLVER.BASE_LR', 0.06, 'SOLVER.LR_SCHEDULER', 'step', 'SOLVER.STEPS', [80, 120, 160],
                         'SOLVER.WARMUP_FACTOR', 0.01, 'SOLVER.WARMUP_ITERS', 10, 'SOLVER.MAX_EPOCHS', 200])
    return cfg


def ev_time_ms(fn, reps, warm=3):
    """Average ms of fn() measured with HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def conv_layers_of(model, x_shape, pkg):
    """[(conv module, input shape)] in execution order, via a shape-only walk of the engine forward."""
    rec = []
    L = pkg.engine.layers
    orig = L.HipConv3d.plan

    def spy(self, x):
        rec.append((self, tuple(x.shape), x.stride(0) if not x.is_contiguous() else 0))
        return orig(self, x)
    L.HipConv3d.plan = spy
    try:
        tp = importlib.import_module('video-graph-ssl_amd.engine.tape')
        model.fwd(tp.Tape(False), tp.Var(torch.randn(x_shape, device='cuda')))
    finally:
        L.HipConv3d.plan = orig
    seen, out = set(), []
    for m, shp, xs in rec:           # plan() is called once per forward per conv (dedupe the pack call)
        if id(m) not in seen:
            seen.add(id(m))
            out.append((m, shp, xs))
    return out


def _sym(c, tail=0):
    """Kernel symbol for a gca_conv_kernel_cfg tuple {rows, cols, splits, classes | fast<<8 | vec<<10 | arithmetic<<12} and
    the plan's two-phase code (tune_*_tail: short-tile rows/32 | main column tiles<<8)."""
    vec, fast, math = c[1] == 256, (c[3] >> 8) & 3, (c[3] >> 12) & 3       # bit 10 only says the class COULD use float4 gathers
    if (c[3] >> 17) & 1:                                                    # pointwise fp16 GEMM kernel (conv3d_pw.hip)
        return 'conv_pw_f16_kernel'
    if (c[3] >> 16) & 1:                                                    # stem kernel (conv3d_stem.hip)
        return 'conv_stem_kernel<%d>' % math
    if (c[3] >> 14) & 1:                                                    # LDS-halo kernel (conv3d_halo.hip)
        return 'conv_halo_kernel<%d,%d,%d>' % (c[0] // 32, c[1] // 128, math)
    if (c[3] >> 15) & 1:                                                    # fp16-storage instantiation of the gather kernel
        return 'conv_igemm_kernel<%d,%d,%d,%s,%d,true>' % (c[0] // 32, c[1], fast, 'true' if vec else 'false', math)
    tb = tail & 255
    if tail and c[2] == 1 and c[1] == 128 and (c[3] & 255) == 1 and fast in (1, 2) and 0 < tb <= 2 and tb * 32 < c[0]:
        return 'conv_igemm_2phase_kernel<%d,%d,%d,%d>' % (c[0] // 32, tb, fast, math)
    return 'conv_igemm_kernel<%d,%d,%d,%s,%d>' % (c[0] // 32, c[1], fast, 'true' if vec else 'false', math)


_MATH = [0]          # conv arithmetic of the kernels being timed: 0 fp32 MFMA, 1 bf16x3, 2 bf16x6 (set in main from --math)
DTYPE = {'f32': 'f32',
         'bf16x6': 'bf16x6 (fp32 tensors; conv products as 6 bf16 MFMAs on hi/mid/lo splits = fp32-grade, fp32 accumulate; '
                   'everything else fp32)',
         'bf16x3': 'bf16x3 (fp32 tensors; conv products as 3 bf16 MFMAs on hi/lo splits, 2^-17 relative, fp32 accumulate; '
                   'everything else fp32)',
         'fp16': 'f16 (feature maps and their gradients stored IEEE fp16; conv = v_mfma_f32_32x32x16_f16, fp32 accumulate; fp32 '
                 'master weights, BatchNorm statistics, projection head, InfoNCE and optimizer)'}


def kernel_timing(pkg, trainer, args):
    """Time every conv launch of one training step (each conv layer's forward x2 [key+query], dgrad, wgrad)
    with HIP events on the launch stream and aggregate per KERNEL SYMBOL (what rocprofv3 --stats reports):
    achieved = algorithmic FLOPs of that symbol's launches in one step / their summed duration."""
    ops = pkg.engine.ops
    b = args.batch
    enc = trainer.model.model.encoder.base_model
    layers = conv_layers_of(enc, (b, 3, args.frames, args.size, args.size), pkg)
    # the projection head's Linear layers run on the same kernels (a Linear is a 1x1x1 conv over a 1x1x1 clip)
    import collections
    Lin = collections.namedtuple('Lin', 'out_channels kernel_size stride padding weight')
    for m in trainer.model.modules():
        if m.__class__.__name__ == 'HipLinear':
            layers.append((Lin(m.out_features, (1, 1, 1), (1, 1, 1), (0, 0, 0), m.weight), (b, m.in_features, 1, 1, 1), 0))
    sym = {}

    def add(name, ms, flops, times, launches=1, nbytes=0.0):
        # `launches`: kernel launches behind one call (a strided dgrad is one launch per stride-residue class)
        # `nbytes`: algorithmic HBM bytes of the call = every operand read once + the result written once
        e = sym.setdefault(name, [0.0, 0.0, 0, 0.0])
        e[0] += ms * times; e[1] += flops * times; e[2] += times * launches; e[3] += nbytes * times

    dev = torch.device('cuda', torch.cuda.current_device())
    for i, (m, shp, xs) in enumerate(layers):
        half = ops.ACT_F16[0] and m.__class__.__name__ != 'Lin'               # the head's Linear layers stay fp32
        plan = ops.conv_plan(shp, m.out_channels, m.kernel_size, m.stride, m.padding, dev, act_f16=half)
        N, K, OD, OH, OW = plan.out_shape
        taps = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2]
        flops = 2.0 * N * K * OD * OH * OW * shp[1] * taps
        x = torch.randn(shp, device='cuda').to(plan.act_dtype)
        dy = torch.randn(plan.out_shape, device='cuda').to(plan.act_dtype)
        w = m.weight.data
        wp0, wp1 = ops.conv_pack(plan, 0, w), ops.conv_pack(plan, 1, w)
        dw = torch.zeros_like(w)
        dx = torch.empty(shp, device='cuda', dtype=plan.act_dtype)
        t = ev_time_ms(lambda: ops.conv_fwd(plan, x, wp0, None, stats=True), 5, 1)
        t_f, t_d = t, None
        c0 = plan.cfg(0)
        nbytes = float(x.element_size()) * (x.numel() + dy.numel()) + 4.0 * w.numel()
        add(_sym(c0, plan.g.tune_fwd_tail), t, flops, 2, 1, nbytes)   # key + query forward
        if i > 0:                                                              # the stem never needs d(input)
            # (the head's first Linear does: its input is the encoder feature)
            t = t_d = ev_time_ms(lambda: ops.conv_dgrad(plan, dy, wp1, dx, False), 5, 1)
            c1 = plan.cfg(1)
            add(_sym(c1, plan.g.tune_dgrad_tail), t, flops, 1, max(1, c1[3] & 255), nbytes)
        t = ev_time_ms(lambda: ops.conv_wgrad(plan, x, dy, dw, True), 5, 1)    # includes the split-K reduce
        cw = plan.cfg(2)
        L_ = pkg.engine.layers
        if (L_.LAZY_BN and not half and m.__class__.__name__ != 'Lin' and shp[0] * shp[2] * shp[3] * shp[4] > L_.BN_SMALL_ELEMS
                and ops.conv_xf_ok(plan)):
            # in the step this conv applies its producer's BatchNorm + ReLU while it stages its input (engine.layers
            # _conv_input): time THAT form of the forward and of the weight gradient, as rocprofv3 sees them
            cp = -(-shp[1] // 16) * 16 + 16
            sc_, sf_ = torch.ones(cp, device='cuda'), torch.zeros(cp, device='cuda')
            t_x = ev_time_ms(lambda: ops.conv_fwd_xf(plan, x, sc_, sf_, wp0, stats=True), 5, 1)
            add(_sym(c0, plan.g.tune_fwd_tail), t_x - t_f, 0.0, 2, 0, 0.0)     # replace the plain forward's time, same launches
            t_f = t_x
            t = ev_time_ms(lambda: ops.conv_wgrad(plan, x, dy, dw, True, xf=(sc_, sf_)), 5, 1)
        add('conv_wgrad_kernel<%dx%d>%s' % (cw[0], cw[1], ' (fp16 storage)' if half else ''), t, flops, 1, 1, nbytes)
        if args.layer_table:
            log('L%02d in%-22s K=%-4d k=%s s=%s  GF %7.2f  cfg f%s d%s w%s/%d  fwd %7.3f ms %6.1f TF | dgrad %s | wgrad %7.3f ms %6.1f TF'
                % (i, shp, K, m.kernel_size, m.stride, flops / 1e9, plan.cfg(0)[:3], plan.cfg(1)[:3] if i > 0 else '-',
                   '%dx%d' % plan.cfg(2)[:2], plan.cfg(2)[2], t_f, flops / 1e9 / t_f,
                   ('%7.3f ms %6.1f TF' % (t_d, flops / 1e9 / t_d)) if t_d else '      --       ', t, flops / 1e9 / t))
        del x, dy, dx
    table = {k: dict(ms_per_step=round(v[0], 4), gflop_per_step=round(v[1] / 1e9, 2), launches_per_step=v[2],
                     tflops=round((v[1] / 1e12) / (v[0] / 1e3), 3), algorithmic_GBps=round(v[3] / 1e9 / (v[0] / 1e3), 1))
             for k, v in sym.items()}
    nprod = {0: 1, 1: 3, 2: 6, 3: 1}[_MATH[0]]
    mfma_peak = PEAK_F32_MFMA_TFLOPS if _MATH[0] == 0 else PEAK_BF16_MFMA_TFLOPS / nprod

    def roof_of(dom):
        ach = table[dom]['tflops']
        tr_ = pmc_traffic(dom)
        common = dict(kernel=dom, traffic=tr_,
                      traffic_source=None if tr_ is None else 'profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload '
                                     '(tools/pmc_step.sh), committed -- not re-measured in this run' % PMC_FILE[_MATH[0] == 3],
                      avg_launch_ms=round(sym[dom][0] / sym[dom][2], 4),
                      launches_per_step=sym[dom][2], flops_per_launch_avg=round(sym[dom][1] / sym[dom][2], 1),
                      algorithmic_bytes_per_launch_avg=round(sym[dom][3] / sym[dom][2], 1))
        if _MATH[0] == 0:
            return dict(bound='mfma', achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                        frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), **common)
        # bf16x3 / bf16x6: 3 / 6 bf16 MFMAs per product -> matrix ceiling 2500/3 (/6) TFLOP/s of conv FLOPs (fp16 storage: one
        # f16 MFMA per product, 2500).  The roofline of the kernel is min(that, arithmetic intensity x HBM peak); report
        # against whichever binds.
        ai = sym[dom][1] / sym[dom][3]                      # FLOP per algorithmic byte
        if ai * PEAK_HBM_GBPS / 1e3 < mfma_peak:
            gbps = table[dom]['algorithmic_GBps']
            return dict(bound='hbm', achieved=gbps, peak=PEAK_HBM_GBPS, unit='GB/s', frac=round(gbps / PEAK_HBM_GBPS, 4),
                        flop_per_byte=round(ai, 1), mfma_frac=round(ach / mfma_peak, 4), tflops=ach, **common)
        return dict(bound='mfma', achieved=ach, peak=round(mfma_peak, 1), unit='TFLOP/s', frac=round(ach / mfma_peak, 4),
                    flop_per_byte=round(ai, 1), note='peak = dense bf16/fp16 MFMA 2500 TFLOP/s / %d products' % nprod, **common)

    dom = max(sym, key=lambda k: sym[k][0])
    roof = roof_of(dom)
    if _MATH[0] == 3:
        # configs[4] is the HBM-bound workload: `roofline` is the dominant kernel AMONG THE HBM-BOUND ones
        # (FLOP per byte below the fp16 machine balance) -- the bulk of the step; the one MFMA-bound exception, the C = 3 stem
        # and its weight gradient, is reported beside it when it tops the table
        hbm = [k for k in sym if sym[k][3] > 0 and sym[k][1] / sym[k][3] * PEAK_HBM_GBPS / 1e3 < mfma_peak]
        if hbm:
            dom_h = max(hbm, key=lambda k: sym[k][0])
            if dom_h != dom:
                roof = dict(roof_of(dom_h), largest_kernel_overall=roof)
    return roof, table


def step_bytes_roofline(table, step_s):
    """configs[4] is HBM-bound as a whole: the conv kernels' algorithmic bytes per step against the time of
    the whole step -- a lower bound on the step's HBM efficiency (BatchNorm / pooling passes move bytes too and are not
    counted here)."""
    gb = sum(v['algorithmic_GBps'] * v['ms_per_step'] / 1e3 for v in table.values())
    return dict(conv_algorithmic_GB_per_step=round(gb, 3), step_ms=round(step_s * 1e3, 3),
                GBps=round(gb / step_s, 1), hbm_frac=round(gb / step_s / PEAK_HBM_GBPS, 4))


PMC_FILE = {False: 'pmc_traffic.json', True: 'pmc_traffic_f16.json'}


def pmc_traffic(symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 PMC passes of this same workload
    (profiles/pmc_traffic.json, produced by tools/pmc_step.sh + tools/pmc_parse.py: separate FETCH_SIZE and
    WRITE_SIZE passes; gfx950 correction 2x FETCH_SIZE, calibrated in that file on the EMA/SGD kernels).
    None when the profile has no entry for this kernel (different launch configuration than profiled)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', PMC_FILE[_MATH[0] == 3])     # (the fp16-storage step has its own passes)
    try:
        with open(path) as f:
            prof = json.load(f)['kernels']
    except (OSError, ValueError, KeyError):
        return None
    norm = lambda s: s.replace(' ', '')
    for k, v in prof.items():
        if norm(k) == norm(symbol) and v.get('launches'):
            return round((2.0 * v['fetch_raw'] + v['write_raw']) / v['launches'], 1)
    return None


def pmc_hbm(prefix):
    """HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) of a head kernel from the committed PMC passes of tools/pmc_hbm.sh
    (profiles/r02_pmc_hbm_kernels.json: the same buffer rotation as the timings below); None if not profiled."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r02_pmc_hbm_kernels.json')
    try:
        with open(path) as f:
            for k, v in json.load(f).items():
                if k.startswith(prefix):
                    return v['hbm_bytes_per_launch']
    except (OSError, ValueError, KeyError):
        pass
    return None


def infonce_timing(pkg, b=32, warm=False):
    """InfoNCE forward (logits + row log-sum-exp + top-k rank + loss) at BASELINE's two queue sizes.  The call rotates over
    enough DISTINCT queue / logits buffers that one lap exceeds the 256 MB Infinity Cache (MI355X_MICROARCH.md: scale past
    L3 before reading bytes/s as HBM bandwidth): every launch streams its queue from HBM and writes logits that are not
    resident.  Timed back to back inside one hipGraph (device time of the op, not of a graph launch)."""
    ops = pkg.engine.ops
    out = {}
    for K in (4096, 65536):
        bytes_alg = K * 128 * 4 + 2 * b * 128 * 4 + b * (K + 1) * 4
        nset = max(8, int(320e6 // bytes_alg) + 1) if not warm else 1      # warm: one buffer set replayed (cache-resident; diagnostics)
        q = torch.nn.functional.normalize(torch.randn(b, 128, device='cuda'))
        k = torch.nn.functional.normalize(torch.randn(b, 128, device='cuda'))
        mems = [torch.nn.functional.normalize(torch.randn(K, 128, device='cuda')) for _ in range(nset)]
        lgs = [torch.empty(b, K + 1, device='cuda') for _ in range(nset)]
        ops.moco_logits_fwd(q, k, mems[0], 1 / 0.07, want_lse=True, want_rank=True, want_loss=True, logits=lgs[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nset * (20 if warm else 1)):
                ops.moco_logits_fwd(q, k, mems[i % nset], 1 / 0.07, want_lse=True, want_rank=True, want_loss=True, logits=lgs[i % nset])
        ms = ev_time_ms(g.replay, 20, 3) / (nset * (20 if warm else 1))
        out['K%d' % K] = dict(ms=round(ms, 5), algorithmic_MB=round(bytes_alg / 1e6, 2), distinct_buffer_sets=nset,
                              working_set_MB=round(nset * bytes_alg / 1e6, 1),
                              GBps=round(bytes_alg / 1e9 / (ms / 1e3), 1),
                              hbm_frac=round(bytes_alg / 1e9 / (ms / 1e3) / PEAK_HBM_GBPS, 4),
                              traffic=pmc_hbm('moco_logits_persist_kernel<4> grid=%d' % (8192 if K == 4096 else 65536)) if b == 32 else None)
        del mems, lgs, g
    return out


def graph_timing(pkg, B=4, C=192, T=8, HW=28):
    """Temporal-graph message passing (TemporalGraphAug GCN einsum + skip, temporal_graph.py:56-64) at the
    BASELINE configs[3] site: S3D base.5, (B,192,8,28,28), 8-node clip graph.  HBM-bound: read support +
    write out = 2*B*C*T*H*W*4 bytes.  Rotates over distinct (support, out) pairs whose total exceeds the
    256 MB Infinity Cache, so the GB/s are HBM traffic."""
    ops = pkg.engine.ops
    bytes_alg = 2 * B * C * T * HW * HW * 4 + B * T * T * 4
    nset = max(8, int(320e6 // bytes_alg) + 1)
    sup = [torch.randn(B, C, T, HW, HW, device='cuda') for _ in range(nset)]
    outs = [torch.empty(B, C, T, HW, HW, device='cuda') for _ in range(nset)]
    adj = torch.softmax(torch.randn(B, T, T, device='cuda'), -1)
    ops.graph_gcn_fwd(adj, sup[0], outs[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nset):
            ops.graph_gcn_fwd(adj, sup[i], outs[i])
    ms = ev_time_ms(g.replay, 20, 3) / nset
    return dict(shape=[B, C, T, HW, HW], ms=round(ms, 5), algorithmic_MB=round(bytes_alg / 1e6, 2), distinct_buffer_sets=nset,
                working_set_MB=round(nset * bytes_alg / 1e6, 1),
                GBps=round(bytes_alg / 1e9 / (ms / 1e3), 1), hbm_frac=round(bytes_alg / 1e9 / (ms / 1e3) / PEAK_HBM_GBPS, 4),
                gflops=round(2.0 * B * C * T * T * HW * HW / 1e9 / (ms / 1e3), 1))


def simsiam_main(args, pkg, dev, ctx, world, rank, barrier):
    """BASELINE configs[3]: SimSiam pre-training iteration on S3D with the temporal-graph blocks on."""
    cfg = pkg.get_defaults()
    cfg.merge_from_list(['MODEL.BACKBONE', 'S3D', 'MODEL.BACKBONE_TYPE', '3D', 'MODEL.DROPOUT', 0.0, 'MODEL.PRETRAINED', False,
                         'MODEL.AUG_FLAG', True, 'INPUT.VIDEO_LENGTH', 16, 'CONTRAST.MEM_TYPE', 'simsiam',
                         'CROSS.FEAT_DIM', 1024, 'SOLVER.BASE_LR', 0.06, 'SOLVER.LR_SCHEDULER', 'step',
                         'SOLVER.STEPS', [80, 120, 160], 'SOLVER.WARMUP_FACTOR', 0.01, 'SOLVER.WARMUP_ITERS', 10,
                         'SOLVER.MAX_EPOCHS', 200])
    bsz, size = args.batch, args.size
    tr = pkg.SimSiamTrainer(cfg, dev, ctx=ctx, use_graph=not args.no_graph, seed=1)
    torch.manual_seed(1 + rank)
    images = torch.randn(bsz, 6, 16, size, size, device=dev)
    for i in range(max(args.warmup, 0) + 3):
        tr.train_step(images)
        if rank == 0 and i < 4:
            torch.cuda.synchronize()
            log('warm-up step %d done' % i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.train_step(images)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = float(out['loss'].item())
    tr.close()
    del tr
    if rank != 0:
        return None
    gt = graph_timing(pkg, bsz, 192, 8, size // 8)
    res = {'metric': 'pretrain_clips_per_sec', 'value': round(bsz * world * args.steps / dt, 3), 'unit': 'clips/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[args.math], 'data': 'synthetic',
           'config': {'workload': 'SimSiam pre-training iteration, S3D + temporal-graph blocks (base.5/9/14), 16-frame %dx%d '
                                  'clips, %d clips/GPU, FEAT_DIM 1024, predictor MLP, SGD (BASELINE.json configs[3])' % (size, size, bsz),
                      'global_batch': bsz * world, 'parallelism': 'dp%d' % world, 'hipgraph': not args.no_graph},
           'final_loss': round(loss, 5),
           'step_tflops_algorithmic': round(6 * (35.958 if size == 224 else 8.949) * bsz / 1e3 / (dt / args.steps), 3),
           'roofline': dict(bound='hbm', kernel='tmix_kernel<8> (graph message passing, 8-node clip graph)', achieved=gt['GBps'],
                            peak=PEAK_HBM_GBPS, unit='GB/s', frac=gt['hbm_frac'], traffic=pmc_hbm('tmix_kernel<8>') if (bsz, size) == (4, 224) else None, avg_launch_ms=gt['ms'],
                            algorithmic_MB=gt['algorithmic_MB']),
           'graph_mix_fwd': gt}
    return res


def fp16_subrecord(pkg, dev, ctx, args):
    """BASELINE configs[4] at N = 1 inside the default run: 3D-ResNet-50, 32 x 224 x 224, 16 clips, fp16 storage."""
    import argparse
    import gc
    a = argparse.Namespace(**vars(args))
    a.math = 'fp16'
    for k, v in FP16_DEFAULTS.items():
        setattr(a, k, v)
    ops = pkg.engine.ops
    ops.set_conv_math('fp16')
    saved = _MATH[0]
    _MATH[0] = 3
    try:
        K = 4096
        tr = pkg.MoCoTrainer(make_cfg(pkg, a, K), dev, ctx=ctx, use_graph=not a.no_graph, seed=1)
        torch.manual_seed(1)
        images = torch.randn(a.batch, 6, a.frames, a.size, a.size, device=dev)
        for _ in range(max(a.warmup, 0) + 3):
            out = tr.train_step(images)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = tr.train_step(images)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = {'value': round(a.batch * a.steps / dt, 3), 'unit': 'clips/s', 'dtype': DTYPE['fp16'], 'steps': a.steps,
               'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'final_loss': round(float(out['loss'].item()), 5),
               'loss_scale': float(out['loss_scale'][0]), 'skipped_steps': int(out['loss_scale'][2]),
               'config': {'workload': 'MoCo pre-training iteration, %s, %d-frame %dx%d clips, %d clips/GPU, queue K=%d, fp16 storage '
                                      '(BASELINE.json configs[4] at N=1)' % (a.backbone, a.frames, a.size, a.size, a.batch, K)}}
        if not a.no_kernel_timing:
            rec['roofline'], kernels = kernel_timing(pkg, tr, a)
            rec['roofline_step'] = step_bytes_roofline(kernels, dt / a.steps)
        tr.close()
        del tr, images
        gc.collect()
        torch.cuda.empty_cache()
        return rec
    finally:
        ops.set_conv_math(args.math)
        _MATH[0] = saved


def simsiam_subrecord(pkg, dev, ctx, args):
    """BASELINE configs[3] inside the default run: S3D + graph blocks + SimSiam, 16 x 224 x 224, 4 clips."""
    import argparse
    import gc
    a = argparse.Namespace(**vars(args))
    a.workload, a.backbone, a.frames, a.size, a.batch = 'simsiam', 'S3D', 16, 224, 4
    rec = simsiam_main(a, pkg, dev, ctx, 1, 0, torch.cuda.synchronize)
    gc.collect()
    torch.cuda.empty_cache()
    return {k: rec[k] for k in ('value', 'unit', 'dtype', 'steps', 'warmup', 'ms_per_step', 'final_loss', 'config',
                                'step_tflops_algorithmic', 'roofline')}


def input_stage_subrecord(pkg, dev, tr, args, base_ms):
    """The same iteration fed through the device-side input stage (engine.input): every step stages a uint8 batch
    (b, 2, T, 128, 171, 3) from pinned host memory on the copy stream while the previous step runs, and gca_clip_prepare
    (crop + flip + normalise + layout) writes the trainer's static batch.  PCIe-inclusive; never the headline value."""
    import numpy as np
    inp = pkg.engine.input
    Hs, Ws = max(128, args.size), max(171, args.size)
    stage = inp.DeviceInputStage(args.batch, args.frames, (Hs, Ws), args.size, dev)
    rng = np.random.RandomState(0)
    frames = torch.from_numpy(rng.randint(0, 256, size=(args.batch, 2, args.frames, Hs, Ws, 3)).astype(np.uint8))
    params = np.zeros((args.batch, 2, 4), dtype=np.int32)
    params[..., 0] = rng.randint(0, Hs - args.size + 1, size=(args.batch, 2))
    params[..., 1] = rng.randint(0, Ws - args.size + 1, size=(args.batch, 2))
    params[..., 2] = rng.randint(0, 2, size=(args.batch, 2))
    for _ in range(2):                               # the loader's part, done once: both pinned slots hold a decoded batch
        stage.stage(frames, params)
    torch.cuda.synchronize()

    def feed():                                      # a loader that has filled the slot in place: only the submit remains
        stage.acquire()
        return stage.submit(check=False)
    nxt = feed()
    for _ in range(3):
        cur, nxt = nxt, feed()
        tr.train_step(cur)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cur, nxt = nxt, feed()
        tr.train_step(cur)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the kernel alone, on buffers that are not cache resident (the uint8 batch is 4x smaller than the fp32 one it replaces)
    out = torch.empty(stage.out_shape(), device=dev)
    fd, pd = frames.to(dev), torch.from_numpy(params).to(dev)
    ms = ev_time_ms(lambda: inp.clip_prepare(fd, pd, stage.mean255, stage.inv_std255, args.size, args.size, out=out), 10, 2)
    px = args.batch * 2 * args.frames * args.size * args.size
    return {'value': round(args.batch * args.steps / dt, 3), 'unit': 'clips/s', 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'ms_per_step_resident_fp32': round(base_ms, 3), 'h2d_MB_per_step': round(stage.frame_bytes / 1e6, 1),
            'fp32_batch_MB_it_replaces': round(px * 3 * 4 / 1e6, 1), 'source_frames': [Hs, Ws],
            'clip_prepare': {'ms': round(ms, 4), 'algorithmic_MB': round(px * 15 / 1e6, 1),
                             'GBps': round(px * 15 / 1e9 / (ms / 1e3), 1), 'hbm_frac': round(px * 15 / 1e9 / (ms / 1e3) / PEAK_HBM_GBPS, 4)},
            'note': 'every step copies a uint8 batch from pinned host memory (filled in place by the loader; double-buffered copy '
                    'stream, overlapped with the previous step) and runs gca_clip_prepare; PCIe-inclusive, decode not included'}


def host_cores():
    """Usable host cores: min(affinity mask, cgroup CPU quota, cpu_count)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(args, K):
    """The CPU oracle (oracle/: torch-CPU restatement pinned to the reference by golden fixtures) running the
    same iteration on the host cores, on a bounded sample (b=4 clips), scaled per clip."""
    from oracle import moco as omoco, wrappers as owrap
    cores = host_cores()
    torch.set_num_threads(cores)
    log('cpu baseline on %d host threads' % cores)
    b = 1 if args.math == 'fp16' else 4          # (one 32x224x224 R3D-50 clip pair is ~2 TFLOP of fp32 CPU work)
    torch.manual_seed(1)
    model, ema = owrap.create_visual_model(args.backbone, args.frames, 128, 'mlp', 'moco')
    ema.load_state_dict(model.state_dict())
    contrast = omoco.RGBMoCo(128, K=K, T=0.07)
    opt = omoco.make_optimizer(model, 0.06, 0.9, 5e-4)
    model.train()
    omoco.set_key_encoder_mode(ema)
    crit = omoco.NCESoftmaxLoss()
    images = torch.randn(b, 6, args.frames, args.size, args.size)
    t0 = time.time()
    omoco.moco_train_step(model, ema, contrast, crit, opt, images, 0.999)      # warm-up
    log('cpu baseline warm-up iteration %.1fs' % (time.time() - t0))
    n, t0 = 0, time.time()
    while n < (1 if args.math == 'fp16' else 3) or (time.time() - t0 < 10 and n < 20):
        omoco.moco_train_step(model, ema, contrast, crit, opt, images, 0.999)
        n += 1
        if time.time() - t0 > 60:
            break
    dt = (time.time() - t0) / n
    return dict(value=round(b / dt, 3), unit='clips/s', cores=cores, kind='port',
                sample='%d full MoCo iterations of b=%d clips (%s, %dx%dx%d, K=%d) with the torch-CPU oracle, '
                       '%d threads' % (n, b, args.backbone, args.frames, args.size, args.size, K, cores))


def main():
    args = parse()
    claim_stdout()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                             '--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...' % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    # GCA_BENCH_REHEARSAL=1: every rank on cuda:0, gloo + host-staged collectives -- a functional rehearsal of
    # the N>1 flow on a one-GPU box (numbers from it are meaningless and the JSON line says so).
    rehearsal = os.environ.get('GCA_BENCH_REHEARSAL', '0') == '1' and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # Launch configurations measured on an MI355X and committed next to the profiles: same kernels every run
    # (what profiles/ describes) and no measuring at start-up.  Anything missing or stale is re-measured.
    seed = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'tune_cache.json')
    if 'GCA_TUNE_CACHE' not in os.environ and os.path.exists(seed):
        import shutil, tempfile
        tmp = os.path.join(tempfile.gettempdir(), 'gca_tune_cache_%d_%d.json' % (os.getuid(), rank))
        shutil.copyfile(seed, tmp)
        os.environ['GCA_TUNE_CACHE'] = tmp
    pkg = importlib.import_module('video-graph-ssl_amd')
    ctx = pkg.parallel.DistCtx()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,      # "nccl" == RCCL on ROCm
                                    device_id=dev)
        ctx = pkg.parallel.DistCtx(rank, world, None, host_staged=rehearsal)
    # GCA_BENCH_DIST_SHAPE=1 at N = 1: run the step in its N > 1 SHAPE (graph segments, ShuffleBN exchange, key gather,
    # bucketed all-reduce overlapped with backward -- real RCCL calls in a one-rank group) on one GPU: what the multi-GPU
    # control flow costs before any byte crosses xGMI.  The JSON line says so; it is not the N = 1 measurement.
    dist_shape = world == 1 and os.environ.get('GCA_BENCH_DIST_SHAPE', '0') == '1'
    if dist_shape:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        ctx = pkg.parallel.DistCtx(0, 1, None, force_active=True)
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    pkg.engine.ops.set_conv_math(args.math)
    _MATH[0] = 3 if args.math == 'fp16' else pkg.engine.ops.CONV_MATH[args.math]
    if args.workload == 'simsiam':
        res = simsiam_main(args, pkg, dev, ctx, world, rank, barrier)
        if res is not None:
            emit(res)
        return
    K = args.queue or (4096 if world == 1 else 65536)
    cfg = make_cfg(pkg, args, K)
    tr = pkg.MoCoTrainer(cfg, dev, ctx=ctx, use_graph=not args.no_graph, seed=1)
    if rank == 0:
        log('trainer built')
    torch.manual_seed(1 + rank)
    images = torch.randn(args.batch, 6, args.frames, args.size, args.size, device=dev)

    for i in range(max(args.warmup, 0) + 3):        # +3: two eager warm-up steps and the hipGraph capture
        tr.train_step(images)
        if rank == 0 and i < 4:
            torch.cuda.synchronize()
            log('warm-up step %d done' % i)
    barrier()
    if rank == 0:
        log('timing %d steps' % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.train_step(images)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device='cpu' if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = float(out['loss'].item())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif dist_shape:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank != 0:
        return
    log('timed region %.3fs, loss %.4f' % (dt, loss))
    global_batch = args.batch * world
    value = global_batch * args.steps / dt
    step_gflop = 4 * FWD_GFLOP_PER_VIEW * args.batch if (args.backbone, args.frames, args.size) == ('R2P1D18', 16, 112) else None
    cfg_idx = 4 if args.math == 'fp16' else (1 if world == 1 else 2)
    res = {
        'metric': 'pretrain_clips_per_sec', 'value': round(value, 3), 'unit': 'clips/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[args.math], 'data': 'synthetic',
        'config': {'workload': 'MoCo pre-training iteration, %s, %d-frame %dx%d clips, %d clips/GPU (global %d), '
                               'queue K=%d, T=0.07, SGD+EMA (BASELINE.json configs[%d])'
                               % (args.backbone, args.frames, args.size, args.size, args.batch, global_batch, K, cfg_idx),
                   'global_batch': global_batch, 'parallelism': 'dp%d' % world, 'hipgraph': not args.no_graph, **({'rehearsal': 'gloo/host-staged on one GPU: NOT a measurement'} if rehearsal else {}),
                   **({'dist_shape': 'N > 1 step shape (segments + RCCL calls) in a one-rank group: control-flow cost only'} if dist_shape else {})},
        'views_per_sec': round(2 * value, 3), 'final_loss': round(loss, 5),
    }
    if step_gflop:
        res['step_tflops_algorithmic'] = round(step_gflop / 1e3 / (dt / args.steps), 3)
    if world == 1 and args.math != 'fp16':
        res['infonce_fwd'] = infonce_timing(pkg, 32)
        log('infonce timing done')
        res['infonce_fwd_ms'] = res['infonce_fwd']['K4096']['ms']
        res['graph_mix_fwd'] = graph_timing(pkg)
    if not args.no_kernel_timing:
        res['roofline'], res['kernels'] = kernel_timing(pkg, tr, args)
        log('kernel timing done')
    default_workload = (args.backbone, args.frames, args.size, args.batch, args.math) == ('R2P1D18', 16, 112, 32, 'bf16x6')
    subs = world == 1 and default_workload and not args.no_sub_workloads and not dist_shape
    if subs:
        res['input_stage'] = input_stage_subrecord(pkg, dev, tr, args, dt / args.steps * 1e3)
        log('input stage timed')
    if world == 1 and not args.no_other_math and args.math != 'fp16':
        # the same workload in the other arithmetic modes (own trainer, own tuned plans), timed the same way
        tr.close()
        del tr
        import gc
        for other in [m for m in ('f32', 'bf16x6', 'bf16x3') if m != args.math]:
            gc.collect()
            pkg.engine.ops.set_conv_math(other)
            _MATH[0] = pkg.engine.ops.CONV_MATH[other]
            tr2 = pkg.MoCoTrainer(cfg, dev, ctx=ctx, use_graph=not args.no_graph, seed=1)
            for _ in range(max(args.warmup, 0) + 3):
                tr2.train_step(images)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out2 = tr2.train_step(images)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            key = 'math_' + other
            res[key] = {'value': round(args.batch * args.steps / dt2, 3), 'unit': 'clips/s', 'dtype': DTYPE[other],
                        'ms_per_step': round(dt2 / args.steps * 1e3, 3), 'final_loss': round(float(out2['loss'].item()), 5)}
            if not args.no_kernel_timing:
                res[key]['roofline'], _ = kernel_timing(pkg, tr2, args)
            log('arithmetic mode %s timed' % other)
            del tr2
        gc.collect()
        pkg.engine.ops.set_conv_math(args.math)
        _MATH[0] = pkg.engine.ops.CONV_MATH[args.math]
    if args.math == 'fp16' and not args.no_kernel_timing:
        res['roofline_step'] = step_bytes_roofline(res['kernels'], dt / args.steps)
    if subs:
        import gc
        tr = None
        gc.collect()
        torch.cuda.empty_cache()
        res['simsiam'] = simsiam_subrecord(pkg, dev, ctx, args)       # BASELINE configs[3]
        log('configs[3] (simsiam) timed')
        res['fp16'] = fp16_subrecord(pkg, dev, ctx, args)              # BASELINE configs[4] at N = 1
        log('configs[4] (fp16 storage) timed')
    if world == 1:
        if not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args, K)
    emit(res)


if __name__ == '__main__':
    main()
    import gc
    gc.collect()                      # captured hipGraphs go before the HIP runtime does
    if torch.cuda.is_available():
        torch.cuda.synchronize()
