"""Thin tensor-level wrappers over the C ABI (one function per kernel family).

torch is used for device memory and the current HIP stream only; every arithmetic
operation below runs in libgca_hip.so.  Geometry-dependent host data (gather tables,
workspace sizes) is cached per geometry.
"""
import ctypes as C
import functools
import os

import torch

from .. import _hip as H
from .._hip import aptr, is_half, ptr, stream

F32 = torch.float32
F16 = torch.float16


def _t3(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(a) for a in v)


# ----------------------------------------------------------------------------- workspace
class _Workspace:
    """One grow-only scratch buffer per device.  All kernels run stream-ordered on the current
    stream, so consecutive users never overlap in time.

    A captured hipGraph holds the RAW pointer of the buffer it was recorded with.  A buffer that was handed out
    during a capture is therefore never freed when a later (eager) caller needs more bytes: it is retired -- kept
    alive next to the new, larger one -- so every existing graph keeps replaying into memory it still owns.
    Growth is geometric, so the retired buffers sum to less than the live one."""

    def __init__(self):
        self.buf = {}
        self.captured = set()       # keys whose current buffer has been baked into a graph
        self.retired = []

    def get(self, nbytes, device):
        key = (device.type, device.index, WS_LANE[0])
        b = self.buf.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if b is None or b.numel() < nbytes:
            if capturing:
                raise RuntimeError('workspace would grow during graph capture; run one eager step first')
            if b is not None and key in self.captured:
                self.retired.append(b)
                self.captured.discard(key)
            grown = 0 if b is None else b.numel() + b.numel() // 2
            b = torch.empty(max(int(nbytes), grown, 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        if capturing:
            self.captured.add(key)
        return b


# Scratch lane: work issued on a second stream next to the main one (the key encoder's forward inside the captured
# single-GPU step, engine/trainer.py) takes its scratch from its own buffer -- lanes never share split-K slabs.
WS_LANE = [0]
WS = _Workspace()


# ----------------------------------------------------------------------------- convolution
AUTOTUNE = os.environ.get('GCA_AUTOTUNE', '1') != '0'

# Measured launch configurations, keyed by (pass, geometry).  GCA_TUNE_CACHE=<file> persists them across runs
# (loaded at import, written at exit by rank 0): a restarted job, or a profiler pass that must see the same
# kernels as the run it explains, skips the measurement.
_TUNE_CACHE = {}
_TUNE_DIRTY = [False]
_TUNE_PATH = os.environ.get('GCA_TUNE_CACHE', '')


CONV_MATH = {'f32': 0, 'bf16x3': 1, 'bf16x6': 2}
ACT_F16 = [False]


def set_conv_math(mode):
    """Arithmetic of the conv kernels: 'f32' (fp32 MFMA), 'bf16x6' (fp32-grade split products) or 'bf16x3' (faster); see
    gca_set_conv_math in include/gca_hip.h.  Plans tuned under one mode re-tune under the other (cache keys differ),
    but a plan object keeps the configuration it was tuned with: set the mode before building the model.

    'fp16' is the fp16-STORAGE path (BASELINE configs[4]): every feature map between the input clip and the pooled
    features, and its gradient, lives in HBM as IEEE fp16; the convs run v_mfma_f32_32x32x16_f16 with fp32 accumulation;
    BatchNorm statistics, parameters (fp32 masters), their gradients, the head and the loss stay fp32."""
    ACT_F16[0] = mode == 'fp16'
    H.call('gca_set_conv_math', CONV_MATH['bf16x6' if mode == 'fp16' else mode])


def get_conv_math():
    return 'fp16' if ACT_F16[0] else ('f32', 'bf16x3', 'bf16x6')[H.lib.gca_get_conv_math()]


def cast_f16(x):
    """fp32 -> dense fp16 copy (the input clips of the fp16-storage path); x may be a channel slice of a wider batch."""
    rows, re = (1, x.numel()) if x.is_contiguous() else (x.shape[0], x[0].numel())
    if not x.is_contiguous() and not x[0].is_contiguous():
        raise RuntimeError('only channel slices of a contiguous (N, Ctot, ...) buffer are supported')
    y = torch.empty(x.shape, dtype=F16, device=x.device)
    H.call('gca_cast_f16', ptr(x), rows, re, re if x.is_contiguous() else x.stride(0), aptr(y), stream())
    return y


def load_tune_cache(path):
    import json
    with open(path) as f:
        for k, v in json.load(f).items():
            _TUNE_CACHE[k] = tuple(v)


def save_tune_cache(path):
    import json
    tmp = '%s.tmp.%d' % (path, os.getpid())
    with open(tmp, 'w') as f:
        json.dump({k: list(v) for k, v in sorted(_TUNE_CACHE.items())}, f, indent=0)
    os.replace(tmp, path)


if _TUNE_PATH:
    import atexit
    if os.path.exists(_TUNE_PATH):
        load_tune_cache(_TUNE_PATH)
    atexit.register(lambda: _TUNE_DIRTY[0] and os.environ.get('RANK', '0') == '0' and save_tune_cache(_TUNE_PATH))


def _time_ms(fn, reps=3):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


class ConvPlan:
    """Everything geometry-dependent about one conv: the ABI struct, device gather tables, and the launch
    configuration pinned by a one-off measurement (the role cudnn.benchmark plays in the reference)."""

    def __init__(self, N, Cin, D, Hh, W, K, k, s, p, device, x_batch_stride=0, act_f16=False):
        kd, kh, kw = _t3(k)
        sd, sh, sw = _t3(s)
        pd, ph, pw = _t3(p)
        OD, OH, OW = (D + 2 * pd - kd) // sd + 1, (Hh + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        self.g = H.ConvGeom(N, Cin, D, Hh, W, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, OD, OH, OW, x_batch_stride)
        self.g.act_f16 = int(bool(act_f16))
        self.act_dtype = F16 if act_f16 else F32
        self.gp = C.byref(self.g)
        self.in_shape = (N, Cin, D, Hh, W)
        self.out_shape = (N, K, OD, OH, OW)
        self.device = device
        self.taps = kd * kh * kw
        self._tables = {}
        self.tuned = [not AUTOTUNE] * 3
        if H.lib.gca_conv_fwd_stat_parts(self.gp) < 0:
            raise ValueError('unsupported conv geometry %r' % ((N, Cin, D, Hh, W, K, k, s, p),))
        self.pack_elems = [H.lib.gca_conv_pack_elems(self.gp, 0), H.lib.gca_conv_pack_elems(self.gp, 1)]
        self.refresh()

    def refresh(self):
        """Sizes that depend on the launch configuration."""
        self.parts = H.lib.gca_conv_fwd_stat_parts(self.gp)
        self.wgrad_ws = H.lib.gca_conv_wgrad_ws_bytes(self.gp)
        self.fwd_ws = H.lib.gca_conv_fwd_ws_bytes(self.gp)
        self.dgrad_ws = H.lib.gca_conv_dgrad_ws_bytes(self.gp) if self.g.x_batch_stride == 0 else 0

    def table(self, which):
        t = self._tables.get(which)
        if t is None:
            rows = H.lib.gca_conv_table_rows(self.gp, which)
            host = torch.empty(rows * 2, dtype=torch.int32)
            H.call('gca_conv_table_build_host', self.gp, which, host.data_ptr())
            t = host.to(self.device)
            self._tables[which] = t
        return t

    def cfg(self, which):
        out = (C.c_int32 * 4)()
        if which == 2:
            H.call('gca_conv_wgrad_cfg', self.gp, out)
        else:
            H.call('gca_conv_kernel_cfg', self.gp, which, out)
        return tuple(out)

    WGRAD_SHAPES = {1: (64, 64), 2: (64, 128), 3: (128, 64), 4: (128, 128), 5: (96, 128), 6: (160, 128), 7: (128, 96),
                    8: (128, 160), 9: (64, 192), 10: (192, 64)}

    def _wgrad_candidates(self, M, Nred, kt):
        """(tile shape index, split) pairs.  Shapes whose padding multiplies mostly zeros are skipped; the kernel
        refuses shapes it was not built for (the tuner then just skips them)."""
        cands = []
        least = min(-(-M // bm) * bm * -(-Nred // bn) * bn for bm, bn in self.WGRAD_SHAPES.values())
        for idx, (bm, bn) in self.WGRAD_SHAPES.items():
            padded = -(-M // bm) * bm * -(-Nred // bn) * bn
            if padded > 1.25 * least:
                continue
            tiles = -(-M // bm) * -(-Nred // bn)
            base = max(1, min(kt // 4, 1024 // max(1, tiles)))
            for f in (0.5, 1, 2):
                sp = max(1, min(kt, 1024, int(base * f)))
                if sp * M * Nred * 4 > (256 << 20):
                    continue
                cands.append((idx, sp))
        # the streaming temporal kernel (conv3d_wgrad_ts.hip: tile 11 = 32, 12 = 64 output channels per wave); its split
        # is over (clip, 16-position chunk) units.  The library refuses it where it does not apply (fp32-MFMA mode, ...).
        g = self.g
        hw = g.H * g.W
        if (g.kh, g.kw, g.sd, g.sh, g.sw, g.ph, g.pw) == (1, 1, 1, 1, 1, 0, 0) and g.kd in (3, 7) and hw % 16 == 0 and not g.act_f16:
            units = g.N * (hw // 16)
            for idx, tm in ((11, 1), (12, 2)):
                tiles = -(-M // (32 * tm)) * -(-g.C // 32)
                for nb in (256, 512, 1024):
                    cands.append((idx, max(1, min(units // 4, -(-nb // tiles)))))
        # the streaming (1,3,3) kernel (tile 13): units = (clip, plane, 16-column chunk); one wave per SIMD, so the block
        # count that fills the chip once (256) and its multiples are the candidates worth timing
        if ((g.kd, g.kh, g.kw, g.sd, g.sh, g.sw, g.pd, g.ph, g.pw) == (1, 3, 3, 1, 1, 1, 0, 1, 1) and (g.W % 4 == 0 or g.W <= 16) and g.H >= 2
                and not g.act_f16):
            units = g.N * g.D * -(-g.W // 16)
            tiles = -(-M // 32) * -(-g.C // 32)
            for nb in (256, 512, 768):
                cands.append((13, max(1, min(units // 4, nb // tiles))))
        # the stem kernel (conv3d_wgrad_stem.hip, tile 14): <= 4 input channels, stride 2 along H and W; its split is over
        # (clip, od) units, one workgroup per (split, group of tap planes)
        if g.C <= 4 and (g.sd, g.sh, g.sw) == (1, 2, 2) and g.K <= 128:
            units = g.N * ((g.D + 2 * g.pd - g.kd) // g.sd + 1)
            groups = (-(-g.kd // 2) if g.K <= 64 else g.kd) if g.kd > 1 else 1
            oh = (g.H + 2 * g.ph - g.kh) // g.sh + 1
            for nb in (256, 512, 1024, 2048):
                sp = max(1, min(units * max(1, oh // 8), nb // groups, 1024))     # (the library chunks the output rows past N * OD units)
                if sp * M * Nred * 4 <= (256 << 20):
                    cands.append((14, sp))
        return sorted(set(cands))

    # ---- one-off launch tuning ------------------------------------------------------------
    def _igemm_candidates(self, M, Ntot, kred):
        """(tile code, split) pairs: tile code = rows (32..160) | 1024 for the 256-column float4 variant."""
        nk = -(-kred // 16)
        g = self.g
        pointwise = g.kh == 1 and g.kw == 1 and g.sh == 1 and g.sw == 1 and g.ph == 0 and g.pw == 0
        cands = []
        for vec in ((0, 1024) if pointwise else (0,)):
            bn = 256 if vec else 128
            for bm in (32, 64, 96, 128, 160):
                padded = -(-M // bm) * bm
                if padded > 1.35 * max(M, 32) and bm > 32:       # skip tile heights that mostly multiply zeros
                    continue
                tiles = -(-M // bm) * -(-Ntot // bn)
                for s in (1, 2, 3, 4, 6, 8, 12, 16):
                    if s > 1 and (tiles >= 1024 or nk // s < 4 or s * M * Ntot * 4 > (96 << 20)):
                        continue
                    cands.append((bm | vec, s))
        return cands

    def _halo_candidates(self, which, M):
        """LDS-halo kernel candidates (tile code | 2048, split, box code): the few boxes with the least padding of the
        output grid x halo size, each with the tile heights that pad M least.  Unit-stride dgrad and forward only."""
        g = self.g
        if which == 0:
            q, m, C = (self.out_shape[2], self.out_shape[3], self.out_shape[4]), (g.sd, g.sh, g.sw), g.C
        else:
            if (g.sd, g.sh, g.sw) != (1, 1, 1):
                return []
            q, m, C = (g.D, g.H, g.W), (1, 1, 1), g.K
        k = (g.kd, g.kh, g.kw)
        if C < 16 or self.taps > 64:
            return []
        out = []
        for bn, nbox in ((128, 3), (256, 2)):
            boxes = []
            d = 1
            while d <= bn:
                h = 1
                while d * h <= bn:
                    w = bn // (d * h)
                    b = (d, h, w)
                    if all(b[i] <= 2 * q[i] for i in range(3)):
                        P = 1
                        for i in range(3):
                            P *= (b[i] - 1) * m[i] + k[i]
                        if P <= 384:
                            cover = 1.0
                            for i in range(3):
                                cover *= -(-q[i] // b[i]) * b[i] / q[i]
                            cost = cover * (1.0 + 0.08 * P / bn) * (1.0 if w >= 8 else (1.1 if w >= 4 else 1.3))
                            boxes.append((cost, b))
                    h *= 2
                d *= 2
            boxes.sort()
            tmax = 5 if bn == 128 else 3
            pads = sorted((-(-M // (32 * t)) * 32 * t, -t) for t in range(1, tmax + 1))
            rows = [32 * -t for _, t in pads[:2]]
            for cost, b in boxes[:nbox]:
                if cost > 2.0:
                    continue
                for bm in rows:
                    out.append((bm | 2048, 1, b[0] | (b[1] << 8) | (b[2] << 16)))
        return out

    def tune(self, which, run, repack=None):
        """Measure the candidate launch configurations of pass `which` (0 fwd, 1 dgrad, 2 wgrad) with `run`
        (a closure launching that pass on real operands) and pin the fastest.  `repack()` re-lays the packed weights
        out for the configuration in force (the LDS-halo kernels read another layout than the gather kernels);
        without it only configurations of the current layout are measured."""
        if self.tuned[which] or torch.cuda.is_current_stream_capturing():
            return
        self.tuned[which] = True
        g = self.g
        N, K, OD, OH, OW = self.out_shape
        key = '%d:%s' % (which, ','.join(str(int(v)) for v in (g.N, g.C, g.D, g.H, g.W, g.K, g.kd, g.kh, g.kw, g.sd, g.sh,
                                                                 g.sw, g.pd, g.ph, g.pw, g.x_batch_stride)))
        key = 'v%d%s:%s' % (H.lib.gca_version(), 'h' if g.act_f16 else ('', 'b', 'c')[H.lib.gca_get_conv_math()], key)
        hit = _TUNE_CACHE.get(key)

        def apply(c):
            # igemm: (tile code, splits, tail code, math code, box code); wgrad: (tile shape, splits, math code); short = zeros
            if which == 0:
                g.tune_fwd_bm, g.tune_fwd_splits = c[0], c[1]
                g.tune_fwd_tail = c[2] if len(c) > 2 else 0
                g.tune_fwd_math = c[3] if len(c) > 3 else 0
                g.tune_fwd_box = c[4] if len(c) > 4 else 0
            elif which == 1:
                g.tune_dgrad_bm, g.tune_dgrad_splits = c[0], c[1]
                g.tune_dgrad_tail = c[2] if len(c) > 2 else 0
                g.tune_dgrad_math = c[3] if len(c) > 3 else 0
                g.tune_dgrad_box = c[4] if len(c) > 4 else 0
            else:
                g.tune_wgrad_tile, g.tune_wgrad_splits = c[0], c[1]
                g.tune_wgrad_math = c[2] if len(c) > 2 else 0
            self.refresh()

        def layout():
            """Packed-weight layout the pass reads under the configuration in force: 0 = k-major fp32 rows (gather
            kernels), else the LDS-halo layout, which also depends on the arithmetic (fp32 rows / 2 / 3 bf16 parts)."""
            return 0 if which == 2 else H.lib.gca_conv_pack_layout(self.gp, which)

        layout0 = layout()
        if hit is not None:
            apply(hit)
            if H.lib.gca_conv_fwd_stat_parts(self.gp) >= 0:
                if layout() == layout0:
                    return                                         # still a valid launch code for this library
                if repack is not None:
                    repack()
                    return
                # The measured configuration reads another packed layout and this caller cannot re-pack (no raw weights:
                # the plain HipConv3d.forward path).  Run on the heuristic shape of the layout at hand and leave BOTH the
                # cache entry and the plan's "untuned" state alone: the next caller that can re-pack adopts the entry.
                apply((0, 0, 0, 0, 0) if which < 2 else (0, 0, 0))
                self.tuned[which] = False
                return
            apply((0, 0, 0, 0, 0) if which < 2 else (0, 0, 0))
        if which == 0:
            cands = [c + (0,) for c in self._igemm_candidates(K, N * OD * OH * OW, g.C * self.taps)]
            cands += self._halo_candidates(0, K)
            if g.C <= 4 and g.sw == 2 and g.kw <= 8:
                cands.append((4096 | 64, 1, 0))                   # stem kernel (conv3d_stem.hip), box by its own heuristic
            if g.act_f16 and self.taps == 1:
                cands.append((8192 | 128, 1, 0))                  # pointwise fp16 GEMM kernel (conv3d_pw.hip)
        elif which == 1:
            cands = [c + (0,) for c in self._igemm_candidates(g.C, g.N * g.D * g.H * g.W, K * self.taps)]
            cands += self._halo_candidates(1, g.C)
            if g.act_f16 and self.taps == 1:
                cands.append((8192 | 128, 1, 0))
        else:
            cands = self._wgrad_candidates(K, g.C * self.taps, -(-(N * OD * OH * OW) // 32))
        # The arithmetic mode is a floor on accuracy: a pass may run a MORE accurate kernel when that one is faster
        # (tune_*_math = 1 + arithmetic; f32 > bf16x6 > bf16x3).  0 = the mode itself.
        maths = (0,) if g.act_f16 else {0: (0,), 2: (0, 1), 1: (0, 3, 1)}[H.lib.gca_get_conv_math()]

        packed_as = [layout0]

        def measure(c):
            apply(c)
            if which == 2 and self.cfg(2)[3] & 255 != c[0]:      # shape not available for this tap count
                return None
            if which < 2:
                kc = self.cfg(which)[3]
                if (bool(c[0] & 2048) != bool((kc >> 14) & 1) or bool(c[0] & 4096) != bool((kc >> 16) & 1)
                        or bool(c[0] & 8192) != bool((kc >> 17) & 1)):
                    return None                                   # halo / stem kernel asked for but not runnable here (or vice versa)
                if layout() != packed_as[0]:
                    if repack is None:
                        return None
                    repack()
                    packed_as[0] = layout()
            try:
                return _time_ms(run)
            except RuntimeError:
                return None

        timed = []
        for m in maths:
            for c in cands:
                c = (c[0], c[1], 0, m, c[2]) if which < 2 else (c[0], c[1], m)
                t = measure(c)
                if t is not None:
                    timed.append((t, c))
        timed.sort()
        if which < 2 and timed and not g.act_f16:          # (two-phase launches are built for fp32 storage only)
            # two-phase launches on the fastest single-launch shapes: tall tiles for the full waves of workgroups, short
            # tiles for the remainder (how many workgroups run at once is not known here, so a few guesses are measured)
            M, Ntot = (K, N * OD * OH * OW) if which == 0 else (g.C, g.N * g.D * g.H * g.W)
            single_class = which == 0 or (g.sd == 1 and g.sh == 1 and g.sw == 1)
            tilesN = -(-Ntot // 128)
            for _, base in list(timed[:2]):
                bm, sp, m = base[0], base[1], base[3]
                if not single_class or sp != 1 or bm >= 1024 or bm <= 32:
                    continue
                tilesM = -(-M // bm)
                seen = set()
                for slots in (512, 768, 1024, 1280):
                    full = (tilesM * tilesN // slots) * slots
                    main_cols = full // tilesM
                    if main_cols <= 0 or main_cols >= tilesN or main_cols in seen:
                        continue
                    seen.add(main_cols)
                    for tail_rows in (32, 64):
                        if tail_rows >= bm:
                            continue
                        c = (bm, 1, (tail_rows // 32) | (main_cols << 8), m, 0)
                        t = measure(c)
                        if t is not None:
                            timed.append((t, c))
            timed.sort()
        best = timed[0][1] if timed else None
        if best is not None:
            _TUNE_CACHE[key] = tuple(int(v) for v in best)
            _TUNE_DIRTY[0] = True
            apply(best)
        else:
            apply((0, 0, 0, 0, 0) if which < 2 else (0, 0, 0))
        if which < 2 and repack is not None and layout() != packed_as[0]:
            repack()


@functools.lru_cache(maxsize=None)
def _conv_plan(N, Cin, D, Hh, W, K, k, s, p, dev_type, dev_index, xbs, math):
    return ConvPlan(N, Cin, D, Hh, W, K, k, s, p, torch.device(dev_type, dev_index), xbs, act_f16=math == 3)


def conv_plan(x_shape, K, k, s, p, device, x_batch_stride=0, act_f16=False):
    N, Cin, D, Hh, W = x_shape
    # one plan (= one set of tuned launch configurations) per geometry AND arithmetic mode (3 = fp16 storage)
    return _conv_plan(N, Cin, D, Hh, W, K, _t3(k), _t3(s), _t3(p), device.type, device.index, int(x_batch_stride),
                      3 if act_f16 else H.lib.gca_get_conv_math())


def conv_pack(plan, which, w, out=None):
    n = plan.pack_elems[which]
    if out is None or out.numel() != n:
        out = torch.zeros(n, dtype=F32, device=w.device)      # (room for either layout of the class; the unused part stays 0)
    H.call('gca_conv_pack', plan.gp, which, ptr(w), ptr(out), stream())
    return out


def _act(plan, t):
    """Pointer of an activation operand of a conv, checked against the storage type the plan was built for."""
    if t.dtype is not plan.act_dtype:
        raise TypeError('conv plan built for %s activations, got %s' % (plan.act_dtype, t.dtype))
    return aptr(t)


def _conv_fwd_launch(plan, x, wpack, bias, y, ss, sq):
    ws = WS.get(plan.fwd_ws, x.device) if plan.fwd_ws else None
    H.call('gca_conv_fwd', plan.gp, _act(plan, x), ptr(wpack), ptr(plan.table(0)), ptr(bias), _act(plan, y), ptr(ss), ptr(sq),
           ptr(ws), stream())


def _repacker(plan, which, w_raw, wpack):
    if w_raw is None:
        return None
    return lambda: H.call('gca_conv_pack', plan.gp, which, ptr(w_raw), ptr(wpack), stream())


def conv_fwd(plan, x, wpack, bias=None, stats=False, w_raw=None):
    """-> y [, (stat_sum, stat_sq)]  with stat layout [K][plan.parts].  w_raw: the unpacked weights behind `wpack`; lets
    the one-off launch tuning try configurations that read another packed layout (it re-packs into `wpack`)."""
    y = torch.empty(plan.out_shape, dtype=plan.act_dtype, device=x.device)
    if not plan.tuned[0]:
        plan.tune(0, lambda: _conv_fwd_launch(plan, x, wpack, bias, y, None, None), _repacker(plan, 0, w_raw, wpack))
    ss = sq = None
    if stats:
        ss = torch.empty((plan.g.K, plan.parts), dtype=F32, device=x.device)
        sq = torch.empty((plan.g.K, plan.parts), dtype=F32, device=x.device)
    _conv_fwd_launch(plan, x, wpack, bias, y, ss, sq)
    return (y, (ss, sq)) if stats else y


def conv_xf_ok(plan):
    """True when, under the launch shapes pinned in `plan`, the conv can consume relu(y*scale+shift) of its PRODUCER on the
    fly (forward on an LDS-halo kernel, weight gradient on a streaming kernel): gca_conv_xf_ok."""
    return plan.tuned[0] and plan.tuned[2] and plan.g.x_batch_stride == 0 and bool(H.lib.gca_conv_xf_ok(plan.gp))


def conv_fwd_xf(plan, y_in, scale, shift, wpack, stats=False):
    """conv_fwd on z = relu(y_in * scale[c] + shift[c]) without materialising z (scale / shift: padded rows of bn_finalize)."""
    y = torch.empty(plan.out_shape, dtype=F32, device=y_in.device)
    ss = sq = None
    if stats:
        ss = torch.empty((plan.g.K, plan.parts), dtype=F32, device=y_in.device)
        sq = torch.empty((plan.g.K, plan.parts), dtype=F32, device=y_in.device)
    ws = WS.get(plan.fwd_ws, y_in.device) if plan.fwd_ws else None
    H.call('gca_conv_fwd_xf', plan.gp, ptr(y_in), ptr(scale), ptr(shift), ptr(wpack), ptr(plan.table(0)), None, ptr(y), ptr(ss),
           ptr(sq), ptr(ws), stream())
    return (y, (ss, sq)) if stats else y


def _conv_dgrad_launch(plan, dy, wpack_t, dx, accumulate):
    ws = WS.get(plan.dgrad_ws, dy.device) if plan.dgrad_ws else None
    H.call('gca_conv_dgrad', plan.gp, _act(plan, dy), ptr(wpack_t), ptr(plan.table(1)), _act(plan, dx), int(accumulate), ptr(ws),
           stream())


def conv_dgrad(plan, dy, wpack_t, dx=None, accumulate=False, w_raw=None):
    if dx is None:
        dx = torch.empty(plan.in_shape, dtype=plan.act_dtype, device=dy.device)
        accumulate = False
    if not plan.tuned[1]:
        scratch = torch.empty(plan.in_shape, dtype=plan.act_dtype, device=dy.device)
        plan.tune(1, lambda: _conv_dgrad_launch(plan, dy, wpack_t, scratch, False), _repacker(plan, 1, w_raw, wpack_t))
    _conv_dgrad_launch(plan, dy, wpack_t, dx, accumulate)
    return dx


def _conv_wgrad_launch(plan, x, dy, dw, accumulate):
    ws = WS.get(plan.wgrad_ws, x.device)
    H.call('gca_conv_wgrad', plan.gp, _act(plan, x), _act(plan, dy), ptr(plan.table(2)), ptr(dw), int(accumulate), ptr(ws), stream())


class DeferredReduce(object):
    """Collects the split-K reductions of the weight gradients of one backward pass (stage) and runs them as ONE launch
    (gca_splitk_reduce_batched) instead of one 8-12 us launch per layer.  Each (gradient tensor, plan) pair owns a
    persistent slab buffer (the partial sums must survive until the flush); the job table of a given set of layers is
    built and uploaded once -- during the eager warm-up steps -- and replayed from then on, hipGraph capture included.
    Results are bit-identical to the per-layer reductions (same fold order)."""

    def __init__(self):
        self.slabs = {}          # (dw data_ptr, plan id) -> uint8 slab buffer
        self.tables = {}         # tuple of job signatures -> (device table, njobs, blocks)
        self.pending = []        # [(slab, dw, n, splits, accumulate)]
        self.launches = 0

    def slab_for(self, dw, plan):
        key = (dw.data_ptr(), id(plan))
        b = self.slabs.get(key)
        if b is None or b.numel() < plan.wgrad_ws:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('weight-gradient slab would be allocated during graph capture; run one eager step first')
            b = self.slabs[key] = torch.empty(max(int(plan.wgrad_ws), 16), dtype=torch.uint8, device=dw.device)
        return b

    def add(self, plan, x, dy, dw, accumulate, xf=None):
        if any(j[1].data_ptr() == dw.data_ptr() for j in self.pending):
            self.flush()                                   # a weight used twice (SimSiam's two views): keep the += order, no race
        slab = self.slab_for(dw, plan)
        splits = C.c_int32(0)
        H.call('gca_conv_wgrad_partial', plan.gp, _act(plan, x), ptr(xf[0]) if xf else None, ptr(xf[1]) if xf else None,
               _act(plan, dy), ptr(plan.table(2)), slab.data_ptr(), C.addressof(splits), stream())
        self.pending.append((slab, dw, dw.numel(), int(splits.value), int(bool(accumulate))))

    def flush(self):
        if not self.pending:
            return
        jobs, self.pending = self.pending, []
        sig = tuple((j[0].data_ptr(), j[1].data_ptr(), j[2], j[3], j[4]) for j in jobs)
        t = self.tables.get(sig)
        if t is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('reduce-job table would be built during graph capture; run one eager step first')
            arr = (H.ReduceJob * len(jobs))()
            for r, (slab, dw, n, splits, acc) in zip(arr, jobs):
                r.slabs, r.dw, r.n, r.splits, r.accumulate = slab.data_ptr(), dw.data_ptr(), n, splits, acc
            blocks = H.lib.gca_reduce_jobs_finalize_host(C.addressof(arr), len(jobs))
            if blocks <= 0:
                raise RuntimeError('gca_reduce_jobs_finalize_host: %d' % blocks)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(jobs[0][1].device)
            t = self.tables[sig] = (dev, len(jobs), int(blocks))
        H.call('gca_splitk_reduce_batched', t[0].data_ptr(), t[1], t[2], stream())
        self.launches += 1


# The collector of the backward pass that is running (set by the trainers around Tape.backward), or None: every
# conv_wgrad then reduces its own slabs at once.
DEFER = [None]


def conv_wgrad(plan, x, dy, dw, accumulate=True, xf=None):
    """xf = (scale, shift): `x` is the producer's PRE-activation tensor, the gradient is taken against
    relu(x * scale[c] + shift[c]) (gca_conv_wgrad_xf; only where conv_xf_ok(plan))."""
    if not plan.tuned[2]:
        if xf is not None:
            raise RuntimeError('conv_wgrad(xf=...) needs a tuned plan (conv_xf_ok)')
        scratch = torch.empty_like(dw)
        plan.tune(2, lambda: _conv_wgrad_launch(plan, x, dy, scratch, False))
    d = DEFER[0]
    if d is not None and dw.is_contiguous():
        d.add(plan, x, dy, dw, accumulate, xf)
        return dw
    if xf is not None:
        ws = WS.get(plan.wgrad_ws, x.device)
        H.call('gca_conv_wgrad_xf', plan.gp, _act(plan, x), ptr(xf[0]), ptr(xf[1]), _act(plan, dy), ptr(plan.table(2)), ptr(dw),
               int(accumulate), ptr(ws), stream())
        return dw
    _conv_wgrad_launch(plan, x, dy, dw, accumulate)
    return dw


def bias_grad(dy, N, K, SP, db, accumulate=True):
    H.call('gca_bias_grad', ptr(dy), N, K, SP, ptr(db), int(accumulate), stream())


# ----------------------------------------------------------------------------- batch norm
def bn_stats(x, N, Cc, SP):
    P = H.lib.gca_bn_stats_parts(N, Cc, SP)
    ss = torch.empty((Cc, P), dtype=F32, device=x.device)
    sq = torch.empty((Cc, P), dtype=F32, device=x.device)
    H.call('gca_bn_stats', aptr(x), N, Cc, SP, ptr(ss), ptr(sq), None, is_half(x), stream())
    return ss, sq


def bn_finalize(ss, sq, count, gamma, beta, eps, momentum, rmean, rvar, nbt):
    """-> (save_mean, save_invstd, scale, shift); running stats updated in place."""
    Cc, P = ss.shape
    # rows padded to a multiple of 16 + 16 floats: a consumer that applies (scale, shift) while it stages its input
    # (conv_fwd_xf) reads whole 16-channel chunks
    Cp = -(-Cc // 16) * 16 + 16
    out = torch.empty((4, Cp), dtype=F32, device=ss.device)      # (the pad is never used as a number: consumers select on c < C)
    H.call('gca_bn_finalize', ptr(ss), ptr(sq), P, Cc, float(count), ptr(gamma), ptr(beta), float(eps),
           float(momentum), ptr(rmean), ptr(rvar), ptr(nbt), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]),
           stream())
    return out[0][:Cc], out[1][:Cc], out[2][:Cc], out[3][:Cc]


def bn_train_fwd(ss, sq, count, gamma, beta, eps, momentum, rmean, rvar, nbt, x, residual, relu, N, Cc, SP, out=None):
    """bn_finalize + bn_apply behind one call (one launch for small N*SP).
    -> (z, save_mean, save_invstd, scale, shift); running stats updated in place."""
    P = ss.shape[1]
    st = torch.empty((4, Cc), dtype=F32, device=ss.device)
    z = torch.empty_like(x) if out is None else out
    H.call('gca_bn_train_fwd', ptr(ss), ptr(sq), P, Cc, float(count), ptr(gamma), ptr(beta), float(eps), float(momentum),
           ptr(rmean), ptr(rvar), ptr(nbt), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), aptr(x), aptr(residual),
           int(relu), N, SP, aptr(z), _slice_stride(z, Cc, SP), is_half(x, residual, z), stream())
    return z, st[0], st[1], st[2], st[3]


# GCA_FUSE_SPLITK_BN=0: a split-K conv in front of a small BatchNorm finishes its slabs in its own launch (A/B runs)
FUSE_SPLITK_BN = os.environ.get('GCA_FUSE_SPLITK_BN', '1') != '0'


def conv_bn_small_fwd(plan, x, wpack, count, gamma, beta, eps, momentum, rmean, rvar, nbt, residual, relu, out=None):
    """conv forward whose pinned launch shape splits the reduction + training-mode BatchNorm of a small map, in two launches:
    the conv leaves its slabs in the scratch lane (gca_conv_fwd_slabs), the BatchNorm kernel folds them, writes y and z
    (gca_bn_train_fwd_slabs).  -> (y, z, save_mean, save_invstd, scale, shift)"""
    N, K, OD, OH, OW = plan.out_shape
    SP = OD * OH * OW
    ws = WS.get(plan.fwd_ws, x.device)
    splits = C.c_int32(0)
    H.call('gca_conv_fwd_slabs', plan.gp, ptr(x), ptr(wpack), ptr(plan.table(0)), ptr(ws), C.addressof(splits), stream())
    y = torch.empty(plan.out_shape, dtype=F32, device=x.device)
    z = torch.empty_like(y) if out is None else out
    st = torch.empty((4, K), dtype=F32, device=x.device)
    H.call('gca_bn_train_fwd_slabs', ptr(ws), int(splits.value), float(count), ptr(gamma), ptr(beta), float(eps), float(momentum),
           ptr(rmean), ptr(rvar), ptr(nbt), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), ptr(y), ptr(residual), int(relu),
           N, K, SP, ptr(z), _slice_stride(z, K, SP), stream())
    return y, z, st[0], st[1], st[2], st[3]


def bn_fold_eval(gamma, beta, rmean, rvar, eps):
    Cc = rmean.numel()
    out = torch.empty((2, Cc), dtype=F32, device=rmean.device)
    H.call('gca_bn_fold_eval', ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), float(eps), Cc, ptr(out[0]), ptr(out[1]),
           stream())
    return out[0], out[1]


def _slice_stride(t, Cc, SP):
    """0 for a contiguous (N,C,...) tensor, else the batch stride of a channel-slice view."""
    if t.is_contiguous():
        return 0
    if t.dim() < 2 or t[0].is_contiguous() is False:
        raise RuntimeError('only channel slices of a contiguous (N, Ctot, ...) buffer are supported')
    return t.stride(0)


def bn_apply(x, scale, shift, residual, relu, N, Cc, SP, out=None):
    z = torch.empty_like(x) if out is None else out
    H.call('gca_bn_apply', aptr(x), ptr(scale), ptr(shift), aptr(residual), int(relu), N, Cc, SP, aptr(z),
           _slice_stride(z, Cc, SP), is_half(x, residual, z), stream())
    return z


def bn_bwd(dz, z, x, gamma, mean, invstd, relu, N, Cc, SP, dgamma, dbeta, dres=None, dres_accumulate=False,
           scale=None, shift=None):
    """relu: False/0 none, True/1 mask from z, 2 mask recomputed from x with the forward's (scale, shift)."""
    dx = torch.empty_like(x)
    ws = WS.get(H.lib.gca_bn_bwd_ws_bytes(N, Cc, SP), x.device)
    H.call('gca_bn_bwd', aptr(dz), aptr(z) if int(relu) == 1 else None, aptr(x), ptr(gamma), ptr(mean), ptr(invstd), int(relu),
           N, Cc, SP, aptr(dx), ptr(dgamma), ptr(dbeta), aptr(dres), int(dres_accumulate), _slice_stride(dz, Cc, SP),
           ptr(scale), ptr(shift), ptr(ws), is_half(dz, z if int(relu) == 1 else None, x, dres), stream())
    return dx


# ----------------------------------------------------------------------------- pooling
class PoolPlan:
    def __init__(self, x_shape, k, s, p):
        N, Cc, D, Hh, W = x_shape
        kd, kh, kw = _t3(k)
        sd, sh, sw = _t3(s)
        pd, ph, pw = _t3(p)
        OD, OH, OW = (D + 2 * pd - kd) // sd + 1, (Hh + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        self.g = H.PoolGeom(N, Cc, D, Hh, W, kd, kh, kw, sd, sh, sw, pd, ph, pw, OD, OH, OW)
        self.gp = C.byref(self.g)
        self.in_shape, self.out_shape = tuple(x_shape), (N, Cc, OD, OH, OW)


@functools.lru_cache(maxsize=None)
def pool_plan(x_shape, k, s, p):
    return PoolPlan(x_shape, k, s, p)


def maxpool_fwd(plan, x, want_argmax=True, scale=None, shift=None):
    """scale/shift: pool relu(x*scale[c]+shift[c]) (the BatchNorm+ReLU in front of the pool) without materialising it."""
    y = torch.empty(plan.out_shape, dtype=x.dtype, device=x.device)
    am = torch.empty(plan.out_shape, dtype=torch.int32, device=x.device) if want_argmax else None
    H.call('gca_maxpool3d_fwd', plan.gp, aptr(x), aptr(y), ptr(am), ptr(scale), ptr(shift), is_half(x), stream())
    return y, am


def maxpool_bwd(plan, dy, argmax, dx=None, accumulate=False):
    if dx is None:
        dx = torch.empty(plan.in_shape, dtype=dy.dtype, device=dy.device)
        accumulate = False
    H.call('gca_maxpool3d_bwd', plan.gp, aptr(dy), ptr(argmax), aptr(dx), int(accumulate), is_half(dy, dx), stream())
    return dx


def avgpool_fwd(plan, x):
    """nn.AvgPool3d(kernel_size=k) (stride k, no padding) on fp32 maps."""
    y = torch.empty(plan.out_shape, dtype=F32, device=x.device)
    H.call('gca_avgpool3d_fwd', plan.gp, ptr(x), ptr(y), stream())
    return y


def avgpool_bwd(plan, dy, dx=None, accumulate=False):
    if dx is None:
        dx = torch.empty(plan.in_shape, dtype=F32, device=dy.device)
        accumulate = False
    H.call('gca_avgpool3d_bwd', plan.gp, ptr(dy), ptr(dx), int(accumulate), stream())
    return dx


def wavgpool_fwd(x, wt, norm):
    N, Cc, D, Hh, W = x.shape
    y = torch.empty((N, Cc), dtype=F32, device=x.device)
    H.call('gca_wavgpool_fwd', aptr(x), ptr(wt), float(norm), N * Cc, D, Hh * W, ptr(y), is_half(x), stream())
    return y


def wavgpool_bwd(dy, wt, norm, x_shape, dtype=F32):
    N, Cc, D, Hh, W = x_shape
    dx = torch.empty(x_shape, dtype=dtype, device=dy.device)
    H.call('gca_wavgpool_bwd', ptr(dy), ptr(wt), float(norm), N * Cc, D, Hh * W, aptr(dx), is_half(dx), stream())
    return dx


# ----------------------------------------------------------------------------- head pieces
def relu_fwd(x):
    y = torch.empty_like(x)
    H.call('gca_relu_fwd', ptr(x), x.numel(), ptr(y), stream())
    return y


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    H.call('gca_relu_bwd', ptr(dy), ptr(y), dy.numel(), ptr(dx), stream())
    return dx


def l2norm_fwd(x, eps=1e-12):
    rows, dim = x.shape
    y = torch.empty_like(x)
    inv = torch.empty(rows, dtype=F32, device=x.device)
    H.call('gca_l2norm_fwd', ptr(x), rows, dim, float(eps), ptr(y), ptr(inv), stream())
    return y, inv


def l2norm_bwd(dy, y, inv):
    rows, dim = y.shape
    dx = torch.empty_like(y)
    H.call('gca_l2norm_bwd', ptr(dy), ptr(y), ptr(inv), rows, dim, ptr(dx), stream())
    return dx


def negcos(p, z, scale, loss_buf, accumulate):
    """loss_buf: (1 + rows) floats; returns dp."""
    rows, dim = p.shape
    dp = torch.empty_like(p)
    H.call('gca_negcos_fwd_bwd', ptr(p), ptr(z), rows, dim, float(scale), ptr(loss_buf), int(accumulate), ptr(dp),
           stream())
    return dp


# ----------------------------------------------------------------------------- MoCo / InfoNCE
_NCE_SYNC = {}


def _nce_sync(device):
    """The zero-initialised ticket counter of the single-launch InfoNCE forward.  Every call leaves it at zero (the last
    workgroup re-arms it; the GCA_NCE_DEBUG early exits either draw no ticket at all or re-arm it too), and calls on ONE
    stream are ordered, so a counter may be shared exactly by the calls that already share a scratch buffer: one per
    (device, scratch lane).  Work issued on a second stream must run under its own lane (ops.WS_LANE, as the trainers' key
    encoder does) -- that rule already holds for every workspace user (the partials of this kernel live in WS.get), and with
    it two concurrent RGBMoCo.forward calls never interleave their tickets."""
    key = (device.type, device.index, WS_LANE[0])
    c = _NCE_SYNC.get(key)
    if c is None:
        c = _NCE_SYNC[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return c


def moco_logits_fwd(q, k, queue, inv_T, want_lse=False, want_rank=False, want_loss=False, logits=None):
    """-> (logits, lse, rank) [+ (loss,) when want_loss: the InfoNCE loss of these logits, fused into the same launch
    (b <= 32) -- the separate nce_loss_fwd call is then unnecessary].  logits: optional (b, K+1) output buffer."""
    b, D = q.shape
    K = queue.shape[0]
    want_lse = want_lse or want_loss
    if logits is None:
        logits = torch.empty((b, K + 1), dtype=F32, device=q.device)
    lse = torch.empty(b, dtype=F32, device=q.device) if want_lse else None
    rank = torch.empty(b, dtype=torch.int32, device=q.device) if want_rank else None
    loss = torch.empty(1, dtype=F32, device=q.device) if want_loss else None
    ws = WS.get(H.lib.gca_infonce_ws_bytes(b, K), q.device)
    H.call('gca_moco_logits_fwd', ptr(q), ptr(k), ptr(queue), b, K, D, float(inv_T), ptr(logits), ptr(lse), ptr(rank),
           ptr(loss), ptr(ws), ptr(_nce_sync(q.device)), stream())
    return (logits, lse, rank, loss) if want_loss else (logits, lse, rank)


def nce_loss_fwd(logits, lse=None):
    b, ncol = logits.shape
    loss = torch.empty(1, dtype=F32, device=logits.device)
    lse_out = None
    if lse is None:
        lse_out = torch.empty(b, dtype=F32, device=logits.device)
    H.call('gca_nce_softmax_loss_fwd', ptr(logits), b, ncol, ptr(lse), ptr(lse_out), ptr(loss), stream())
    return loss, (lse if lse is not None else lse_out)


def nce_loss_bwd(logits, lse, gscale_dev=None, gscale_host=1.0):
    b, ncol = logits.shape
    dl = torch.empty_like(logits)
    H.call('gca_nce_softmax_loss_bwd', ptr(logits), ptr(lse), b, ncol, ptr(gscale_dev), float(gscale_host), ptr(dl),
           stream())
    return dl


def moco_logits_bwd(k, queue, inv_T, dlogits=None, logits=None, lse=None, gscale_dev=None, gscale_host=1.0,
                    ov_start=0, ov_rows=None, ov_start_dev=None):
    b, D = k.shape
    K = queue.shape[0]
    dq = torch.empty((b, D), dtype=F32, device=k.device)
    ws = WS.get(H.lib.gca_infonce_ws_bytes(b, K), k.device)
    ov_n = 0 if ov_rows is None else ov_rows.shape[0]
    H.call('gca_moco_logits_bwd', ptr(dlogits), ptr(logits), ptr(lse), ptr(gscale_dev), float(gscale_host), ptr(k),
           ptr(queue), b, K, D, float(inv_T), int(ov_start), ptr(ov_start_dev), int(ov_n), ptr(ov_rows), ptr(dq),
           ptr(ws), stream())
    return dq


def queue_enqueue(queue, keys, ptr_index, save=False, ptr_dev=None):
    K, D = queue.shape
    n = keys.shape[0]
    saved = torch.empty((n, D), dtype=F32, device=queue.device) if save else None
    H.call('gca_queue_enqueue', ptr(queue), K, D, ptr(keys), n, int(ptr_index), ptr(ptr_dev), ptr(saved), stream())
    return saved


def queue_advance(ptr_dev, n, K):
    H.call('gca_queue_advance', ptr(ptr_dev), int(n), int(K), stream())


# ----------------------------------------------------------------------------- graph block
def graph_adj_fwd(gq, gk, u, max_hop, alpha, temperature):
    B, Ci, T = gq.shape[0], gq.shape[1], gq.shape[2]
    HW = gq.shape[3] * gq.shape[4]
    out = torch.empty((3, B, T, T), dtype=F32, device=gq.device)
    ws = WS.get(H.lib.gca_graph_gram_ws_bytes(B, Ci, T, HW), gq.device)
    H.call('gca_graph_adj_fwd', ptr(gq), ptr(gk), B, Ci, T, HW, int(max_hop), float(alpha), float(temperature), ptr(u),
           ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(ws), stream())
    return out[0], out[1], out[2]          # sim, adj_pre, adj


def graph_adj_bwd(dadj, gq, gk, sim, pre, adj, max_hop, alpha, temperature):
    B, Ci, T = gq.shape[0], gq.shape[1], gq.shape[2]
    HW = gq.shape[3] * gq.shape[4]
    dgq, dgk = torch.empty_like(gq), torch.empty_like(gk)
    H.call('gca_graph_adj_bwd', ptr(dadj), ptr(gq), ptr(gk), ptr(sim), ptr(pre), ptr(adj), B, Ci, T, HW, int(max_hop),
           float(alpha), float(temperature), ptr(dgq), ptr(dgk), stream())
    return dgq, dgk


def graph_gcn_fwd(adj, s, out=None):
    B, Cc, T = s.shape[0], s.shape[1], s.shape[2]
    if out is None:
        out = torch.empty_like(s)
    H.call('gca_graph_gcn_fwd', ptr(adj), ptr(s), B, Cc, T, s.shape[3] * s.shape[4], ptr(out), stream())
    return out


def graph_gcn_bwd(adj, s, dout, want_dadj=True):
    B, Cc, T = s.shape[0], s.shape[1], s.shape[2]
    ds = torch.empty_like(s)
    dadj = torch.empty((B, T, T), dtype=F32, device=s.device) if want_dadj else None
    ws = WS.get(H.lib.gca_graph_gcn_bwd_ws_bytes(B, Cc, T, s.shape[3] * s.shape[4]), s.device) if want_dadj else None
    H.call('gca_graph_gcn_bwd', ptr(adj), ptr(s), ptr(dout), B, Cc, T, s.shape[3] * s.shape[4], ptr(ds), ptr(dadj),
           ptr(ws), stream())
    return ds, dadj


# ----------------------------------------------------------------------------- flat-arena updates
def ema_update(p_ema, p, m):
    H.call('gca_ema_update', ptr(p_ema), ptr(p), p.numel(), float(m), stream())


def sgd_step(p, g, buf, chunk_lr, chunk_wd, lr_scale, momentum, nesterov, grad_clip=None):
    """grad_clip: the (norm, coef) pair of grad_clip_coef -- gradients are scaled by coef inside the update."""
    H.call('gca_sgd_step', ptr(p), ptr(g), ptr(buf), p.numel(), ptr(chunk_lr), ptr(chunk_wd), float(lr_scale),
           float(momentum), int(nesterov), 0, ptr(grad_clip), stream())


def grad_clip_coef(g, max_norm, out=None):
    """clip_grad_norm_ over the flat gradient arena -> device tensor (total_norm, clip coefficient, 0, 0)."""
    if out is None:
        out = torch.empty(4, dtype=F32, device=g.device)
    ws = WS.get(H.lib.gca_grad_clip_ws_bytes(), g.device)
    H.call('gca_grad_clip_coef', ptr(g), g.numel(), float(max_norm), ptr(out), ptr(ws), stream())
    return out


# Dynamic loss scaling of the fp16-storage path (apex amp's LossScaler defaults: x2 after 2000 clean steps, x0.5 and a
# skipped optimizer step on inf / nan, scale capped at 2^24), kept on the device: see gca_grad_unscale_clip.
LOSS_SCALE_GROWTH, LOSS_SCALE_BACKOFF, LOSS_SCALE_INTERVAL, LOSS_SCALE_MAX = 2.0, 0.5, 2000, 2.0 ** 24
# The (scale, clean steps, skipped steps, steps) state of the trainer that is running its step, or None for fp32 storage:
# loss-gradient seeds inside the models (SimSiam's negative cosine) multiply by its first element.
LOSS_SCALE_STATE = [None]


def loss_scale_state(initial, device):
    return torch.tensor([float(initial), 0.0, 0.0, 0.0], dtype=F32, device=device)


def grad_unscale_clip(g, state, max_norm=None, out=None):
    """Un-scale (by the device-resident loss scale in `state`), overflow-check and clip the flat gradient arena without
    touching it: -> device tensor (norm of g / S, factor for the SGD kernel, skip flag, S); `state` is advanced."""
    if out is None:
        out = torch.empty(4, dtype=F32, device=g.device)
    ws = WS.get(H.lib.gca_grad_clip_ws_bytes(), g.device)
    H.call('gca_grad_unscale_clip', ptr(g), g.numel(), float(max_norm or 0.0), ptr(state), LOSS_SCALE_GROWTH,
           LOSS_SCALE_BACKOFF, int(LOSS_SCALE_INTERVAL), LOSS_SCALE_MAX, ptr(out), ptr(ws), stream())
    return out


def scale_dev_(y, a_dev, a_host=1.0):
    """y *= a_dev[0] * a_host (a device-resident factor: the loss scale)."""
    H.call('gca_scale_dev', ptr(y), y.numel(), ptr(a_dev), float(a_host), stream())


def fill(t, v):
    H.call('gca_fill', ptr(t), t.numel(), float(v), stream())


def axpy(y, x, a=1.0):
    if is_half(y, x):
        H.call('gca_axpy_f16', aptr(y), aptr(x), y.numel(), float(a), stream())
    else:
        H.call('gca_axpy', ptr(y), ptr(x), y.numel(), float(a), stream())


def scale_(y, a):
    H.call('gca_scale', ptr(y), y.numel(), float(a), stream())


def gather_rows(src, idx):
    """dst[r] = src[idx[r]] over dim 0.  src may be a batch-strided view (each row itself contiguous)."""
    rows = idx.numel()
    re = src[0].numel()
    if not src[0].is_contiguous() or idx.dtype != torch.long:
        raise ValueError('gather_rows: rows must be contiguous, idx int64')
    dst = torch.empty((rows,) + tuple(src.shape[1:]), dtype=F32, device=src.device)
    H.call('gca_gather_rows', ptr(src), ptr(idx), rows, re, src.stride(0) if src.shape[0] > 1 else re, ptr(dst), stream())
    return dst
