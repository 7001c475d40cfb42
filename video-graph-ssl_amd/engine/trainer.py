"""Pre-training step drivers (MoCo and SimSiam) on the HIP engine.

Restates the iteration of tools/train_video_contrast_dis.py:395-454 (_train_moco) and :479-523
(_train_simsiam) without its host syncs (3x .item() per iteration, :429-431) and defects
(SURVEY.md fact 7).  One iteration =

    key encoder fwd (no grad, BN in train mode)      ShuffleBN exchange / negatives all-gather (N>1)
    query encoder fwd                                InfoNCE: logits GEMM + LSE + loss + top-k rank
    enqueue keys -> queue                            dq -> encoder backward (tape)
    gradient all-reduce (N>1) -> fused SGD           fused EMA of the key encoder

Everything between the collectives is a fixed kernel sequence, so it is captured into hipGraphs
(one for N=1; three segments around the collectives for N>1) and replayed per step.
"""
import atexit
import math
import os
import weakref

import torch

from . import layers as L
from . import ops
from .arena import arena_of
from .input import StagedBatch
from .layers import BatchedPacker
from .tape import Tape, Var
from .. import parallel as par
from ..lib.memory import create_contrast, create_criterion
from ..lib.modeling import create_visual_model
from ..lib.solver import make_lr_scheduler, make_optimizer
from ..lib.solver.build import clip_value_of


def set_key_encoder_mode(model_ema):
    """eval() but every BatchNorm stays in train mode (tools/...dis.py:383-389)."""
    model_ema.eval()
    for m in model_ema.modules():
        if m.__class__.__name__.find('BatchNorm') != -1:
            m.train()


# Every live _Graphed, weakly: captured hipGraphs must be destroyed while the HIP runtime is still up.  A graph that is
# only released by interpreter shutdown (a trainer held at module scope, the INTEGRATION.md usage) is torn down after the
# runtime and aborts the process, so the package itself resets whatever is still alive from an atexit hook -- registered at
# import, i.e. after torch's own hooks, and atexit runs last-registered first.
_LIVE_GRAPHS = weakref.WeakSet()


def release_all_graphs():
    """Destroy every captured hipGraph of this process (idempotent).  Runs at interpreter exit; callable by hand."""
    live = list(_LIVE_GRAPHS)
    for g in live:
        g.release()
    if live and torch.cuda.is_available():
        try:
            torch.cuda.synchronize()
        except RuntimeError:
            pass


atexit.register(release_all_graphs)


class _Graphed(object):
    """Capture-once / replay wrapper for a no-argument closure working on static buffers."""

    def __init__(self, fn, enabled, pool=None):
        self.fn, self.enabled, self.graph, self.warm, self.pool = fn, enabled, None, 0, pool
        _LIVE_GRAPHS.add(self)

    def release(self):
        """Drop the captured graph (the next run() re-captures after its eager warm-up)."""
        g, self.graph, self.warm = self.graph, None, 0
        if g is not None:
            torch.cuda.synchronize()
            g.reset()

    def run(self):
        if not self.enabled:
            return self.fn()
        if self.graph is None:
            if self.warm < 2:               # eager warm-up: builds plans/tables/workspaces, warms allocator
                self.warm += 1
                return self.fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # with a process group alive, its watchdog thread polls events while we capture: only THIS thread's
            # calls must be checked against the capture (the default "global" mode would fail the capture)
            mode = 'thread_local' if torch.distributed.is_available() and torch.distributed.is_initialized() else 'global'
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode=mode):    # records the launches, does not execute them
                self.fn()
            self.graph = g
        self.graph.replay()


def _check_unsupported(cfg):
    """Options the reference's trainer honours and this one does not must fail loudly, not train differently."""
    apex = getattr(cfg, 'APEX', None)
    if apex is not None and bool(getattr(apex, 'FLAG', False)):
        raise NotImplementedError('APEX.FLAG (apex amp, tools/train_video_contrast_dis.py:134-141) is not available: '
                                  'select the reduced-precision conv path with engine.ops.set_conv_math instead')


# GCA_FORK_KEY=0: key and query encoders on one stream inside the captured step as well (A/B runs)
FORK_KEY_ENCODER = os.environ.get('GCA_FORK_KEY', '1') != '0'
# gradient all-reduce bucket: 8 M fp32 = 32 MB (four buckets for R(2+1)D-18 + head); 0 = one all-reduce after the backward pass
BUCKET_ELEMS = int(os.environ.get('GCA_BUCKET_ELEMS', 8 << 20))


class _ShapeOnly(object):
    """Stand-in for the fp32 batch a StagedBatch will become (shape / device bookkeeping of _ensure_static)."""

    def __init__(self, shape, device):
        self.shape, self.device, self.dtype = torch.Size(shape), device, torch.float32


# GCA_DEFER_REDUCE=0: every weight gradient reduces its split-K slabs in its own launch (A/B runs)
DEFER_REDUCE = os.environ.get('GCA_DEFER_REDUCE', '1') != '0'


class _TrainerBase(object):
    def _backward(self, tape, upto):
        """Tape.backward with the weight-gradient split-K reductions of the stage collected into one launch."""
        if not DEFER_REDUCE:
            return tape.backward(upto)
        if getattr(self, '_deferred', None) is None:
            self._deferred = ops.DeferredReduce()
        if ops.DEFER[0] is not None:
            raise RuntimeError('another backward pass is collecting weight-gradient reductions')
        ops.DEFER[0] = self._deferred
        try:
            tape.backward(upto)
        finally:
            ops.DEFER[0] = None

    def _buckets_from_log(self, arena, log):
        """log: (backward closure index, parameter) pairs of one eager backward pass (engine.layers.GRAD_LOG).  Cuts the
        arena into buckets that a reverse sweep of the tape completes one after the other (parallel.plan_buckets)."""
        first = {}
        for idx, prm in log:
            if idx >= 0:
                first[id(prm)] = min(first.get(id(prm), idx), idx)
        fc = [first.get(id(prm)) for prm in arena.params]
        return par.plan_buckets(fc, arena.offsets, arena.sizes, arena.total, int(getattr(self, 'bucket_elems', BUCKET_ELEMS)))

    def close(self):
        """Release the captured hipGraphs now (they are re-captured if train_step is called again)."""
        for g in (self._segments or []):
            g.release()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _loss_scale():
    """INITIAL loss scale of the fp16-storage path (a power of two: scaling and un-scaling are exact in fp32; the
    activation gradients, stored fp16, stay clear of the subnormal range).  From there the scale is dynamic, as apex amp's
    is in the reference (tools/train_video_contrast_dis.py:134-141,413-418): a step whose gradients hold an inf / nan is
    skipped and halves it, 2000 clean steps double it -- all decided on the device (ops.grad_unscale_clip).  None for
    fp32 storage."""
    if not ops.ACT_F16[0]:
        return None
    s = float(os.environ.get('GCA_LOSS_SCALE', '1024'))
    if s <= 0 or math.frexp(s)[0] != 0.5:
        raise ValueError('GCA_LOSS_SCALE must be a positive power of two, got %r' % s)
    return s


class MoCoTrainer(_TrainerBase):
    def __init__(self, cfg, device, ctx=None, use_graph=True, seed=None):
        _check_unsupported(cfg)
        self.cfg, self.device = cfg, torch.device(device)
        self.clip = clip_value_of(cfg)
        self.ctx = ctx or par.DistCtx()
        ls = _loss_scale()
        # fp16 storage: (scale, clean steps in a row, skipped steps, steps) on the device; None for fp32 storage
        self.scale_state = None if ls is None else ops.loss_scale_state(ls, torch.device(device))
        self.loss_scale = 1.0 if ls is None else ls           # the INITIAL scale (the live one is scale_state[0], on the device)
        if seed is not None:
            torch.manual_seed(seed)
        self.model, self.model_ema = create_visual_model(cfg)
        self.model.to(self.device)
        self.model_ema.to(self.device)
        self.arena_q, self.arena_k = arena_of(self.model), arena_of(self.model_ema)
        if self.arena_q.total != self.arena_k.total:
            raise RuntimeError('query / key encoders must share one parameter layout')
        par.broadcast_(self.arena_q.flat, self.ctx)
        self.arena_k.flat.copy_(self.arena_q.flat)                       # _momentum_update(m=0), :146
        self.contrast = create_contrast(cfg, 0).to(self.device)
        par.broadcast_(self.contrast.memory, self.ctx)                    # _broadcast_memory, :121
        self.criterion = create_criterion(cfg, 0)
        self.optimizer = make_optimizer(cfg, self.model)
        self.scheduler = make_lr_scheduler(cfg, self.optimizer)
        self.model.train()
        set_key_encoder_mode(self.model_ema)
        self.alpha = float(cfg.CONTRAST.ALPHA)
        self.inv_T = 1.0 / float(cfg.CONTRAST.NCE_T)
        self.K = int(cfg.CONTRAST.NCE_K)
        self.ptr_dev = torch.zeros(1, dtype=torch.long, device=self.device)
        self.use_graph = bool(use_graph)
        self.step_count = 0
        self.perm_seed = int(getattr(cfg.MODEL, 'SEED', 1))
        self._static = None
        self._segments = None
        self._packers = None        # built after the first (eager) step, when every layer has its plan
        self._buckets = None        # N > 1: [(closure index, lo, hi)] gradient buckets in completion order
        self._tape = None
        self._planned = False
        self._plans = None          # N > 1: look-ahead ShuffleBN exchange plans (parallel.ExchangePlans)
        self._reducer = par.BucketReducer(self.arena_q.grad, self.ctx)
        self._side = None           # second stream of the captured step (key encoder forward)
        self._fwd = None            # N > 1: (tape, query feature) between the forward segment and the rest
        self.out = {}

    # -------------------------------------------------------------------------------- buffers
    def _ensure_static(self, images):
        b = images.shape[0]
        if self._static is None or self._static['images'].shape != images.shape:
            W = self.ctx.world
            D = int(self.cfg.CROSS.FEAT_DIM)
            self._static = dict(
                images=torch.empty(tuple(images.shape), dtype=torch.float32, device=self.device),
                key_in=torch.empty((b,) + (3,) + tuple(images.shape[2:]), dtype=torch.float32, device=self.device)
                if self.ctx.active else None,
                k_shuf=torch.empty((b, D), dtype=torch.float32, device=self.device),
                k=torch.empty((b, D), dtype=torch.float32, device=self.device),
                all_k=torch.empty((b * W, D), dtype=torch.float32, device=self.device),
                enq_idx=torch.empty(b, dtype=torch.long, device=self.device))
            self._segments = None
            self._packers = None
        return self._static

    def _pack(self, which):
        """Refresh the packed weights of a whole encoder in one launch (layers skip their own packing)."""
        if self._packers is not None:
            self._packers[which].run()

    def _unpack(self, which):
        if self._packers is not None:
            self._packers[which].release()

    # -------------------------------------------------------------------------------- phases
    def _phase_key(self):
        """k (shuffled order on N>1) = key_encoder(key_in); no tape, BN train mode."""
        s = self._static
        x2 = s['key_in'] if self.ctx.active else torch.chunk(s['images'], 2, dim=1)[1]
        self._pack('k')
        kv = self.model_ema.fwd(Tape(False), Var(x2))
        self._unpack('k')
        s['k_shuf'].copy_(kv.t)

    def _phase_query_fwd(self):
        """Query encoder forward (needs nothing from the key encoder)."""
        s = self._static
        self.optimizer.zero_grad()
        self._pack('q')
        tape = Tape(True)
        qv = self.model.fwd(tape, Var(torch.chunk(s['images'], 2, dim=1)[0]))
        return tape, qv

    def _phase_query_rest(self, tape, qv, upto=0):
        """InfoNCE, enqueue, dq and the backward pass down to tape index `upto` (0 = all of it; N > 1 runs the rest in stages,
        one per gradient bucket: _phase_backward_stage)."""
        s = self._static
        mem = self.contrast.memory
        logits, lse, rank, loss = ops.moco_logits_fwd(qv.t, s['k'], mem, self.inv_T, want_lse=True, want_rank=True,
                                                      want_loss=True)
        saved = ops.queue_enqueue(mem, s['all_k'], 0, save=True, ptr_dev=self.ptr_dev)
        # DDP averages gradients: fold 1/world into the loss-gradient scale
        qv.grad = ops.moco_logits_bwd(s['k'], mem, self.inv_T, logits=logits, lse=lse,
                                      gscale_dev=self.scale_state, gscale_host=1.0 / self.ctx.world, ov_rows=saved,
                                      ov_start_dev=self.ptr_dev)
        ops.queue_advance(self.ptr_dev, s['all_k'].shape[0], self.K)
        self.out = dict(loss=loss, logits=logits, rank=rank, q=qv.t)
        self._backward(tape, upto)
        self._tape = tape if upto > 0 else None
        if upto == 0:
            self._unpack('q')

    def _phase_backward_stage(self, upto, last):
        self._backward(self._tape, upto)
        if last:
            self._tape = None
            self._unpack('q')

    def _phase_update(self):
        clip = None
        if self.scale_state is not None:
            # fp16 storage: the gradients were computed x S.  One pass over the arena yields the un-scaled norm, the factor
            # (clip coefficient / S) the SGD kernel applies on the fly and the overflow verdict: inf / nan anywhere -> the
            # optimizer step is skipped (parameters and momentum untouched) and S is halved, as amp does.  The EMA below
            # still runs: the reference's _momentum_update is its own call, which amp does not patch (:440).
            clip = ops.grad_unscale_clip(self.arena_q.grad, self.scale_state, self.clip)
            self.out['grad_norm'] = clip
            self.out['loss_scale'] = self.scale_state
        elif self.clip is not None:                                           # :420-423, after the gradient all-reduce
            clip = self.optimizer.clip_grad_norm(self.clip)
            self.out['grad_norm'] = clip
        self.optimizer.step(grad_clip=clip)
        ops.ema_update(self.arena_k.flat, self.arena_q.flat, self.alpha)      # :440

    def _key_and_query_fwd(self):
        """Key encoder forward and query encoder forward (they share nothing until InfoNCE).  Inside a captured hipGraph the
        key encoder runs on a SECOND stream next to the query encoder: the latency-bound kernels of the deep, small layers
        of one fill the CUs the other leaves idle (measured +5 % on the whole step at N = 1).  Eager runs keep one stream (the
        launch tuner times kernels there) but already use the key lane's own scratch buffer, so nothing grows during the
        capture.  -> (tape, query feature Var)"""
        fork = FORK_KEY_ENCODER and torch.cuda.is_current_stream_capturing()
        ops.WS_LANE[0] = 1
        try:
            if fork:
                cur = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._phase_key()
            else:
                self._phase_key()
        finally:
            ops.WS_LANE[0] = 0
        tape, qv = self._phase_query_fwd()
        if fork:
            cur.wait_stream(self._side)
        return tape, qv

    def _single_gpu_all(self):
        s = self._static
        tape, qv = self._key_and_query_fwd()
        # N=1: BN batch statistics are permutation invariant, so the clips are NOT physically shuffled;
        # only the enqueue order (all_k = k in shuffled order, :222) is reproduced.
        s['k'].copy_(s['k_shuf'])
        s['all_k'].copy_(ops.gather_rows(s['k_shuf'], s['enq_idx']))
        self._phase_query_rest(tape, qv, 0)
        self._phase_update()

    def _phase_fwd_pair(self):
        """N > 1, first segment: both encoder forwards; the tape lives on until the next segment (after the key gather)."""
        self._fwd = self._key_and_query_fwd()

    def _phase_rest(self, upto):
        (tape, qv), self._fwd = self._fwd, None
        self._phase_query_rest(tape, qv, upto)

    # -------------------------------------------------------------------------------- step
    def train_step(self, images, shuffle_ids=None):
        """images: (b, 6, T, H, W) fp32 on the device, or a StagedBatch of engine.input.DeviceInputStage (uint8 frames on
        their way to the device: the crop / flip / normalise / layout pass then writes the static batch directly).
        Returns dict(loss, logits, rank, q) of device tensors (no host sync).  shuffle_ids: optional host int64 permutation
        of the node batch."""
        staged = images if isinstance(images, StagedBatch) else None
        if staged is not None:
            images = _ShapeOnly(staged.stage.out_shape(), self.device)
        elif images.device != self.device or images.dtype != torch.float32:
            raise RuntimeError('train_step needs fp32 clips already resident on %s (or a StagedBatch)' % self.device)
        s = self._ensure_static(images)
        b, W = images.shape[0], self.ctx.world
        user_ids = shuffle_ids
        if shuffle_ids is None and not self.ctx.active:
            shuffle_ids = par.shared_permutation(b * W, self.perm_seed, self.step_count)
        if staged is not None:
            staged.stage.prepare(staged, s['images'])
        else:
            s['images'].copy_(images)
        self.optimizer._sync_tables()
        if not self.ctx.active:
            s['enq_idx'].copy_(shuffle_ids)
            if self._segments is None:
                self._segments = [_Graphed(self._single_gpu_all, self.use_graph)]
            self._segments[0].run()
        else:
            if self._plans is None or self._plans.b != b:
                self._plans = par.ExchangePlans(b, self.ctx, self.perm_seed, self.device)
            plan = self._plans.get(self.step_count, user_ids)
            if self._segments is None:
                # One graph segment per stage: key forward || query forward (two streams) | InfoNCE + backward down to the first bucket
                # boundary | one backward stage per further gradient bucket | update.  The all-reduce of a bucket is
                # issued behind the segment that completes it and runs on RCCL's stream while the next stage computes
                # (DDP's overlap, tools/train_video_contrast_dis.py:143,419, with a few 32 MB buckets).  Activations
                # allocated while one segment is captured are read by the next: the segments share one graph memory pool.
                pool = torch.cuda.graph_pool_handle() if self.use_graph else None
                if self._buckets is None:
                    self._buckets = [(0, 0, self.arena_q.total)]         # first step: un-staged; planned below
                bk = self._buckets
                last = len(bk) - 1
                segs = [_Graphed(self._phase_fwd_pair, self.use_graph, pool),
                        _Graphed(lambda u=(0 if last == 0 else bk[0][0]): self._phase_rest(u), self.use_graph, pool)]
                for i in range(1, len(bk)):
                    segs.append(_Graphed(lambda u=(0 if i == last else bk[i][0]), l=(i == last): self._phase_backward_stage(u, l),
                                         self.use_graph, pool))
                segs.append(_Graphed(self._phase_update, self.use_graph, pool))
                self._segments = segs
            x2 = torch.chunk(s['images'], 2, dim=1)[1]
            s['key_in'].copy_(par.shuffle_exchange_planned(x2, plan, self.ctx, ops.gather_rows))
            self._segments[0].run()
            s['all_k'].copy_(par.gather_keys(s['k_shuf'], self.ctx))
            s['k'].copy_(ops.gather_rows(s['all_k'], plan['unshuffle_idx']))
            bk = self._buckets
            be = int(getattr(self, 'bucket_elems', BUCKET_ELEMS))
            planning = len(bk) == 1 and not self._planned and 0 < be < self.arena_q.total
            log = None
            if planning:
                if L.GRAD_LOG is not None:
                    raise RuntimeError('another trainer is planning its gradient buckets (engine.layers.GRAD_LOG is live)')
                L.GRAD_LOG = log = []
            try:
                for i, (_, lo, hi) in enumerate(bk):
                    self._segments[1 + i].run()
                    self._reducer.launch(lo, hi)
            finally:
                if planning:
                    L.GRAD_LOG = None        # whatever happened in the step, the log never outlives it
            self._reducer.wait()
            self._segments[-1].run()
            if user_ids is None:
                self._plans.prefetch(self.step_count + 1)        # host index work of the NEXT step, while this one runs
            if planning:
                # the first step ran un-staged with the gradient log on: cut the arena into buckets by the closure that
                # completes each and rebuild the segments, so that every later step overlaps all-reduce and backward
                self._buckets = self._buckets_from_log(self.arena_q, log)
                self._planned = True             # only once the buckets exist: a step that raised is planned again
                for g in self._segments:
                    g.release()
                self._segments = None
        if self._packers is None:
            self._packers = dict(k=BatchedPacker(self.model_ema, (0,)), q=BatchedPacker(self.model, (0, 1)))
        self.contrast.index = (self.contrast.index + b * W) % self.K      # host mirror of ptr_dev
        self.step_count += 1
        return self.out

    def state_dict(self, epoch=0):
        """Checkpoint dict with the reference's keys (tools/...dis.py:274-286) + the queue pointer."""
        return {'epoch': epoch, 'state_dict': self.model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                'contrast': self.contrast.state_dict(), 'model_ema': self.model_ema.state_dict(),
                'queue_index': int(self.contrast.index), 'step_count': int(self.step_count),
                **({} if self.scale_state is None else {'loss_scale_state': self.scale_state.detach().cpu().clone()})}

    def load_state_dict(self, sd):
        """Resume from state_dict() -- or from a reference checkpoint, which lacks `queue_index` / `step_count`
        (the reference restarts the queue pointer at 0 on resume; so do we for such a file)."""
        self.model.load_state_dict(sd['state_dict'])
        self.model_ema.load_state_dict(sd['model_ema'])
        self.optimizer.load_state_dict(sd['optimizer'])
        self.contrast.load_state_dict(sd['contrast'])
        self.contrast.index = int(sd.get('queue_index', 0)) % self.K
        self.ptr_dev.fill_(self.contrast.index)
        self.step_count = int(sd.get('step_count', 0))
        if self.scale_state is not None and sd.get('loss_scale_state') is not None:
            self.scale_state.copy_(sd['loss_scale_state'])
        return int(sd.get('epoch', 0))


class SimSiamTrainer(_TrainerBase):
    def __init__(self, cfg, device, ctx=None, use_graph=True, seed=None):
        _check_unsupported(cfg)
        self.cfg, self.device = cfg, torch.device(device)
        self.clip = clip_value_of(cfg)
        self.ctx = ctx or par.DistCtx()
        ls = _loss_scale()
        self.scale_state = None if ls is None else ops.loss_scale_state(ls, torch.device(device))
        if seed is not None:
            torch.manual_seed(seed)
        self.model, ema = create_visual_model(cfg)
        if ema is not None:
            raise ValueError('SimSiamTrainer needs CONTRAST.MEM_TYPE == simsiam')
        self.model.to(self.device)
        self.arena = arena_of(self.model)
        par.broadcast_(self.arena.flat, self.ctx)
        self.optimizer = make_optimizer(cfg, self.model)
        self.scheduler = make_lr_scheduler(cfg, self.optimizer)
        self.model.train()
        self.use_graph = bool(use_graph)
        self._static, self._segments, self._packer, self.out = None, None, None, {}
        self._buckets, self._planned, self._tape = None, False, None     # N > 1: gradient buckets (see MoCoTrainer)
        self._reducer = par.BucketReducer(self.arena.grad, self.ctx)

    def _fwd_bwd(self, upto=0):
        """Forward, loss and the backward pass down to tape index `upto` (0 = all of it; N > 1 runs the rest in stages)."""
        self.optimizer.zero_grad()
        if self._packer is not None:
            self._packer.run()
        tape = Tape(True)
        lv = self.model.fwd(tape, Var(self._static))
        self.out = dict(loss=lv.t)
        ops.LOSS_SCALE_STATE[0] = self.scale_state            # fp16 storage: SimSiam.fwd's loss-gradient seed is scaled by S
        try:
            self._backward(tape, upto)
        finally:
            ops.LOSS_SCALE_STATE[0] = None
        self._tape = tape if upto > 0 else None
        if upto == 0 and self._packer is not None:
            self._packer.release()

    def _bwd_stage(self, upto, last):
        ops.LOSS_SCALE_STATE[0] = self.scale_state
        try:
            self._backward(self._tape, upto)
        finally:
            ops.LOSS_SCALE_STATE[0] = None
        if last:
            self._tape = None
            if self._packer is not None:
                self._packer.release()

    def _update(self):
        if self.ctx.active:
            ops.scale_(self.arena.grad, 1.0 / self.ctx.world)      # DDP's mean (the loss gradient is seeded inside the model)
        clip = None
        if self.scale_state is not None:                       # fp16 storage: un-scale + overflow check + clip, see MoCoTrainer
            clip = ops.grad_unscale_clip(self.arena.grad, self.scale_state, self.clip)
            self.out['grad_norm'] = clip
            self.out['loss_scale'] = self.scale_state
        elif self.clip is not None:                            # tools/train_video_contrast_dis.py:497-500
            clip = self.optimizer.clip_grad_norm(self.clip)
            self.out['grad_norm'] = clip
        self.optimizer.step(grad_clip=clip)

    def train_step(self, images):
        """images: (b, 6, T, H, W) fp32 on the device, or a StagedBatch (engine.input.DeviceInputStage)."""
        staged = images if isinstance(images, StagedBatch) else None
        shape = torch.Size(staged.stage.out_shape()) if staged is not None else images.shape
        if self._static is None or self._static.shape != shape:
            self._static = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
            self._packer = None
            self._segments = None
        if staged is not None:
            staged.stage.prepare(staged, self._static)
        else:
            self._static.copy_(images)
        self.optimizer._sync_tables()
        if not self.ctx.active:
            if self._segments is None:
                self._segments = [_Graphed(self._fwd_bwd, self.use_graph), _Graphed(self._update, self.use_graph)]
            self._segments[0].run()
            self._segments[1].run()
        else:
            # staged backward, one graph segment per gradient bucket, all-reduce of a bucket overlapped with the stages
            # below it (as MoCoTrainer.train_step); the first step runs un-staged with the gradient log on and plans the buckets
            if self._segments is None:
                pool = torch.cuda.graph_pool_handle() if self.use_graph else None
                if self._buckets is None:
                    self._buckets = [(0, 0, self.arena.total)]
                bk = self._buckets
                last = len(bk) - 1
                segs = [_Graphed(lambda u=(0 if last == 0 else bk[0][0]): self._fwd_bwd(u), self.use_graph, pool)]
                for i in range(1, len(bk)):
                    segs.append(_Graphed(lambda u=(0 if i == last else bk[i][0]), l=(i == last): self._bwd_stage(u, l),
                                         self.use_graph, pool))
                segs.append(_Graphed(self._update, self.use_graph, pool))
                self._segments = segs
            bk = self._buckets
            be = int(getattr(self, 'bucket_elems', BUCKET_ELEMS))
            planning = len(bk) == 1 and not self._planned and 0 < be < self.arena.total
            log = None
            if planning:
                if L.GRAD_LOG is not None:
                    raise RuntimeError('another trainer is planning its gradient buckets (engine.layers.GRAD_LOG is live)')
                L.GRAD_LOG = log = []
            try:
                for i, (_, lo, hi) in enumerate(bk):
                    self._segments[i].run()
                    self._reducer.launch(lo, hi)
            finally:
                if planning:
                    L.GRAD_LOG = None
            self._reducer.wait()
            self._segments[-1].run()
            if planning:
                self._buckets = self._buckets_from_log(self.arena, log)
                self._planned = True
                for g in self._segments:
                    g.release()
                self._segments = None
        if self._packer is None:
            self._packer = BatchedPacker(self.model, (0, 1))
        return self.out
