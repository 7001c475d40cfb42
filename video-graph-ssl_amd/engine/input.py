"""Device-side input stage: uint8 frames in, normalised (b, 6, T, H, W) clips out, H2D overlapped with the step.

The reference decodes and augments on the host and copies the finished fp32 batch to the GPU inside the iteration
(tools/train_video_contrast_dis.py:402, 154 MB per 32-clip batch).  At ~21 ms per iteration that copy is worth 15 % of a
step.  Here the host keeps what only it can do (JPEG decode, resize, colour jitter, blur: cv2 / albumentations work,
lib/data/transform/build.py:45-62) and hands over uint8 frames + one (crop origin, flip) record per clip view; the rest --
crop, horizontal flip, VideoNormalize, VideoToTensor, the concatenation of the two views
(lib/data/datasets/video_contrast_dataset.py:196-203) -- is ONE kernel pass (gca_clip_prepare) writing straight into the
trainer's static input buffer.

Pipelining: two pinned host buffers and two device buffers; ``stage()`` copies batch t+1 on a copy stream while the
captured step of batch t runs; ``prepare()`` makes the compute stream wait for that copy (an event, no host sync) and
launches the kernel.  A slot is reused only after the kernel that read it has been issued and its event has passed.
"""
import numpy as np
import torch

from . import ops
from .. import _hip as H


def normalize_constants(mean, std, max_pixel_value=255.0):
    """(mean*255, 1/(std*255)) in fp32, rounded where VideoNormalize.normalize rounds (consistency_transforms.py:54-60)."""
    m = np.array(mean, dtype=np.float32)
    m *= max_pixel_value
    s = np.array(std, dtype=np.float32)
    s *= max_pixel_value
    return m, np.reciprocal(s, dtype=np.float32)


def crop_coords(height, width, crop_height, crop_width, h_start, w_start):
    """albumentations' get_random_crop_coords, as F.random_crop(img, h, w, h_start, w_start) uses it (VideoRandomCrop,
    the crop step of VideoRandomResizedCrop): fractions in [0, 1) -> integer origin."""
    return int((height - crop_height) * h_start), int((width - crop_width) * w_start)


def clip_prepare(frames, params, mean255, inv_std255, H_out, W_out, out=None, out_dtype=torch.float32):
    """frames (b, views, T, Hs, Ws, 3) uint8 device tensor, params (b, views, 4) int32 device tensor {h0, w0, flip, 0}
    -> (b, 3*views, T, H_out, W_out) fp32 | fp16.  `out`: optional preallocated result (the trainer's static batch)."""
    if frames.dtype is not torch.uint8 or frames.dim() != 6 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise ValueError('frames must be a contiguous (b, views, T, Hs, Ws, 3) uint8 tensor')
    if not frames.is_cuda:
        raise RuntimeError('clip_prepare needs the frames on the GPU (there is no CPU fallback)')
    b, views, T, Hs, Ws, _ = frames.shape
    if params.dtype is not torch.int32 or tuple(params.shape) != (b, views, 4) or not params.is_contiguous():
        raise ValueError('params must be a contiguous (b, views, 4) int32 tensor')
    if out is None:
        out = torch.empty((b, 3 * views, T, H_out, W_out), dtype=out_dtype, device=frames.device)
    elif tuple(out.shape) != (b, 3 * views, T, H_out, W_out) or not out.is_contiguous() or out.dtype not in (torch.float32, torch.float16):
        raise ValueError('out must be a contiguous (b, 3*views, T, H, W) fp32 / fp16 tensor')
    m = np.ascontiguousarray(mean255, dtype=np.float32)
    d = np.ascontiguousarray(inv_std255, dtype=np.float32)
    H.call('gca_clip_prepare', frames.data_ptr(), b, views, T, Hs, Ws, params.data_ptr(), m.ctypes.data, d.ctypes.data,
           H_out, W_out, out.data_ptr(), int(out.dtype is torch.float16), ops.stream())
    return out


class StagedBatch(object):
    """One batch on its way to the GPU: device uint8 frames + params and the event that marks the end of its copy."""
    __slots__ = ('frames', 'params', 'ready', 'slot', 'stage')

    def __init__(self, stage, slot, frames, params, ready):
        self.stage, self.slot, self.frames, self.params, self.ready = stage, slot, frames, params, ready


class DeviceInputStage(object):
    def __init__(self, batch, frames, src_size, out_size, device, views=2, mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), slots=2):
        self.b, self.views, self.T = int(batch), int(views), int(frames)
        self.Hs, self.Ws = (src_size, src_size) if isinstance(src_size, int) else tuple(src_size)
        self.H, self.W = (out_size, out_size) if isinstance(out_size, int) else tuple(out_size)
        if self.H > self.Hs or self.W > self.Ws:
            raise ValueError('crop %r larger than the source frames %r' % ((self.H, self.W), (self.Hs, self.Ws)))
        self.device = torch.device(device)
        self.mean255, self.inv_std255 = normalize_constants(mean, std)
        shape = (self.b, self.views, self.T, self.Hs, self.Ws, 3)
        self._host = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self._hostp = [torch.empty((self.b, self.views, 4), dtype=torch.int32).pin_memory() for _ in range(slots)]
        self._dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(slots)]
        self._devp = [torch.empty((self.b, self.views, 4), dtype=torch.int32, device=self.device) for _ in range(slots)]
        self._consumed = [None] * slots          # event recorded on the compute stream after the slot's kernel was issued
        self._copied = [None] * slots            # event of the slot's last H2D copy (its pinned buffer is free after it)
        self._next = 0
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.frame_bytes = int(np.prod(shape))

    def out_shape(self):
        return (self.b, 3 * self.views, self.T, self.H, self.W)

    def acquire(self):
        """-> (frames, params): the pinned host buffers of the next slot -- (b, views, T, Hs, Ws, 3) uint8 and (b, views, 4)
        int32 {h0, w0, flip, 0} -- for the loader to fill IN PLACE (decoded frames land in pinned memory once; no second host
        copy).  Blocks only if the slot's previous H2D copy is still in flight.  Follow with submit()."""
        s = self._next
        if self._copied[s] is not None:
            self._copied[s].synchronize()            # the pinned buffers of this slot are about to be overwritten by the host
        return self._host[s], self._hostp[s]

    def submit(self, check=True):
        """Start the asynchronous H2D copy of the slot handed out by the last acquire(); returns a StagedBatch."""
        s = self._next
        self._next = (s + 1) % len(self._host)
        if check:
            p = self._hostp[s]
            if (int(p[..., 0].min()) < 0 or int(p[..., 1].min()) < 0 or int(p[..., 0].max()) > self.Hs - self.H
                    or int(p[..., 1].max()) > self.Ws - self.W):
                raise ValueError('crop window outside the source frame')
        with torch.cuda.stream(self.copy_stream):
            if self._consumed[s] is not None:
                self.copy_stream.wait_event(self._consumed[s])    # the kernel that read this device slot has been issued and passed
            self._dev[s].copy_(self._host[s], non_blocking=True)
            self._devp[s].copy_(self._hostp[s], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._copied[s] = ready
        return StagedBatch(self, s, self._dev[s], self._devp[s], ready)

    def stage(self, frames, params):
        """Convenience for callers that hold the batch elsewhere: frames (b, views, T, Hs, Ws, 3) uint8 host tensor / ndarray,
        params (b, views, >=3) integers {h0, w0, flip}.  = acquire() + one host copy into the pinned slot + submit()."""
        f = torch.as_tensor(frames)
        p = torch.as_tensor(np.asarray(params))
        if tuple(f.shape) != tuple(self._host[0].shape) or f.dtype is not torch.uint8:
            raise ValueError('frames must be uint8 of shape %r, got %s %r' % (tuple(self._host[0].shape), f.dtype, tuple(f.shape)))
        if p.dim() != 3 or tuple(p.shape[:2]) != (self.b, self.views) or p.shape[2] < 3:
            raise ValueError('params must be (b, views, >=3): h0, w0, flip')
        hf, hp = self.acquire()
        hf.copy_(f)
        hp.zero_()
        hp[..., :3].copy_(p[..., :3].to(torch.int32))
        return self.submit()

    def prepare(self, staged, out):
        """Compute stream: wait for the copy, then crop + flip + normalise + layout change into `out`."""
        if staged.stage is not self:
            raise ValueError('batch was staged by another DeviceInputStage')
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(staged.ready)
        clip_prepare(staged.frames, staged.params, self.mean255, self.inv_std255, self.H, self.W, out=out)
        done = torch.cuda.Event()
        done.record(cur)
        self._consumed[staged.slot] = done
        return out
