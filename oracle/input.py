"""Oracle for the device-side input stage (gca_clip_prepare).  Test infrastructure -- see oracle/__init__.py.

numpy restatement of the tail of the reference's sample construction:

* ``video_normalize``    -- VideoNormalize.normalize, lib/data/transform/consistency_transforms.py:53-65: fp32 mean*255,
                            fp32 reciprocal of std*255, ``img.astype(float32); img -= mean; img *= denominator``.
* ``video_to_tensor``    -- VideoToTensor.__call__, :20-25 (3D branch): stack T frames on a 4th axis (H,W,C,T), transpose to
                            (C,T,H,W), contiguous.
* ``hflip``              -- VideoRandomHorizontalFlip's branch (:"return [F.hflip_cv2(img) ...]"): albumentations'
                            ``hflip_cv2`` is ``cv2.flip(img, 1)`` = reverse the W axis (third-party, not under
                            /root/reference and not installed here; no version pinned by the reference; restated from its
                            published definition).
* ``random_crop_coords`` -- albumentations' ``get_random_crop_coords`` behind ``F.random_crop(img, h, w, h_start, w_start)``
                            (VideoRandomCrop, VideoRandomResizedCrop's crop step): y1 = int((H - h) * h_start),
                            x1 = int((W - w) * w_start).  Published definition, as above.
* ``make_sample``        -- VisualDataset.get_item, lib/data/datasets/video_contrast_dataset.py:196-203: both views through
                            the transform, concatenated on dim 0 -> (6, T, H, W).

Parity pin: ``video_normalize`` and ``video_to_tensor`` are checked against the REFERENCE's own classes
(tests/golden/input.npz, written by tests/golden/make_golden.py gen_input, which imports consistency_transforms.py with
inert placeholder modules for the absent cv2 / albumentations imports -- neither class touches them).  ``hflip`` and
``random_crop_coords`` restate third-party definitions and are pinned by nothing in the reference: "parity unpinned" for
those two index maps (they are integer re-orderings; the GPU path is compared with them bit for bit).
"""
import numpy as np
import torch


def normalize_constants(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), max_pixel_value=255.0):
    """-> (mean*255, 1/(std*255)) as float32 arrays, rounded exactly where VideoNormalize.normalize rounds (:54-60)."""
    m = np.array(mean, dtype=np.float32)
    m *= max_pixel_value
    s = np.array(std, dtype=np.float32)
    s *= max_pixel_value
    return m, np.reciprocal(s, dtype=np.float32)


def video_normalize(img, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    m, d = normalize_constants(mean, std)
    img = img.astype(np.float32)
    img -= m
    img *= d
    return img


def hflip(img):
    return np.ascontiguousarray(img[:, ::-1, ...])


def random_crop_coords(height, width, crop_height, crop_width, h_start, w_start):
    y1 = int((height - crop_height) * h_start)
    x1 = int((width - crop_width) * w_start)
    return y1, x1


def crop(img, y1, x1, crop_height, crop_width):
    return img[y1:y1 + crop_height, x1:x1 + crop_width]


def video_to_tensor(clips):
    a = np.stack(clips, axis=3)                                   # (H, W, C, T)
    return torch.from_numpy(a.transpose(2, 3, 0, 1)).contiguous()   # (C, T, H, W)


def make_view(frames, h0, w0, flip, H, W, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """frames: (T, Hs, Ws, 3) uint8 -> (3, T, H, W) float32: crop, flip, normalise, to-tensor (the order of build.py:45-62)."""
    out = []
    for img in frames:
        img = crop(img, h0, w0, H, W)
        if flip:
            img = hflip(img)
        out.append(video_normalize(img, mean, std))
    return video_to_tensor(out)


def make_sample(frames2, params2, H, W, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """frames2: (views, T, Hs, Ws, 3) uint8; params2: (views, >=3) ints {h0, w0, flip} -> (3*views, T, H, W) float32."""
    return torch.cat([make_view(frames2[v], int(params2[v][0]), int(params2[v][1]), bool(params2[v][2]), H, W, mean, std)
                      for v in range(len(frames2))], dim=0)


def make_batch(frames, params, H, W, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """frames: (b, views, T, Hs, Ws, 3) uint8; params: (b, views, >=3) -> (b, 3*views, T, H, W) float32 (default_collate)."""
    return torch.stack([make_sample(frames[i], params[i], H, W, mean, std) for i in range(len(frames))])
