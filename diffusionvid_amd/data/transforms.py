"""Test-time image transform of the DiffusionVID path: Resize(min 600, max 1000) + ToTensor.

Reference: mega_core/data/transforms/build.py:89-97 (test pipeline for METHOD "diffusion": Resize, ToTensor, no Normalize --
the model normalises itself, diffusion_det.py:301-303) and transforms.py:31-70 (`Resize`: the size is chosen from the
CURRENT frame and re-used for its reference frames; `torchvision.transforms.functional.resize` on a PIL image, i.e.
`PIL.Image.resize(..., BILINEAR)`).

Pillow's bilinear resize is a separable triangle filter whose support grows with the down-scaling factor, evaluated in
8-bit fixed point: per axis, double-precision weights normalised to sum 1 and rounded to 22-bit integers, an integer
dot product per output sample, rounding and clamping to uint8 BETWEEN the horizontal and the vertical pass (Pillow
src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc).
`resample_tables` builds those integer tables on the host; the two passes themselves run either in numpy
(`resize_u8_numpy`, small images / tests) or on the GPU (csrc/resize.hip through ops.resize_u8_to_f32), bit-identical
to Pillow's output, so that frames can travel to the GPU as uint8 at their native size (a 1280x720 JPEG frame is 2.8 MB
instead of 7.5 MB of padded fp32) and be resized, scaled to [0,1] and zero-padded there.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def get_size(image_size_wh, min_size=600, max_size=1000):
    """transforms.py:39-59: (w, h) of the source -> (oh, ow); shorter side to `min_size` unless that pushes the longer side
    beyond `max_size`."""
    w, h = image_size_wh
    size = min_size
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        return (int(size * h / w), size)
    return (size, int(size * w / h))


def resample_tables(in_size, out_size):
    """One axis of Pillow's BILINEAR resample: -> (bounds int32 [out, 2] = (first source sample, count), weights int32
    [out, ksize]) in 22-bit fixed point."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * inv)) for x in range(xmax)], dtype=np.float64)
        total = 0.0
        for v in w:                      # left-to-right double accumulation, as the C loop does
            total += float(v)
        if total != 0.0:
            w = w / total
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)
        kk[xx, :xmax] = q.astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass_numpy(img, bounds, kk, axis):
    """one integer resample pass along `axis` (0: rows / vertical, 1: columns / horizontal) of a uint8 HWC array"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        a, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[o, :n].astype(np.int64), src[a:a + n], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8_numpy(img_hwc, out_hw):
    """Pillow-identical BILINEAR resize of a uint8 HWC image (horizontal pass, then vertical; a pass whose size does not
    change is skipped, as in ImagingResample)."""
    h, w = img_hwc.shape[:2]
    oh, ow = out_hw
    out = np.ascontiguousarray(img_hwc)
    if ow != w:
        out = _pass_numpy(out, *resample_tables(w, ow), axis=1)
    if oh != h:
        out = _pass_numpy(out, *resample_tables(h, oh), axis=0)
    return out


def _as_u8_hwc(image):
    """PIL image / numpy HWC / torch HWC uint8 -> contiguous numpy uint8 [H, W, 3] (RGB)"""
    if isinstance(image, torch.Tensor):
        image = image.cpu().numpy()
    if not isinstance(image, np.ndarray):
        image = np.asarray(image.convert("RGB"))
    assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3
    return np.ascontiguousarray(image)


class ResizeToTensor:
    """Resize + ToTensor on the HOST with Pillow itself -- the reference's path; frames come out as fp32 CHW in [0, 1]."""

    def __init__(self, min_size=600, max_size=1000):
        self.min_size, self.max_size = min_size, max_size
        self.size = None

    def __call__(self, image, is_current=False):
        from PIL import Image
        if not hasattr(image, "resize") or isinstance(image, (np.ndarray, torch.Tensor)):
            image = Image.fromarray(_as_u8_hwc(image))
        if is_current or self.size is None:
            self.size = get_size(image.size, self.min_size, self.max_size)
        oh, ow = self.size
        out = np.array(image.convert("RGB").resize((ow, oh), Image.BILINEAR))          # a writable copy
        return torch.from_numpy(out).permute(2, 0, 1).to(torch.float32).div(255)


class ResizeToTensorDevice:
    """The same transform with the resample on the GPU: the uint8 frame is uploaded at its native size and leaves the
    kernels as fp32 CHW in [0, 1], zero-padded to `size_divisible` (what to_image_list would do next), bit-identical to
    ResizeToTensor + padding."""

    def __init__(self, device, min_size=600, max_size=1000, size_divisible=32):
        self.device = torch.device(device)
        self.min_size, self.max_size, self.div = min_size, max_size, size_divisible
        self.size = None
        self._tables = {}
        self._pinned_ring = {}

    def _pinned(self, shape, depth=4):
        """next [pinned buffer, event of its last upload] of a ring of `depth` staging buffers for frames of this shape"""
        ring = self._pinned_ring.setdefault(shape, {"slots": [], "next": 0})
        if len(ring["slots"]) < depth:
            ring["slots"].append([torch.empty(shape, dtype=torch.uint8).pin_memory(), None])
            return ring["slots"][-1]
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % depth
        return slot

    def tables(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._tables:
            b, k = resample_tables(in_size, out_size)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
        return self._tables[key]

    def __call__(self, image, is_current=False):
        from .. import ops
        if isinstance(image, torch.Tensor) and image.is_cuda:
            src = image
        else:
            # decoded frame -> a pinned, writable staging buffer of its size (kept per size) -> async H2D.  (A from_numpy view
            # of the decoder's read-only buffer is pageable: its "non_blocking" upload only works because the runtime stages it
            # synchronously.)  The staging buffer is reused, so the copy engine must have read it before the next frame lands.
            u8 = _as_u8_hwc(image)
            slot = self._pinned(u8.shape)
            if slot[1] is not None:
                slot[1].synchronize()          # four uploads back: long done in practice
            slot[0].numpy()[...] = u8
            src = slot[0].to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(self.device))
        h, w = src.shape[:2]
        if is_current or self.size is None:
            self.size = get_size((w, h), self.min_size, self.max_size)
        oh, ow = self.size
        d = self.div
        ph, pw = (-(-oh // d) * d, -(-ow // d) * d) if d else (oh, ow)
        tx = self.tables(w, ow) if ow != w else None
        ty = self.tables(h, oh) if oh != h else None
        out = ops.resize_u8_to_f32(src, oh, ow, ph, pw, tx, ty)
        out.image_size = (oh, ow)
        return out
