"""Batched-image container with the reference's surface (mega_core/structures/image_list.py:7-72):
`ImageList(tensors, image_sizes)`, `.to(...)`, and `to_image_list(x, size_divisible)`.

Semantics kept: a list/tuple of CHW tensors is zero-padded (bottom/right) to one common shape whose
H and W are rounded up to `size_divisible`; `image_sizes` records the un-padded (h, w) of every
image -- the detector reads `image_sizes[0]` for the box scale and the result BoxList size, while
the backbone sees the padded tensor (diffusion_det.py:444, :529, :613).
"""
import torch
import torch.nn.functional as F


class ImageList:
    __slots__ = ("tensors", "image_sizes")

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def _round_up(v, m):
    return v if m <= 0 else -(-v // m) * m


def is_image_list(x):
    """An ImageList by its surface, not its class: the reference's collator wraps frames in
    `mega_core.structures.image_list.ImageList` (collate_batch.py:24-35) and those objects arrive here unchanged."""
    return (not isinstance(x, torch.Tensor)) and hasattr(x, "tensors") and hasattr(x, "image_sizes")


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, ImageList):
        return tensors
    if is_image_list(tensors):              # a foreign ImageList: same tensor, same sizes, this repo's wrapper
        return ImageList(tensors.tensors, list(tensors.image_sizes))
    if isinstance(tensors, torch.Tensor):
        if size_divisible > 0:
            tensors = [tensors]             # single image that still needs alignment padding
        else:
            batch = tensors if tensors.dim() == 4 else tensors.unsqueeze(0)
            if batch.dim() != 4:
                raise AssertionError("expected a CHW or NCHW tensor")
            return ImageList(batch, [t.shape[-2:] for t in batch])
    if not isinstance(tensors, (tuple, list)):
        raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
    sizes = [t.shape[-2:] for t in tensors]
    out_h = _round_up(max(s[0] for s in sizes), size_divisible)
    out_w = _round_up(max(s[1] for s in sizes), size_divisible)
    padded = [F.pad(t, (0, out_w - t.shape[-1], 0, out_h - t.shape[-2])) for t in tensors]
    return ImageList(torch.stack(padded, dim=0), sizes)
