"""BoxList result container -- API mirror of mega_core/structures/bounding_box.py:9-250: bbox, size=(w,h), mode, extra
fields, clip_to_image, to, indexing, and the geometric methods the reference's callers use on detections and ground truth
(`resize` -- do_vid_evaluation, evaluation/vid/vid_eval.py:17-21 --, `transpose`, `crop`, `area`, `copy_with_fields`).
Pinned against the reference's own class by golden g17 (tests/test_oracle_golden.py)."""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


def _carry_fields(src, dst, apply):
    """extra fields of `src` onto `dst`: tensors as they are, box-aligned objects (masks, keypoints) through `apply`
    (bounding_box.py:104-107)"""
    for name, value in src.extra_fields.items():
        dst.add_field(name, value if isinstance(value, torch.Tensor) else apply(value))
    return dst


_MODES = ("xyxy", "xywh")


def _same_kind(src, boxes):
    """a BoxList of `src`'s image size and mode around other box rows"""
    return BoxList(boxes, src.size, src.mode)


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        # same contract and messages as the reference's constructor (bounding_box.py:20-37): fp32 [n, 4] rows on the device they came from
        on = bbox.device if torch.is_tensor(bbox) else torch.device("cpu")
        rows = torch.as_tensor(bbox, dtype=torch.float32, device=on)
        problem = None
        if rows.dim() != 2:
            problem = "bbox should have 2 dimensions, got {}".format(rows.dim())
        elif rows.shape[-1] != 4:
            problem = "last dimension of bbox should have a size of 4, got {}".format(rows.shape[-1])
        elif mode not in _MODES:
            problem = "mode should be 'xyxy' or 'xywh'"
        if problem:
            raise ValueError(problem)
        self.bbox, self.size, self.mode = rows, image_size, mode          # size = (image_width, image_height)
        self.extra_fields = {}

    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, bbox):
        self.extra_fields.update(bbox.extra_fields)

    def _split_into_xyxy(self):
        """four [n, 1] corner columns whatever the mode (xywh widths count pixels: TO_REMOVE = 1, bounding_box.py:76-89)"""
        x1, y1, a, b = self.bbox.split(1, dim=-1)
        if self.mode == "xyxy":
            return x1, y1, a, b
        return x1, y1, x1 + (a - 1).clamp(min=0), y1 + (b - 1).clamp(min=0)

    def convert(self, mode):
        if mode not in _MODES:
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        x1, y1, x2, y2 = self._split_into_xyxy()
        if mode == "xywh":      # from xyxy (TO_REMOVE = 1 convention of the reference)
            out = torch.cat((x1, y1, x2 - x1 + 1, y2 - y1 + 1), dim=-1)
        else:
            out = torch.cat((x1, y1, x2, y2), dim=-1)
        bl = BoxList(out, self.size, mode=mode)
        bl._copy_extra_fields(self)
        return bl

    def resize(self, size, *args, **kwargs):
        """copy scaled to an image of `size` = (width, height) (bounding_box.py:91-127).  With one common ratio the
        stored numbers are scaled as they are (xywh widths included); otherwise corners are scaled per axis and the
        result goes back to the mode it came from."""
        rw, rh = (float(new) / float(old) for new, old in zip(size, self.size))
        if rw == rh:
            out = BoxList(self.bbox * rw, size, mode=self.mode)
            return _carry_fields(self, out, lambda v: v.resize(size, *args, **kwargs))
        x1, y1, x2, y2 = self._split_into_xyxy()
        out = BoxList(torch.cat((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=-1), size, mode="xyxy")
        return _carry_fields(self, out, lambda v: v.resize(size, *args, **kwargs)).convert(self.mode)

    def transpose(self, method):
        """horizontal / vertical flip (bounding_box.py:129-165; the horizontal one counts pixels, the vertical one does not)"""
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        width, height = self.size
        x1, y1, x2, y2 = self._split_into_xyxy()
        if method == FLIP_LEFT_RIGHT:
            x1, x2 = width - x2 - 1, width - x1 - 1
        else:
            y1, y2 = height - y2, height - y1
        out = BoxList(torch.cat((x1, y1, x2, y2), dim=-1), self.size, mode="xyxy")
        return _carry_fields(self, out, lambda v: v.transpose(method)).convert(self.mode)

    def crop(self, box):
        """boxes relative to the window `box` = (left, top, right, bottom), clamped to it; empty boxes stay
        (bounding_box.py:167-193)"""
        w, h = box[2] - box[0], box[3] - box[1]
        x1, y1, x2, y2 = self._split_into_xyxy()
        cols = ((x1 - box[0]).clamp(min=0, max=w), (y1 - box[1]).clamp(min=0, max=h),
                (x2 - box[0]).clamp(min=0, max=w), (y2 - box[1]).clamp(min=0, max=h))
        out = BoxList(torch.cat(cols, dim=-1), (w, h), mode="xyxy")
        return _carry_fields(self, out, lambda v: v.crop(box)).convert(self.mode)

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    def copy_with_fields(self, fields, skip_missing=False):
        wanted = fields if isinstance(fields, (list, tuple)) else [fields]
        missing = [f for f in wanted if f not in self.extra_fields]
        if missing and not skip_missing:
            raise KeyError("Field '{}' not found in {}".format(missing[0], self))
        out = _same_kind(self, self.bbox)
        out.extra_fields.update({f: self.extra_fields[f] for f in wanted if f in self.extra_fields})
        return out

    def _mapped(self, boxes, pick):
        out = _same_kind(self, boxes)
        out.extra_fields.update({name: pick(value) for name, value in self.extra_fields.items()})
        return out

    def to(self, device):
        """boxes and every field that can move (bounding_box.py:197-204)"""
        return self._mapped(self.bbox.to(device), lambda v: v.to(device) if hasattr(v, "to") else v)

    def __getitem__(self, item):
        """row selection, applied to every field alike (bounding_box.py:206-210)"""
        return self._mapped(self.bbox[item], lambda v: v[item])

    def __len__(self):
        return int(self.bbox.shape[0])

    def clip_to_image(self, remove_empty=True):
        """in place: corners into [0, width - 1] x [0, height - 1] (pixel-counting convention, bounding_box.py:215-225); with
        `remove_empty` the boxes that keep a positive width and height are returned"""
        limit = (self.size[0] - 1, self.size[1] - 1)
        for col in range(4):
            self.bbox[:, col].clamp_(min=0, max=limit[col & 1])
        if not remove_empty:
            return self
        b = self.bbox
        return self[(b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])]

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)
