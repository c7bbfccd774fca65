"""Host-fed frame delivery: pinned host frames -> HBM, double-buffered, overlapped with compute.

The reference times `.to(device)` of every item inside its inference loop (mega_core/engine/inference.py:29-40): frames
leave the DataLoader as fp32 CHW tensors on the host.  At ~1700 frames/s that is 12.5 GB/s of H2D traffic (7.5 MB per
padded 608x1024 fp32 frame), which only stays off the critical path if the copies run ahead of the kernels on their own
stream.  `HostFedVideo` wraps a host-resident dataset: frames sit in pinned memory; when the first item of a look-ahead
group is requested the group's frames are already in (or on their way into) one of two device staging buffers, and the
copy of the NEXT group's frames is queued on a side stream -- it starts once the kernels that still read that buffer
(two groups back) have drained, and overlaps the current group's kernels.  Items then carry device tensors (views of the
staging buffer), so the detector sees exactly what `to(device)` would have given it.
"""
import torch

from ..structures.image_list import ImageList


class HostFedVideo:
    def __init__(self, host_dataset, device, frames_per_group, cyclic=False):
        assert host_dataset.device.type == "cpu"
        self.ds = host_dataset
        self.device = torch.device(device)
        self.unit = int(frames_per_group)
        self.span = self.unit + host_dataset.max_offset + host_dataset.global_size     # frames a group's items may touch
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._host = {}
        self._buf = [None, None]
        self._ready = [None, None]          # (video, group) staged in each buffer, event of its last copy
        self._slot_of = {}
        self.h2d_bytes = 0
        self._model = None
        self._pending = None
        self.cyclic = cyclic            # after the last video the first one follows (a bench pass repeated back to back)

    def __len__(self):
        return len(self.ds)

    def pin(self):
        """materialise every video of the host dataset as ONE pinned tensor [frames, 3, H, W] (outside any timed region: this
        stands for the DataLoader workers' collated output queue); a group's window is then a single contiguous copy"""
        for v, first in enumerate(self.ds.start_index):
            n = self.ds.frame_seg_len[first]
            if v in self._host:
                continue
            sample = self.ds.frame(v, 0).tensors
            buf = torch.empty((n,) + tuple(sample.shape[1:]), dtype=sample.dtype).pin_memory()
            for f in range(n):
                buf[f].copy_(self.ds.frame(v, f).tensors[0])
            self._host[v] = buf
            self.ds._cache.clear()                   # the pageable copies are not needed any more
        return self

    def reset(self):
        """forget what is staged (a new pass over the data copies everything again)"""
        self._ready = [None, None]
        self._slot_of = {}

    def _window(self, v, g):
        n = self.ds.frame_seg_len[self.ds.start_index[v]]
        lo = g * self.unit
        hi = min(n, lo + self.unit + self.ds.max_offset)
        frames = list(range(lo, hi))
        if g == 0:
            frames = sorted(set(frames) | set(range(min(n, self.ds.global_size))))
        return frames

    def _stage(self, v, g, slot):
        frames = self._window(v, g)
        if not frames:
            return
        host = self._host[v]
        if self._buf[slot] is None or self._buf[slot].shape[1:] != host.shape[1:] or self._buf[slot].shape[0] < self.span:
            self._buf[slot] = torch.empty((self.span,) + tuple(host.shape[1:]), dtype=host.dtype, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)            # kernels queued so far may still read this buffer (two groups back)
        with torch.cuda.stream(self.copy_stream):
            # runs of consecutive frames -> one async copy each (a window is one run; group 0 of a short video may add the
            # global frames as a second one)
            j = 0
            a = 0
            while a < len(frames):
                b = a
                while b + 1 < len(frames) and frames[b + 1] == frames[b] + 1:
                    b += 1
                src = host[frames[a]:frames[b] + 1]
                self._buf[slot][j:j + (b - a + 1)].copy_(src, non_blocking=True)
                self.h2d_bytes += src.numel() * src.element_size()
                j += b - a + 1
                a = b + 1
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._ready[slot] = ((v, g), ev, {f: j for j, f in enumerate(frames)})

    def _enter_group(self, v, g):
        slot = next((i for i in (0, 1) if self._ready[i] is not None and self._ready[i][0] == (v, g)), None)
        if slot is None:                             # cold start: nothing ran ahead for this group
            cur = self._slot_of.get("slot")
            slot = 0 if cur is None else cur ^ 1
            self._stage(v, g, slot)
        torch.cuda.current_stream(self.device).wait_event(self._ready[slot][1])
        self._slot_of = {v: (g, slot), "slot": slot}
        n = self.ds.frame_seg_len[self.ds.start_index[v]]
        nxt = None
        if (g + 1) * self.unit < n:
            nxt = (v, g + 1, slot ^ 1)               # runs ahead, under this group's kernels
        elif v + 1 < len(self.ds.start_index) or self.cyclic:
            nxt = ((v + 1) % len(self.ds.start_index), 0, slot ^ 1)          # next video's first group
        if nxt is not None:
            if self._model is not None:
                self._pending = nxt                  # issued by the detector once this group's own uploads and first kernels are queued
            else:
                self._stage(*nxt)

    def attach(self, model):
        """Let the detector trigger the next group's copies right after it has queued its own small uploads (noise draws) and
        its first kernels: a pageable upload issued AFTER a 1-GB prefetch waits for that whole transfer (same DMA queue) while the
        host blocks on it, which serialises copy and compute (measured: 219 ms per video against 179 resident)."""
        self._model = model
        model.after_first_launch = self._kick
        return self

    def _kick(self):
        if self._pending is not None:
            nxt, self._pending = self._pending, None
            self._stage(*nxt)

    def _frame(self, v, f):
        h, w = self.ds.height, self.ds.width
        staged = self._slot_of.get(v)
        j = None if staged is None else self._ready[staged[1]][2].get(f)
        if j is None:
            # a frame outside the staged window (iteration entered mid-group: a sharded rank's leading calls deliver the local
            # frames of the group that starts `max_offset` calls later): its own synchronous copy, counted like the others
            src = self._host[v][f:f + 1]
            self.h2d_bytes += src.numel() * src.element_size()
            return ImageList(src.to(self.device), [torch.Size((h, w))])
        return ImageList(self._buf[staged[1]][j:j + 1], [torch.Size((h, w))])

    def __getitem__(self, idx):
        ds = self.ds
        v, frame_id = ds.video_of[idx], ds.frame_seg_id[idx]
        # stage / enter whenever the requested (video, group) is not the staged one -- not only at a group's first frame, so
        # an index range that resumes mid-group works too.  Exception: the last `max_offset` calls of a group that is not
        # staged (a sharded rank's leading calls, engine.video_shard_plan) only queue frames of the NEXT group; they take the
        # single-frame path of _frame instead of staging a whole window for nothing.
        g = frame_id // self.unit
        if self._slot_of.get(v, (None, None))[0] != g and (frame_id % self.unit == 0 or frame_id % self.unit < self.unit - ds.max_offset):
            self._enter_group(v, g)
        return ds.item(idx, frame=self._frame)       # the index protocol stays the dataset's own; nothing of it is patched
