"""Synthetic video dataset that reproduces the reference's test-time item protocol.

Restates the index arithmetic of VIDMEGADataset._get_test (mega_core/data/datasets/vid_mega.py:164-250)
and __init__ (:10-33) for the DiffusionVID config (SHUFFLED_CUR_TEST False): every dataset index is
one frame of one video; an item carries the current frame, the local reference frame(s) `ref_l`,
the global reference frames `ref_g` (only on frame 0 when STOP_UPDATE_AFTER_INIT_TEST) and the
bookkeeping ints the detector reads.  Frames are seeded random tensors already in
the post-ToTensor [0,1] domain and zero-padded to DATALOADER.SIZE_DIVISIBILITY like
BatchCollator/to_image_list do (collate_batch.py:24-37).
"""
import numpy as np
import torch

from ..structures.image_list import ImageList
from ..utils import synthetic


class SyntheticVIDDataset:
    def __init__(self, video_lengths, cfg, height=600, width=1000, device="cpu", shuffle_seed=None, video_base=0, smooth=False,
                 emit_ref_ahead=True):
        mega = cfg.MODEL.VID.MEGA
        self.max_offset = mega.MAX_OFFSET
        self.all_frame_interval = mega.ALL_FRAME_INTERVAL
        self.key_frame_location = mega.KEY_FRAME_LOCATION
        self.global_enable = mega.GLOBAL.ENABLE
        self.global_size = mega.GLOBAL.SIZE
        self.stop_update_after_init_g_test = mega.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST
        self.infer_batch = cfg.INPUT.INFER_BATCH
        self.lookahead = max(1, int(getattr(cfg.INPUT, "LOOKAHEAD_BATCHES", 1)))
        self.size_divisible = cfg.DATALOADER.SIZE_DIVISIBILITY
        self.height, self.width, self.device = height, width, torch.device(device)
        self.video_base = video_base
        self.smooth = smooth
        self.emit_ref_ahead = emit_ref_ahead      # False: the reference's unchanged item dict (the engine builds the look-ahead)
        self.frame_seg_len, self.frame_seg_id, self.video_of = [], [], []
        self.start_index, self.start_id, self.shuffled_index = [], [], {}
        rng = np.random.RandomState(shuffle_seed) if shuffle_seed is not None else None
        for v, n in enumerate(video_lengths):
            first = len(self.frame_seg_len)
            self.start_index.append(first)
            order = np.arange(n)
            if rng is not None:          # GLOBAL.SHUFFLE (vid_mega.py:26-29); identity when unseeded here
                rng.shuffle(order)
            self.shuffled_index[first] = order
            for f in range(n):
                self.frame_seg_len.append(n)
                self.frame_seg_id.append(f)
                self.video_of.append(v)
                self.start_id.append(first)
        self._cache = {}

    def __len__(self):
        return len(self.frame_seg_len)

    # ---- pure index protocol (vid_mega.py:178-221) -------------------------------------------
    def ref_ids(self, idx):
        frame_id, seg_len = self.frame_seg_id[idx], self.frame_seg_len[idx]
        ref_id_final = min(frame_id + self.max_offset, seg_len - 1)
        if frame_id == 0:
            ref_id_start = max(ref_id_final - self.all_frame_interval + 1, 0)
        else:
            num_ref = min(1, self.all_frame_interval)          # frame_diff == 1 for consecutive frames
            ref_id_start = max(ref_id_final - num_ref + 1, 0)
        ref_l = list(range(ref_id_start, ref_id_final + 1))
        ref_g = []
        if self.global_enable:
            size = self.global_size if frame_id == 0 else (0 if self.stop_update_after_init_g_test else 1)
            order = self.shuffled_index[self.start_id[idx]]
            for i in range(size):
                ref_g.append(int(order[(idx - self.start_id[idx] + self.global_size - i - 1) % seg_len]))
        return ref_l, ref_g, ref_id_final

    # ---- frames ---------------------------------------------------------------------------------
    def frame(self, video, f):
        key = (video, f)
        if key not in self._cache:
            img = synthetic.synthetic_frame(f, self.height, self.width, video=self.video_base + video, smooth=self.smooth)
            d = self.size_divisible
            ph = -(-self.height // d) * d if d else self.height
            pw = -(-self.width // d) * d if d else self.width
            padded = torch.zeros((1, 3, ph, pw), dtype=torch.float32)
            padded[0, :, :self.height, :self.width] = img
            self._cache[key] = ImageList(padded.to(self.device), [torch.Size((self.height, self.width))])
        return self._cache[key]

    def preload(self):
        for idx in range(len(self)):
            self.frame(self.video_of[idx], self.frame_seg_id[idx])

    def __getitem__(self, idx):
        return self.item(idx)

    def item(self, idx, frame=None):
        """The item of dataset index `idx`; `frame(video, f)` replaces the dataset's own frame source (data/prefetch.py hands
        out views of its device staging buffers through it) -- the index protocol stays this class's."""
        frame = frame or self.frame
        v, frame_id, seg_len = self.video_of[idx], self.frame_seg_id[idx], self.frame_seg_len[idx]
        ref_l, ref_g, ref_id_final = self.ref_ids(idx)
        images = {
            "cur": frame(v, frame_id),
            "ref_l": [frame(v, i) for i in ref_l],
            "ref_g": [frame(v, i) for i in ref_g],
            "frame_category": 0 if frame_id == 0 else 1,
            "frame_id": frame_id,
            "start_id": 0,
            "end_id": seg_len - 1,
            "seg_len": seg_len,
            "last_queue_id": ref_id_final,
        }
        if self.emit_ref_ahead and self.lookahead > 1 and frame_id % (self.infer_batch * self.lookahead) == 0:
            # INPUT.LOOKAHEAD_BATCHES extension: the 8 frame slots each of the next batches will be built from, i.e.
            # exactly what calls fb-7 .. fb will deliver through `ref_l` (last frame repeated past the end of the video)
            images["ref_ahead"] = {
                fb: [frame(v, min(fb - self.infer_batch + 1 + i + self.max_offset, seg_len - 1)) for i in range(self.infer_batch)]
                for fb in range(frame_id + self.infer_batch, min(frame_id + self.infer_batch * self.lookahead, seg_len), self.infer_batch)}
        return images, None, [idx + i for i in range(self.infer_batch)]


class PooledVIDDataset(SyntheticVIDDataset):
    """A VID-val-shaped synthetic set (hundreds of videos, 10^5 frames) does not fit in HBM as distinct fp32 frames (176126 x
    7.5 MB = 1.3 TB): frame f of video v is drawn from a pool of `pool` distinct seeded frames.  Protocol, video lengths
    and per-video state are those of the full set; only the pixel content repeats."""

    def __init__(self, video_lengths, cfg, pool=128, **kw):
        super().__init__(video_lengths, cfg, **kw)
        self.pool = int(pool)

    def frame(self, video, f):
        return super().frame(0, (video * 31 + f) % self.pool)
