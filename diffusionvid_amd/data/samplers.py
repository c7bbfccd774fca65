"""Test-time sharding: contiguous index ranges snapped forward to video starts, one range per rank
(mirror of VIDTestDistributedSampler, mega_core/data/samplers/distributed.py:69-115).  Videos are the
unit because the detector keeps per-video state.  Where the reference's `find_zero` falls off the
end of `start_index` it returns None (and a rank then re-runs the whole set); here that case maps
to len(dataset), i.e. an empty tail shard -- results are identical after the rank-0 merge."""
import math


class VIDTestDistributedSampler:
    def __init__(self, dataset, num_replicas, rank):
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / num_replicas))
        self.start = self.find_zero(rank * self.num_samples)
        self.end = self.find_zero((rank + 1) * self.num_samples)

    def find_zero(self, offset):
        n = len(self.dataset)
        if offset >= n:
            return n
        for index in self.dataset.start_index:
            if index >= offset:
                return index
        return n

    def __iter__(self):
        return iter(range(self.start, self.end))

    def __len__(self):
        return self.num_samples


def balanced_video_partition(video_lengths, num_replicas):
    """Videos -> ranks, balanced by frame count: longest-processing-time greedy -- videos in descending
    length (ties: lower index first) each go to the rank with the fewest frames so far (ties: lower rank).  Returns a
    list (per rank) of video indices in ascending order.  The reference cuts the frame range into `world` equal pieces
    and snaps each cut FORWARD to the next video start (samplers/distributed.py:83-95), which leaves ranks up to one
    video apart and the last rank short by design; with per-video state the video is the only valid unit either way."""
    order = sorted(range(len(video_lengths)), key=lambda v: (-int(video_lengths[v]), v))
    load = [0] * num_replicas
    parts = [[] for _ in range(num_replicas)]
    for v in order:
        r = min(range(num_replicas), key=lambda i: (load[i], i))
        parts[r].append(v)
        load[r] += int(video_lengths[v])
    return [sorted(p) for p in parts]


class VIDBalancedTestSampler:
    """Drop-in for VIDTestDistributedSampler with the balanced partition: iterates the dataset indices of this rank's
    videos, each video as its contiguous index range (the detector needs a video's frames in order)."""

    def __init__(self, dataset, num_replicas, rank):
        starts = list(dataset.start_index)
        ends = starts[1:] + [len(dataset)]
        self.lengths = [e - s for s, e in zip(starts, ends)]
        self.videos = balanced_video_partition(self.lengths, num_replicas)[rank]
        self.ranges = [(starts[v], ends[v]) for v in self.videos]
        self.num_samples = sum(e - s for s, e in self.ranges)

    def __iter__(self):
        for s, e in self.ranges:
            yield from range(s, e)

    def __len__(self):
        return self.num_samples


def vid_val_shaped_lengths(num_videos=555, total_frames=176126, seed=2015):
    """A 555-video-shaped synthetic set (BASELINE.json configs[4]): ImageNet-VID val has 555 snippets and 176126 frames
    (mean 317); individual lengths are not in the reference repo, so they are drawn from a seeded
    log-normal (sigma 0.7: a long tail of multi-thousand-frame snippets next to many short ones, like the real list),
    clipped to >= 8 frames and rescaled to the exact total."""
    import numpy as np
    rng = np.random.RandomState(seed)
    raw = np.exp(rng.normal(0.0, 0.7, size=num_videos))
    lens = np.maximum(8, np.round(raw / raw.sum() * total_frames)).astype(np.int64)
    lens[np.argmax(lens)] += total_frames - int(lens.sum())
    assert lens.sum() == total_frames and lens.min() >= 8
    return lens.tolist()
