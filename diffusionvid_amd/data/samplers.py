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
