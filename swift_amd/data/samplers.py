"""Samplers of the training data pipeline (mirror reference src/swift/data/samplers.py).

``InfiniteSampler``: endless, seeded, rank-strided index stream with a sliding-window reshuffle (samplers.py:9-54);
``DeltaBatchSampler``: gives every sample of a batch the same forecast interval (:59-85); ``AttributeSubset``:
``Subset`` that forwards attribute access to the wrapped dataset (:90-98).
"""
from __future__ import annotations

import numpy as np
from torch.utils.data import BatchSampler, Sampler, Subset


class InfiniteSampler(Sampler):
    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size
        self.offset = 1

    def set_offset(self, offset: int):
        """number of forecast steps each sample must leave room for"""
        assert isinstance(offset, int) and offset > 0, "offset must be positive"
        self.offset = offset

    def __iter__(self):
        n = len(self.dataset)
        order = np.arange(n)
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.default_rng(self.seed + self.offset - 1)
            rnd.shuffle(order)
            window = int(np.rint(n * self.window_size))
        k = 0
        while True:
            i = k % n
            if k % self.num_replicas == self.rank and order[i] + self.offset - 1 < n:
                yield (order[i], self.offset) if self.offset > 1 else order[i]
            if window >= 2:
                j = (i - rnd.integers(window)) % n
                order[i], order[j] = order[j], order[i]
            k += 1


class DeltaBatchSampler(BatchSampler):
    def __init__(self, sampler: InfiniteSampler, batch_size: int, intervals, seed: int = 0, drop_last: bool = False):
        super().__init__(sampler=sampler, batch_size=batch_size, drop_last=drop_last)
        self.intervals = list(intervals)
        self.rng = np.random.default_rng(seed)

    def __iter__(self):
        for batch in super().__iter__():
            delta = int(self.rng.choice(self.intervals))
            yield [(e[0], e[1], delta) if isinstance(e, tuple) else (e, self.sampler.offset, delta) for e in batch]


class AttributeSubset(Subset):
    def __init__(self, dataset, indices):
        super().__init__(dataset, indices)
        self.dataset = dataset

    def __getattr__(self, attr):
        return getattr(self.dataset, attr)
