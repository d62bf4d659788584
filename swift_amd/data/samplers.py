"""Index streams of the training data pipeline.

The index stream of a run is part of its reproducibility contract (a resumed or re-sharded run must see the samples the
reference would have seen), so ``InfiniteSampler`` emits exactly the stream of reference ``src/swift/data/samplers.py:9-54``
-- EDM's endless sampler: a seeded permutation that keeps reshuffling itself through random swaps inside a trailing window
-- but is organised differently: the stream is produced one *lap* (one pass over the ``n`` positions) at a time.  A lap draws
all of its ``n`` swap distances with a single vectorised ``Generator.integers`` call (numpy's bounded-integer stream is the
same drawn one at a time or in bulk; ``tests/test_host_logic.py`` pins that and the stream itself against a fixture made by
the reference's sampler), and the rank stride is carried across laps as a phase instead of a global counter.

``DeltaBatchSampler`` (reference :59-85) stamps one forecast interval on every sample of a batch; ``AttributeSubset``
(:90-98) is a ``Subset`` that answers attribute look-ups from the wrapped dataset.
"""
from __future__ import annotations

from typing import Iterator, List, Sequence, Tuple, Union

import numpy as np
from torch.utils.data import BatchSampler, Sampler, Subset

Item = Union[int, Tuple[int, int]]


class _Lap:
    """One pass over the positions of the permutation: emits the entries this rank owns, then applies the lap's swaps."""

    __slots__ = ("order", "dist")

    def __init__(self, order: np.ndarray, dist):
        self.order, self.dist = order, dist

    def run(self, phase: int, stride: int, last_valid: int) -> Iterator[int]:
        """``phase``: position of this lap's first entry inside the rank cycle.  Yields the owned entries that leave room
        for the forecast offset (value <= ``last_valid``); position i is swapped with a position at most
        ``window - 1`` behind it (cyclically) right after it has been visited."""
        order, dist, n = self.order, self.dist, self.order.size
        for i in range(n):
            v = order[i]
            if (phase + i) % stride == 0 and v <= last_valid:
                yield v
            if dist is not None:
                j = (i - int(dist[i])) % n
                order[i], order[j] = order[j], v


class InfiniteSampler(Sampler):
    """Endless, seeded, rank-strided stream of dataset indices; ``(index, offset)`` pairs once ``set_offset(k > 1)`` asked for
    k-step samples.  Same constructor as the reference's."""

    def __init__(self, dataset, rank: int = 0, num_replicas: int = 1, shuffle: bool = True, seed: int = 0,
                 window_size: float = 0.5):
        if len(dataset) <= 0:
            raise AssertionError("empty dataset")
        if not (num_replicas > 0 and 0 <= rank < num_replicas):
            raise AssertionError(f"rank {rank} outside 0..{num_replicas - 1}")
        if not 0 <= window_size <= 1:
            raise AssertionError("window_size is a fraction of the dataset")
        super().__init__()
        self.dataset = dataset
        self.rank, self.num_replicas = rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size
        self.offset = 1

    def set_offset(self, offset: int) -> None:
        """Number of consecutive steps a sample spans (multistep finetuning): indices too close to the end are skipped."""
        assert isinstance(offset, int) and offset > 0, "offset must be positive"
        self.offset = offset

    def __iter__(self) -> Iterator[Item]:
        n = len(self.dataset)
        span = self.offset  # fixed for the life of this iterator: the trainer builds a new one after every set_offset (trainer.py:356-376)
        order = np.arange(n)
        rng, window = None, 0
        if self.shuffle:
            rng = np.random.default_rng(self.seed + span - 1)  # a different permutation per rollout length
            rng.shuffle(order)
            window = int(np.rint(n * self.window_size))
        swapping = window >= 2
        # global step k = lap * n + i is owned by this rank when k % num_replicas == rank
        phase = (-self.rank) % self.num_replicas
        while True:
            lap = _Lap(order, rng.integers(window, size=n) if swapping else None)
            for v in lap.run(phase, self.num_replicas, n - span):
                yield (v, span) if span > 1 else v
            phase = (phase + n) % self.num_replicas


class DeltaBatchSampler(BatchSampler):
    """Batches of ``(index, offset, delta)``: every sample of a batch shares one forecast interval ``delta`` drawn from
    ``intervals`` with this sampler's own seeded generator."""

    def __init__(self, sampler: InfiniteSampler, batch_size: int, intervals: Sequence[int], seed: int = 0,
                 drop_last: bool = False):
        super().__init__(sampler=sampler, batch_size=batch_size, drop_last=drop_last)
        self.intervals = list(intervals)
        self.rng = np.random.default_rng(seed)

    def _stamp(self, item: Item, delta: int) -> Tuple[int, int, int]:
        index, offset = item if isinstance(item, tuple) else (item, self.sampler.offset)
        return index, offset, delta

    def __iter__(self) -> Iterator[List[Tuple[int, int, int]]]:
        for batch in super().__iter__():
            delta = int(self.rng.choice(self.intervals))
            yield [self._stamp(item, delta) for item in batch]


class AttributeSubset(Subset):
    """``Subset`` whose unknown attributes (statistics, variable names, ``intervals`` ...) resolve on the full dataset."""

    def __init__(self, dataset, indices):
        super().__init__(dataset, indices)
        self.dataset = dataset

    def __getattr__(self, name):
        return getattr(self.dataset, name)
