"""Operator boundary of the network family (mirrors reference src/swift/models/abstract.py:12-62)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Sequence, Union

import torch

_Shape2D = Union[int, Sequence[int]]


class AbstractNetwork(torch.nn.Module, ABC):
    """All denoisers implement ``forward(x, t, auxiliary=None, *args, **kwargs)`` (abstract.py:26-35)."""

    def __init__(self, img_resolution: _Shape2D, in_channels: int, out_channels: int):
        super().__init__()
        self.img_resolution = img_resolution
        self.in_channels = in_channels
        self.out_channels = out_channels

    @abstractmethod
    def forward(self, x: torch.Tensor, t: torch.Tensor, auxiliary: Optional[torch.Tensor] = None, *args, **kwargs):
        raise NotImplementedError("subclass must implement this.")


class Shape2D:
    """int | list | tuple -> (h, w); anything else is a TypeError, as in abstract.py:43-62.

    Config containers must therefore be converted to plain lists before they
    reach a constructor (the reference instantiates with ``_convert_="object"``).
    """

    def __init__(self, shape: _Shape2D):
        if isinstance(shape, int):
            self.shape = (shape, shape)
        elif isinstance(shape, (list, tuple)):
            self.shape = tuple(shape)
        else:
            raise TypeError(f"Invalid type {type(shape)}")
        assert len(self.shape) == 2 and all(isinstance(v, int) for v in self.shape)
        self.height, self.width = self.shape
