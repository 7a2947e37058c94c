"""Drop-in for actorshq.dataset.occupancy_grid_native (occupancy_grid.cu:8-95): a ring of (G,G,G) uint8
volumes [z][y][x] kept as plain HBM buffers (no texture hardware); `add_grid` returns the int64 handle the
sampler consumes. The trilinear / clamp / normalised-coordinate fetch of the reference's texture descriptor is
done in software inside the sampler kernels."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


class OccupanyGrid:  # sic: the reference spells it this way (occupancy_grid.cu:8)
    def __init__(self, grid_resolution: int, buffer_size: int):
        self._handle = ctypes.c_void_p()
        self.grid_resolution = int(grid_resolution)
        self.buffer_size = int(buffer_size)
        check(_lib.lib().hrf_occgrid_create(self.grid_resolution, self.buffer_size, ctypes.byref(self._handle)))

    def add_grid(self, grid: torch.Tensor) -> int:
        # CHECK_CONTIGUITY_AND_DEVICE(grid, torch::kCUDA)  (occupancy_grid.cu:59)
        if not grid.is_contiguous():
            raise RuntimeError("Tensor not contiguous: grid")
        if not grid.is_cuda:
            raise RuntimeError("Tensor is not on the expected device: grid")
        if grid.dtype != torch.uint8 or grid.dim() != 3:
            raise RuntimeError("grid must be a (G,G,G) uint8 tensor")
        out = ctypes.c_int64()
        check(_lib.lib().hrf_occgrid_add(self._handle, ptr(grid), grid.shape[0], grid.shape[1], grid.shape[2],
                                         stream_ptr(), ctypes.byref(out)))
        return out.value

    def __del__(self):
        try:
            if self._handle:
                _lib.lib().hrf_occgrid_destroy(self._handle)
                self._handle = ctypes.c_void_p()
        except Exception:
            pass
