"""`actorshq.toolbox.occupancy_grid_generation_native` on gfx950 (occupancy_grid_generation.cu:82-125): same function
name, argument order and error behaviour; the kernel is hrf_occgrid_from_masks (include/hrf.h)."""
from __future__ import annotations

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..ops import _chk


def generate_from_masks(masks: torch.Tensor, projection_matrices: torch.Tensor, landscape_modes: torch.Tensor,
                        camera_coverage_threshold: int, grid_resolution: int, width: int, height: int) -> torch.Tensor:
    """masks (C, width*height) uint8, projection_matrices (C,4,4) float32 (world->pixel, transposed by the caller so
    that memory is column-major, generate_occupancy_grids_from_masks.py:54-61), landscape_modes (C) bool
    -> (G,G,G) uint8 [z][y][x], 255 where at least `camera_coverage_threshold` cameras see foreground."""
    if masks.size(1) != width * height:
        raise RuntimeError("The number mask entries per camera has to be equal to width*height!")
    _chk(masks, "masks", torch.uint8)
    _chk(projection_matrices, "projection_matrices", torch.float32)
    _chk(landscape_modes, "landscape_modes")
    if landscape_modes.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError("landscape_modes must be bool")
    C = projection_matrices.size(0)
    G = int(grid_resolution)
    grid = torch.empty(G, G, G, dtype=torch.uint8, device=masks.device)
    check(_lib.lib().hrf_occgrid_from_masks(ptr(masks), ptr(projection_matrices), ptr(landscape_modes.view(torch.uint8)),
                                            int(camera_coverage_threshold), C, G, int(width), int(height), ptr(grid),
                                            stream_ptr()))
    return grid


def dilate_masks(masks: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """cv2.dilate(mask, ones((k,k)), iterations=1) for a (N,H,W) uint8 stack on the device
    (generate_occupancy_grids_from_masks.py:64-77 does this per image on the CPU)."""
    _chk(masks, "masks", torch.uint8)
    n, h, w = masks.shape
    out = torch.empty_like(masks)
    check(_lib.lib().hrf_mask_dilate(ptr(masks), w, h, int(kernel_size), n, ptr(out), stream_ptr()))
    return out
