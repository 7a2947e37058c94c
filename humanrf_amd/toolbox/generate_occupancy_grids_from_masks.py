"""generate_occupancy_grid_from_masks (actorshq/toolbox/generate_occupancy_grids_from_masks.py:17-99) for an in-memory
dataset: the mask stack of one frame is dilated and carved on the device; file IO (VolumetricDataset, npz output) stays
with the caller."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import occupancy_grid_generation_native as native


def projection_matrices_for(cameras: Sequence, device) -> torch.Tensor:
    """world->pixel matrices stacked and transposed exactly as the reference driver does (:54-61)."""
    p = np.stack([cam.projection_matrix_world2pixel() for cam in cameras], axis=0).astype(np.float32)
    return torch.from_numpy(p).permute(0, 2, 1).to(device=device).contiguous()


def generate_occupancy_grid_from_masks(masks: torch.Tensor, cameras: Sequence, grid_resolution: int,
                                       camera_coverage_threshold: int, dilate: bool = True) -> torch.Tensor:
    """masks: (C, H, W) uint8 foreground masks of ONE frame (scaled cameras: the scene lives in [-0.5, 0.5]^3).
    -> (G,G,G) uint8 occupancy grid. Dilation margin as in the reference: max(width, height) // 128 pixels."""
    C, H, W = masks.shape
    width, height = max(cameras[0].width, cameras[0].height), min(cameras[0].width, cameras[0].height)
    if dilate:
        k = max(width, height) // 128
        if k > 0:
            masks = native.dilate_masks(masks.contiguous(), k)
    landscape = torch.tensor([cam.width > cam.height for cam in cameras], device=masks.device, dtype=torch.bool)
    proj = projection_matrices_for(cameras, masks.device)
    return native.generate_from_masks(masks.reshape(C, -1).contiguous(), proj, landscape, camera_coverage_threshold,
                                      grid_resolution, width, height)
