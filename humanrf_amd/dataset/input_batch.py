"""InputBatch: the per-ray / per-sample tensors handed from the sampler to the renderer.
Same fields, shapes and dtypes as actorshq/dataset/input_batch.py:8-50."""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class InputBatch:
    ray_origins: torch.Tensor = None           # (#rays, 3) float
    ray_directions: torch.Tensor = None        # (#rays, 3) float
    minmaxes: torch.Tensor = None              # (#rays, 2) float
    rgba: torch.Tensor = None                  # (#rays, 4) float
    ray_masks: torch.Tensor = None             # (>= #rays, 1) bool
    frame_numbers: torch.Tensor = None         # (#rays, 1) int32
    unique_frame_numbers: torch.Tensor = None  # (K, 1) int32
    camera_numbers: torch.Tensor = None        # (#rays, 1) int32
    sample_distances: torch.Tensor = None      # (#samples, 1) float
    ray_indices: torch.Tensor = None           # (#samples,) int64, sorted, one contiguous run per ray
    width: int = None
    height: int = None

    @property
    def num_rays(self) -> int:
        return self.ray_origins.shape[0]

    @property
    def num_samples(self) -> int:
        return self.sample_distances.shape[0]
