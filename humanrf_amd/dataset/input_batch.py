"""InputBatch: what the sampler hands to the pruning pass and the renderer. The attribute names are the reference's
(actorshq/dataset/input_batch.py:8-50) because prune_samples / render / merge_input_batches and the reference's trainer
address them by name; the record itself is organised by the role a field plays:

  per drawn ray      ray_masks (>= R rows: which of the drawn rays survived the sampler)
  per surviving ray  ray_origins, ray_directions (R,3) f32 | minmaxes (R,2) f32 | rgba (R,4) f32 in [0,1]
                     frame_numbers, camera_numbers (R,1) i32
  per sample         sample_distances (N,1) f32 | ray_indices (N,) i64, sorted: one contiguous run per ray
  per batch          unique_frame_numbers (K,1) i32 | width, height (image size the pixel ids refer to)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Optional, Tuple

import torch

PER_RAY_FIELDS: Tuple[str, ...] = ("ray_origins", "ray_directions", "minmaxes", "rgba", "frame_numbers", "camera_numbers")
PER_SAMPLE_FIELDS: Tuple[str, ...] = ("sample_distances", "ray_indices")


@dataclass
class InputBatch:
    ray_origins: Optional[torch.Tensor] = None
    ray_directions: Optional[torch.Tensor] = None
    minmaxes: Optional[torch.Tensor] = None
    rgba: Optional[torch.Tensor] = None
    ray_masks: Optional[torch.Tensor] = None
    frame_numbers: Optional[torch.Tensor] = None
    unique_frame_numbers: Optional[torch.Tensor] = None
    camera_numbers: Optional[torch.Tensor] = None
    sample_distances: Optional[torch.Tensor] = None
    ray_indices: Optional[torch.Tensor] = None
    width: Optional[int] = None
    height: Optional[int] = None

    @property
    def num_rays(self) -> int:
        """R: rays that survived the sampler (rows of every per-ray field)."""
        return int(self.ray_origins.shape[0])

    @property
    def num_samples(self) -> int:
        """N: samples currently attached to the rays (shrinks when prune_samples runs)."""
        return int(self.sample_distances.shape[0])

    def tensors(self) -> Iterator[Tuple[str, torch.Tensor]]:
        """(name, tensor) of every tensor field that is set."""
        for name, value in vars(self).items():
            if isinstance(value, torch.Tensor):
                yield name, value

    def check(self) -> None:
        """Row counts of the per-ray / per-sample fields and the ordering the wavefront-per-ray kernels rely on."""
        for name in PER_RAY_FIELDS:
            t = getattr(self, name)
            if t is not None and t.shape[0] != self.num_rays:
                raise RuntimeError(f"InputBatch.{name} has {t.shape[0]} rows for {self.num_rays} rays")
        if self.ray_indices is not None:
            if self.ray_indices.shape[0] != self.num_samples:
                raise RuntimeError("InputBatch.ray_indices and sample_distances disagree on the number of samples")
            if self.num_samples > 1 and bool((self.ray_indices[1:] < self.ray_indices[:-1]).any()):
                raise RuntimeError("InputBatch.ray_indices must be sorted (one contiguous run of samples per ray)")
