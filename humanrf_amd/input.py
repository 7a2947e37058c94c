"""merge_input_batches with the semantics of humanrf/input.py:10-55: concatenate variable-size batches, re-base
ray_indices, cut at whole rays below max_num_samples, recompute the unique frame numbers.
The fields of an InputBatch play one of four roles -- per drawn ray (ray_masks), per surviving ray, per sample, scalar --
and the merge treats each role in one place."""
from __future__ import annotations

from typing import List, Optional

import torch

from .dataset.input_batch import PER_RAY_FIELDS, InputBatch

_PER_RAY = PER_RAY_FIELDS
_PER_SAMPLE = ("sample_distances",)   # ray_indices are per sample too, but need re-basing: handled separately
_SCALAR = ("width", "height")


def _cat(batches: List[InputBatch], name: str) -> Optional[torch.Tensor]:
    first = getattr(batches[0], name)
    if first is None:
        return None
    if not isinstance(first, torch.Tensor):
        raise RuntimeError("Unknown data type in the input_batches!")
    return torch.cat([getattr(b, name) for b in batches], dim=0)


def merge_input_batches(input_batches: List[InputBatch], max_num_samples: Optional[int] = None) -> InputBatch:
    merged = InputBatch()
    for name in _PER_RAY + _PER_SAMPLE + ("ray_masks", "unique_frame_numbers"):
        setattr(merged, name, _cat(input_batches, name))
    for name in _SCALAR:
        value = getattr(input_batches[0], name)
        if value is not None and not isinstance(value, int):
            raise RuntimeError("Unknown data type in the input_batches!")
        setattr(merged, name, value)

    # ray indices of batch k are shifted by the number of rays of batches 0..k-1
    if input_batches[0].ray_indices is not None:
        shifted, base = [], 0
        for b in input_batches:
            shifted.append(b.ray_indices + base if base else b.ray_indices)
            base += b.num_rays
        merged.ray_indices = torch.cat(shifted, dim=0)

    # whole rays only: the ray that owns sample number max_num_samples and every ray after it are dropped
    if max_num_samples is not None and merged.num_samples > max_num_samples:
        first_dropped = merged.ray_indices[max_num_samples]
        keep_samples = merged.ray_indices < first_dropped
        for name in _PER_RAY:
            t = getattr(merged, name)
            if t is not None:
                setattr(merged, name, t[:first_dropped])
        for name in _PER_SAMPLE:
            t = getattr(merged, name)
            if t is not None:
                setattr(merged, name, t[keep_samples])
        if merged.ray_masks is not None:  # entries of the drawn rays that belong to the kept surviving rays
            merged.ray_masks = merged.ray_masks[merged.ray_masks.cumsum(0) < first_dropped]
        merged.ray_indices = merged.ray_indices[keep_samples]

    if merged.frame_numbers is not None:
        merged.unique_frame_numbers = torch.unique(merged.frame_numbers, sorted=False, return_inverse=False).view(-1, 1)
    return merged
