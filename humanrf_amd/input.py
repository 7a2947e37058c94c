"""merge_input_batches with the semantics of humanrf/input.py:10-55: concatenate variable-size batches,
re-base ray_indices, cut at whole rays below max_num_samples, recompute the unique frame numbers.
Private per-sample caches the pruning pass attaches to a batch (see volume_rendering.prune_samples) are merged
and cut exactly like the reference's per-sample tensors."""
from __future__ import annotations

from typing import List, Optional

import torch

from .dataset.input_batch import InputBatch


def merge_input_batches(input_batches: List[InputBatch], max_num_samples: Optional[int] = None) -> InputBatch:
    final = InputBatch()
    for key, val in vars(input_batches[0]).items():
        if key != "ray_indices":
            if val is None:
                setval = None
            elif isinstance(val, torch.Tensor):
                setval = torch.cat([getattr(b, key) for b in input_batches], dim=0)
            elif isinstance(val, int):
                setval = getattr(input_batches[0], key)
            else:
                raise RuntimeError("Unknown data type in the input_batches!")
            setattr(final, key, setval)

    if input_batches[0].ray_indices is not None:
        parts = [input_batches[0].ray_indices]
        acc = 0
        for i in range(1, len(input_batches)):
            acc += input_batches[i - 1].num_rays
            parts.append(input_batches[i].ray_indices + acc)
        final.ray_indices = torch.cat(parts, dim=0)

    if max_num_samples is not None:
        num_rays = final.num_rays
        num_samples = final.num_samples
        if num_samples > max_num_samples:
            cutoff = final.ray_indices[max_num_samples]
            for key, val in list(vars(final).items()):
                if isinstance(val, torch.Tensor):
                    setval = val
                    if key == "ray_masks":
                        setval = val[val.cumsum(0) < cutoff]
                    elif val.shape[0] == num_rays:
                        setval = val[:cutoff]
                    elif val.shape[0] == num_samples:
                        setval = val[final.ray_indices < cutoff]
                    setattr(final, key, setval)

    if final.frame_numbers is not None:
        final.unique_frame_numbers = torch.unique(final.frame_numbers, sorted=False, return_inverse=False).view(-1, 1)
    return final
