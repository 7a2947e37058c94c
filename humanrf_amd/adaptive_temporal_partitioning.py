"""Adaptive temporal partitioning with the semantics of humanrf/adaptive_temporal_partitioning.py:8-107:
greedy clustering of consecutive frames by the expansion factor of the union of their occupancy grids,
segment sizes drawn from {6, 12, 25, 50, 100}. The reference loads every grid through NumPy on the CPU; here
`get_grid(frame)` may return a device tensor and the 256^3 union / popcount run wherever that tensor lives.
One-off preprocessing: it sets the per-segment hash-table sizes (humanrf.py:106-109), nothing in the step."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch

PREDEFINED_SEGMENT_SIZES = [6, 12, 25, 50, 100]


def get_segment_size(num_frames: int) -> int:
    for idx, segment_size in enumerate(PREDEFINED_SEGMENT_SIZES[:-1]):
        if num_frames < PREDEFINED_SEGMENT_SIZES[idx + 1]:
            return segment_size
    return PREDEFINED_SEGMENT_SIZES[-1]


def get_final_segment_size(num_frames_left: int) -> int:
    for segment_size in PREDEFINED_SEGMENT_SIZES:
        if num_frames_left <= segment_size:
            return segment_size
    return PREDEFINED_SEGMENT_SIZES[-1]


def compute_adaptive_segment_sizes(get_grid: Callable[[int], torch.Tensor], sorted_frame_numbers: Sequence[int],
                                   expansion_factor_threshold: float = 1.25) -> List[int]:
    min_size, max_size = min(PREDEFINED_SEGMENT_SIZES), max(PREDEFINED_SEGMENT_SIZES)
    cluster, cluster_frames, initial = None, 0, 0
    sizes: List[int] = []
    idx, total, decided = 0, len(sorted_frame_numbers), 0
    while idx < total:
        grid = torch.as_tensor(get_grid(sorted_frame_numbers[idx])) == 255
        if cluster_frames == 0:
            initial = int(grid.sum())
            cluster = grid.clone()
        else:
            cluster |= grid  # Equation (2)
        cluster_frames += 1
        if cluster_frames >= min_size:
            expansion = int(cluster.sum()) / max(initial, 1)  # Equation (4)
            if expansion > expansion_factor_threshold or cluster_frames >= max_size:
                size = get_segment_size(cluster_frames)
                decided += size
                cluster, cluster_frames = None, 0
                idx = decided
                sizes.append(size)
                continue
        idx += 1
    if decided < total:
        sizes.append(get_final_segment_size(total - decided))
    assert sum(sizes) >= total
    return sizes
