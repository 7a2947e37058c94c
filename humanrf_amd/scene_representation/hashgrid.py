"""Host-side geometry of the multi-resolution hash grids (tcnn HashGrid) and of the
temporal segments (humanrf/scene_representation/humanrf.py:79-120). Pure NumPy; produces the
hrf_segment_meta array the kernels index."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .._lib import HRF_MAX_LEVELS, LevelMeta, SegmentMeta

PREDEFINED_SEGMENT_SIZES = [6, 12, 25, 50, 100]  # humanrf/adaptive_temporal_partitioning.py:8


def per_level_scale(base_resolution: int, finest_resolution: int, n_levels: int) -> float:
    # decomposition4d.py:73
    return float(np.exp(np.log(finest_resolution / base_resolution) / (n_levels - 1)))


def level_table(n_levels: int, log2_hashmap_size: int, base_resolution: int, pls: float) -> List[Tuple[float, int, int, int, int]]:
    """[(scale, res, size, offset, hashed)] per level; float steps in fp32 like tcnn's grid_scale()."""
    f32 = np.float32
    log2_pls = np.log2(f32(pls)).astype(f32)
    out, offset = [], 0
    for l in range(n_levels):
        scale = f32(np.exp2(f32(f32(l) * log2_pls)).astype(f32) * f32(base_resolution) - f32(1.0))
        res = int(np.ceil(scale)) + 1
        n = min(res ** 3, 2 ** 31 - 1)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        hashed = res ** 3 > n
        if hashed and (n & (n - 1)) != 0:
            raise ValueError("hashed level whose size is not a power of two")
        out.append((float(scale), res, n, offset, int(hashed)))
        offset += n
    return out


def segment_log2_hashmap_size(segment_size: int, log2_hashmap_size: int) -> int:
    # humanrf.py:107-109
    return int(np.round(np.log2(segment_size / max(PREDEFINED_SEGMENT_SIZES) * (2 ** log2_hashmap_size))))


def build_segment_meta(segment_sizes: Sequence[int], n_levels: int, log2_hashmap_size: int, base_resolution: int,
                       finest_resolution: int):
    """-> (ctypes array of SegmentMeta, per-segment entries per encoding, total entries over all segments
    and encodings)."""
    if n_levels > HRF_MAX_LEVELS:
        raise ValueError(f"n_levels > {HRF_MAX_LEVELS} is not supported")
    pls = per_level_scale(base_resolution, finest_resolution, n_levels)
    metas = (SegmentMeta * len(segment_sizes))()
    entries_per_seg, total = [], 0
    for s, size in enumerate(segment_sizes):
        lv = level_table(n_levels, segment_log2_hashmap_size(size, log2_hashmap_size), base_resolution, pls)
        entries = lv[-1][3] + lv[-1][2]
        metas[s].table_offset = total
        metas[s].entries = entries
        metas[s].n_levels = n_levels
        for l, (scale, res, n, off, hashed) in enumerate(lv):
            metas[s].levels[l] = LevelMeta(scale, res, n, off, hashed)
        entries_per_seg.append(entries)
        total += 4 * entries
    return metas, entries_per_seg, total


def frame_tables(sorted_frame_numbers: Sequence[int], segment_sizes: Sequence[int]):
    """frame number -> segment number / normalized local frame number lookup (humanrf.py:79-98)."""
    num_frames = len(sorted_frame_numbers)
    end = np.cumsum(segment_sizes, dtype=np.int32)
    end[-1] = min(end[-1], num_frames)
    start = np.concatenate((np.zeros(1, dtype=np.int32), end[:-1]))
    f2s = np.full((sorted_frame_numbers[-1] + 1), fill_value=-1, dtype=np.int32)
    f2l = np.full((sorted_frame_numbers[-1] + 1), fill_value=-1, dtype=np.float32)
    for s in range(len(segment_sizes)):
        frames = [sorted_frame_numbers[j] for j in range(start[s], end[s])]
        for local, fn in enumerate(frames):
            f2s[fn] = s
            f2l[fn] = local / len(frames)
    return f2s, f2l
