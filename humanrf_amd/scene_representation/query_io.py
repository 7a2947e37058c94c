"""Query records exchanged with the scene representation. Field names and meanings are those of the reference's
humanrf/scene_representation/query_io.py:6-20 (code written against it constructs them by keyword); the records here
additionally check what the kernels rely on and expose the sizes the callers keep asking for."""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Optional

import torch


def _rows(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.shape[0])


@dataclass
class QueryInput:
    """One row per queried sample."""

    is_training: bool                                      # selects the camera embedding (training) or zeros (eval)
    positions: torch.Tensor                                # (N, 3) scene coordinates in [-0.5, 0.5]
    directions: Optional[torch.Tensor] = None              # (N, 3) unit view directions (radiance queries only)
    frame_numbers: Optional[torch.Tensor] = None           # (N, 1) or (N,) int32 absolute frame numbers
    unique_frame_numbers: Optional[torch.Tensor] = None    # (K, 1) frames present in the batch (unused by the kernels)
    camera_numbers: Optional[torch.Tensor] = None          # (N, 1) or (N,) int32 camera ids (embedding lookup)

    def __post_init__(self) -> None:
        n = _rows(self.positions)
        for f in fields(self):
            if f.name in ("is_training", "positions", "unique_frame_numbers"):
                continue
            r = _rows(getattr(self, f.name))
            if r is not None and r != n:
                raise RuntimeError(f"QueryInput.{f.name} has {r} rows, positions has {n}")

    @property
    def num_queries(self) -> int:
        return _rows(self.positions)


@dataclass
class QueryOutput:
    """Field values at the queried samples; radiance / geometry features are absent from density-only queries."""

    density: torch.Tensor                                  # (N, 1) fp32, sigma = exp(h0) * density_scale
    geometry_features: Optional[torch.Tensor] = None       # (N, 15) fp16, the other outputs of the density network
    radiance: Optional[torch.Tensor] = None                # (N, 3) RGB in [0, 1]

    @property
    def num_queries(self) -> int:
        return _rows(self.density)
