"""Decomposition4D: one temporal segment's feature grid (four 3-D multi-resolution hash grids + four 1-D
dense vector grids), API of humanrf/scene_representation/decomposition4d.py:42-135, computed by the fused
hrf_encode4d_* kernels instead of 4 tcnn.Encoding calls + compose_tensors.

HumanRF does not instantiate this class (it keeps all segments in one flat buffer); it exists for callers that
use the reference's per-segment module directly, and for tests."""
from __future__ import annotations

import torch

from .. import ops
from . import hashgrid


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, xyzt, tables, vectors):
        module._refresh_half()
        seg = torch.zeros(xyzt.shape[0], dtype=torch.int32, device=xyzt.device)
        need = any(ctx.needs_input_grad)
        feats, enc = ops.encode4d_fwd(xyzt, seg, module._tables_h, vectors.detach().unsqueeze(0).contiguous(),
                                      module._seg_meta, 1, save_enc=need)
        ctx.module = module
        ctx.save_for_backward(xyzt, seg, enc, vectors)
        return feats

    @staticmethod
    def backward(ctx, d_feats):
        xyzt, seg, enc, vectors = ctx.saved_tensors
        module = ctx.module
        d_tables = torch.zeros(module.tables.numel(), dtype=torch.float32, device=xyzt.device)
        d_vectors = torch.zeros_like(vectors)
        ops.encode4d_bwd(xyzt, seg, enc, vectors.detach().unsqueeze(0).contiguous(), module._seg_meta, 1,
                         d_feats.contiguous(), 1.0, d_tables, d_vectors)
        return None, None, d_tables.view_as(module.tables), d_vectors


class Decomposition4D(torch.nn.Module):
    def __init__(self, ngp_n_levels: int = 16, ngp_n_features_per_level: int = 2, ngp_log2_hashmap_size: int = 19,
                 ngp_base_resolution: int = 32, ngp_finest_resolution: int = 2048,
                 vectors_finest_resolution: int = 2048, device: str = "cuda", seed: int = 1337):
        super().__init__()
        if ngp_n_levels != 16 or ngp_n_features_per_level != 2:
            raise NotImplementedError("kernels are specialised for 16 levels x 2 features")
        gen = torch.Generator().manual_seed(seed)
        pls = hashgrid.per_level_scale(ngp_base_resolution, ngp_finest_resolution, ngp_n_levels)
        lv = hashgrid.level_table(ngp_n_levels, ngp_log2_hashmap_size, ngp_base_resolution, pls)
        self.entries = lv[-1][3] + lv[-1][2]
        from .._lib import LevelMeta, SegmentMeta
        metas = (SegmentMeta * 1)()
        metas[0].table_offset = 0
        metas[0].entries = self.entries
        metas[0].n_levels = ngp_n_levels
        for l, row in enumerate(lv):
            metas[0].levels[l] = LevelMeta(*row)
        self.register_buffer("_seg_meta", torch.frombuffer(bytearray(bytes(metas)), dtype=torch.uint8).clone(),
                             persistent=False)
        feature_size = ngp_n_levels * ngp_n_features_per_level
        # decomposition4d.py:76-78
        self.vectors = torch.nn.Parameter(torch.randn(4, vectors_finest_resolution, feature_size, generator=gen) * 0.1)
        # (xyz, xyt, yzt, xzt) tables, each (entries, 2): tcnn's flat per-encoding `params` side by side
        self.tables = torch.nn.Parameter((torch.rand(4, self.entries, 2, generator=gen) * 2.0 - 1.0) * 1e-4)
        self.register_buffer("_tables_h", torch.zeros(4 * self.entries * 2 + 2, dtype=torch.float16), persistent=False)
        self._ver = None
        self.to(torch.device(device))

    def _refresh_half(self):
        ver = (self.tables._version, self.tables.data_ptr(), self._tables_h.data_ptr())
        if ver != self._ver:
            with torch.no_grad():
                self._tables_h[:self.tables.numel()].copy_(self.tables.reshape(-1))
            self._ver = (self.tables._version, self.tables.data_ptr(), self._tables_h.data_ptr())

    def forward(self, xyz, times):
        xyzt = torch.cat((xyz, times), axis=-1).float().contiguous()  # decomposition4d.py:125
        return _EncodeFn.apply(self, xyzt, self.tables, self.vectors)
