"""HumanRF scene representation on the MI355X kernels.

Mirrors humanrf/scene_representation/humanrf.py:13-220 (constructor arguments, `density`, `forward`,
`get_params`, the two frame lookup buffers) but is organised for the hardware instead of for tcnn:

  * all temporal segments' hash tables live in ONE flat fp32 master parameter (+ one fp16 shadow copy the
    kernels gather from) and stay resident in HBM -- the reference swaps inactive segments to the CPU on
    every call (humanrf.py:169-179); with 288 GB there is nothing to swap. One kernel serves samples of
    mixed segments through a per-sample segment id.
  * encode (4 hash grids + vector compose), sigma_net (+truncated_exp) and color_net are three launches
    forward and two backward, with no per-segment Python loop, no boolean-mask gather/scatter and no
    host synchronisation (the reference syncs at humanrf.py:162).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Tuple

import torch

from .. import ops
from . import hashgrid
from .query_io import QueryInput, QueryOutput


def _xavier_uniform(out_f: int, in_f: int, gen: torch.Generator) -> torch.Tensor:
    # tcnn FullyFusedMLP initialisation
    bound = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen) * 2.0 - 1.0) * bound



def _call_hook(hook) -> None:
    """Hooks a TrainEngine registers with the model (`_tables_ready`, `_master_sync`) are weak references to bound methods: the
    model does not keep the engine (and its process group) alive, and deepcopy / pickle of the model do not drag it along."""
    if hook is None:
        return
    import weakref
    fn = hook() if isinstance(hook, weakref.ReferenceType) else hook
    if fn is not None:
        fn()


class _FieldFn(torch.autograd.Function):
    """(table, vector, MLP, embedding parameters) -> (sigma (N,), radiance (N,3), geometry_features (N,15)).
    Backward = hrf_mlp_bwd + hrf_encode4d_bwd with a fixed internal gradient scale (tcnn uses 128 as well)."""

    @staticmethod
    def forward(ctx, model, xyzt, seg, ray_dirs, sample_ray, ray_cameras, use_emb, table_params, vectors,
                sigma_params, color_params, emb_weight):
        model._refresh_half()
        need_grad = any(ctx.needs_input_grad)  # (grad mode is always off inside Function.forward)
        feats, enc = ops.encode4d_fwd(xyzt, seg, model._tables_h, vectors.detach(), model._seg_meta, model.num_segments,
                                      save_enc=need_grad)
        sw1, sw2 = model._sigma_w()
        cw1, cw2, cw3 = model._color_w()
        h, sigma = ops.density_mlp_fwd(feats, sw1, sw2, float(model.density_scale))
        emb = emb_weight.detach() if emb_weight is not None else None
        rgb_h = ops.color_mlp_fwd(ray_dirs, sample_ray, h, emb, ray_cameras, model.camera_embedding_dim, use_emb,
                                  cw1, cw2, cw3, geo_dim=model.geometry_feature_dim)
        ctx.model = model
        ctx.use_emb = use_emb
        ctx.has_emb = emb_weight is not None
        ctx.save_for_backward(xyzt, seg, enc, feats, ray_dirs, sample_ray, ray_cameras, vectors, emb_weight)
        geo = h[:, 1:1 + model.geometry_feature_dim]
        ctx.mark_non_differentiable(geo)
        return sigma, rgb_h.float(), geo

    @staticmethod
    def backward(ctx, d_sigma, d_rgb, _d_geo):
        model = ctx.model
        xyzt, seg, enc, feats, ray_dirs, sample_ray, ray_cameras, vectors, emb_weight = ctx.saved_tensors
        dev = xyzt.device
        scale = float(model.internal_grad_scale)
        sw1, sw2 = model._sigma_w()
        cw1, cw2, cw3 = model._color_w()
        g_sigma = torch.zeros(model.sigma_params.numel(), dtype=torch.float32, device=dev)
        g_color = torch.zeros(model.color_params.numel(), dtype=torch.float32, device=dev)
        g_emb = torch.zeros_like(emb_weight) if ctx.has_emb else None
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        n1 = 64 * 32
        kin = model.color_in_pad
        d_sigma = (d_sigma.float() * scale).contiguous() if d_sigma is not None else torch.zeros(xyzt.shape[0], device=dev)
        d_rgb = (d_rgb.float() * scale).contiguous() if d_rgb is not None else torch.zeros(xyzt.shape[0], 3, device=dev)
        d_feats = ops.mlp_bwd(feats, ray_dirs, sample_ray, emb_weight.detach() if ctx.has_emb else None, ray_cameras,
                              model.camera_embedding_dim, ctx.use_emb and ctx.has_emb, sw1, sw2, cw1, cw2, cw3,
                              float(model.density_scale), d_rgb, d_sigma,
                              g_sigma[:n1], g_sigma[n1:], *model.split_color(g_color), g_emb, flags, level_major=True, geo_dim=model.geometry_feature_dim)
        d_tables = torch.zeros(model.table_params.numel(), dtype=torch.float32, device=dev)
        d_vectors = torch.zeros_like(vectors)
        ops.encode4d_bwd(xyzt, seg, enc, vectors.detach(), model._seg_meta, model.num_segments, d_feats, scale,
                         d_tables, d_vectors, level_major=True)
        inv = 1.0 / scale
        # an fp16 overflow inside the backward must surface as a non-finite gradient (GradScaler found_inf)
        poison = torch.where(flags[0] != 0, float("inf"), 0.0).to(torch.float32)
        g_sigma = g_sigma * inv + poison
        g_color = g_color * inv
        if g_emb is not None:
            g_emb = g_emb * inv
        return (None, None, None, None, None, None, None, d_tables, d_vectors, g_sigma, g_color, g_emb)


class HumanRF(torch.nn.Module):
    def __init__(
        self,
        density_scale: float,
        sorted_frame_numbers: Tuple[int, ...],
        n_features_per_level: int,
        log2_hashmap_size: int,
        n_levels: int,
        coarsest_resolution: int,
        finest_resolution: int,
        geometry_feature_dim: int,
        n_neurons: int,
        n_hidden_layers_density: int,
        n_hidden_layers_color: int,
        sh_degree: int,
        segment_sizes: Tuple[int, ...],
        camera_embedding_dim: int,
        device: str = "cuda",
        seed: int = 1337,
        mlp_precision: str = "fp16",
        **kwargs,
    ):
        """Same arguments as the reference constructor (humanrf.py:14-31); `device`, `seed` and `mlp_precision` are
        extras. mlp_precision: "fp16" = tcnn's FullyFusedMLP arithmetic (the reference configuration); "bf16" = the two
        MLPs compute in bf16 on the matrix cores (BASELINE.json configs[4]) -- hash tables stay fp16 either way."""
        super().__init__()
        if mlp_precision not in ("fp16", "bf16"):
            raise ValueError("mlp_precision must be 'fp16' or 'bf16'")
        self.mlp_precision = mlp_precision
        # What the gfx950 kernels are specialised for (model_args.py:10-35): two features per level, 64-neuron networks with one
        # (sigma_net) hidden layer, degree-4 spherical harmonics. n_levels and geometry_feature_dim are free
        # within the kernels' fixed row widths (round 5): 2..16 levels in 32-wide feature rows (tcnn pads sigma_net's input with
        # ones to a multiple of 16; columns beyond that hold zeros), 0..15 geometry features in the colour network's
        # [SH 16 | geo | embedding | ones] input of 32 or 48 columns. n_hidden_layers_color (round 6): 1, 2 (the reference's
        # configurations) or 3 -- the colour kernels are instantiated per depth, sigma_net (which the prune march and the
        # render-pass encode kernel carry inside) stays at one hidden layer.
        if (n_features_per_level, n_neurons, n_hidden_layers_density, sh_degree) != (2, 64, 1, 4) \
                or int(n_hidden_layers_color) not in (1, 2, 3):
            raise NotImplementedError(
                "the gfx950 kernels are specialised for n_features_per_level=2, n_neurons=64, n_hidden_layers_density=1, "
                "sh_degree=4 (humanrf/args/model_args.py:10-35); n_levels (2..16), geometry_feature_dim (0..15) and "
                "n_hidden_layers_color (1..3) are free")
        self.n_hidden_layers_color = int(n_hidden_layers_color)
        if not 2 <= int(n_levels) <= 16:
            raise NotImplementedError("n_levels must be in [2, 16] (the per-level scale divides by n_levels - 1, decomposition4d.py:73; "
                                      "the kernels' feature rows hold 16 levels)")
        if not 0 <= int(geometry_feature_dim) <= 15:
            raise NotImplementedError("geometry_feature_dim must be in [0, 15] (sigma_net's 16 padded outputs: density + 15)")
        if not 1 <= int(geometry_feature_dim) + int(camera_embedding_dim) <= 32:
            raise NotImplementedError("16 + geometry_feature_dim + camera_embedding_dim must lie in (16, 48]: the colour network's "
                                      "kernels are built for 32 and 48 input columns")
        self.n_levels, self.geometry_feature_dim = int(n_levels), int(geometry_feature_dim)
        if not 0 <= camera_embedding_dim <= 17:
            raise NotImplementedError("camera_embedding_dim must be in [0, 17]")
        self.density_scale = density_scale
        self.num_frames = len(sorted_frame_numbers)
        self.num_segments = len(segment_sizes)
        self.segment_sizes = tuple(int(s) for s in segment_sizes)
        self.camera_embedding_dim = camera_embedding_dim
        self.total_feature_dim = n_levels * n_features_per_level
        self.vec_res = finest_resolution
        self.color_in_pad = 16 * ((16 + self.geometry_feature_dim + camera_embedding_dim + 15) // 16)   # tcnn's padded input width
        self.sigma_in_pad = 16 * ((self.total_feature_dim + 15) // 16)
        self.internal_grad_scale = 128.0
        dev = torch.device(device)
        gen = torch.Generator().manual_seed(seed)

        if camera_embedding_dim > 0:
            self.camera_embeddings = torch.nn.Embedding(160, camera_embedding_dim)  # humanrf.py:76-77
            with torch.no_grad():
                self.camera_embeddings.weight.copy_(torch.randn(160, camera_embedding_dim, generator=gen))

        f2s, f2l = hashgrid.frame_tables(sorted_frame_numbers, segment_sizes)
        self._f2s_host = f2s   # host copy (data-parallel exchange: segments of the frames in the pools)
        self.register_buffer("frame_numbers_to_segment_numbers", torch.from_numpy(f2s))
        self.register_buffer("frame_numbers_to_normalized_local_frame_numbers", torch.from_numpy(f2l))
        # frame number -> rank among the sorted frames: scheduling key of the prune march (ops.ray_segment_order)
        rank = torch.zeros(len(f2s), dtype=torch.int32)
        rank[torch.as_tensor(list(sorted_frame_numbers), dtype=torch.long)] = torch.arange(len(sorted_frame_numbers), dtype=torch.int32)
        self.register_buffer("_frame_rank", rank, persistent=False)
        self.num_frames = len(sorted_frame_numbers)

        metas, self.entries_per_segment, total_entries = hashgrid.build_segment_meta(
            segment_sizes, n_levels, log2_hashmap_size, coarsest_resolution, finest_resolution)
        self._metas_host = metas
        self.total_entries = total_entries
        # largest level table (entries) of any segment: decides whether the binned gradient scatter serves the model
        self.max_level_entries = max(int(metas[s].levels[l].size) for s in range(len(segment_sizes)) for l in range(n_levels))
        meta_bytes = bytes(metas)
        self.register_buffer("_seg_meta", torch.frombuffer(bytearray(meta_bytes), dtype=torch.uint8).clone(),
                             persistent=False)

        # tcnn grid init: U(-1e-4, 1e-4) (A.1); vectors: randn * 0.1 (decomposition4d.py:76-78)
        self.table_params = torch.nn.Parameter((torch.rand(total_entries * 2, generator=gen) * 2.0 - 1.0) * 1e-4)
        # 1-D vectors: (S, 4, R, 2 n_levels) in the reference; the kernels' rows are 32 wide, the columns of levels that do not
        # exist stay zero (their products with the zero per-encoding features give zero gradients)
        vec = torch.zeros(self.num_segments, 4, finest_resolution, 32)
        vec[..., :self.total_feature_dim] = torch.randn(self.num_segments, 4, finest_resolution, self.total_feature_dim,
                                                        generator=gen) * 0.1
        self.vectors = torch.nn.Parameter(vec)
        # sigma_net: (64, 32) + (16, 64) in the kernels' layout; the reference's first matrix is (64, sigma_in_pad): the columns
        # beyond it multiply zeros (any value would do; they start at zero and receive zero gradients)
        w1 = torch.zeros(64, 32)
        w1[:, :self.sigma_in_pad] = _xavier_uniform(64, self.sigma_in_pad, gen)
        self.sigma_params = torch.nn.Parameter(torch.cat([w1.reshape(-1), _xavier_uniform(16, 64, gen).reshape(-1)]))
        self.color_params = torch.nn.Parameter(torch.cat([
            _xavier_uniform(64, self.color_in_pad, gen).reshape(-1)]
            + [_xavier_uniform(64, 64, gen).reshape(-1) for _ in range(self.n_hidden_layers_color - 1)]
            + [_xavier_uniform(16, 64, gen).reshape(-1)]))
        # +2 halves: the paired 8-byte gather may read one entry past the last table (value unused)
        self.register_buffer("_tables_h", torch.zeros(total_entries * 2 + 2, dtype=torch.float16), persistent=False)
        # 16-bit copies of the MLP weights in the kernels' arithmetic type (their dtype selects it, ops._mlp_mode)
        w16 = torch.bfloat16 if mlp_precision == "bf16" else torch.float16
        self.register_buffer("_sigma_h", torch.empty(self.sigma_params.numel(), dtype=w16), persistent=False)
        self.register_buffer("_color_h", torch.empty(self.color_params.numel(), dtype=w16), persistent=False)
        self._half_versions = None
        # hooks a TrainEngine installs (data parallel, sharded exchange): wait for an in-flight all-gather of the fp16 tables
        # before they are read; complete the fp32 masters before they are serialised
        self._tables_ready = None
        self._master_sync = None
        self.to(dev)

    def __getstate__(self):
        """deepcopy / pickle of the model (an EMA copy, torch.save(model)): the hooks an attached TrainEngine installed are weak
        references to ITS bound methods (weakref.WeakMethod can be neither copied nor pickled, and a copy of the model is not the
        engine's model anyway); a copy starts without hooks and re-casts its 16-bit shadow tensors on first use."""
        state = self.__dict__.copy()
        state["_tables_ready"] = None
        state["_master_sync"] = None
        state["_half_versions"] = None
        state["_metas_host"] = bytes(self._metas_host)      # (a ctypes array type made on the fly has no importable name)
        return state

    def __setstate__(self, state):
        from .._lib import SegmentMeta
        raw = state["_metas_host"]
        if isinstance(raw, (bytes, bytearray)):
            state = dict(state)
            state["_metas_host"] = (SegmentMeta * (len(raw) // ctypes.sizeof(SegmentMeta))).from_buffer_copy(raw)
        super().__setstate__(state)

    # ------------------------------------------------------------------ fp16 shadow copies
    def _refresh_half(self) -> None:
        """The kernels read fp16 copies (tcnn keeps fp32 master params and casts per step, A.1); re-cast
        whenever an optimizer (or load_state_dict) touched the fp32 masters."""
        _call_hook(self._tables_ready)
        ver = (self.table_params._version, self.sigma_params._version, self.color_params._version,
               self.table_params.data_ptr(), self._tables_h.data_ptr())
        if ver != self._half_versions:
            with torch.no_grad():
                self._tables_h[:self.table_params.numel()].copy_(self.table_params)
                self._sigma_h.copy_(self.sigma_params)
                self._color_h.copy_(self.color_params)
            self._half_versions = (self.table_params._version, self.sigma_params._version,
                                   self.color_params._version, self.table_params.data_ptr(),
                                   self._tables_h.data_ptr())

    def mark_half_fresh(self) -> None:
        """Called by the fused optimizer, which refreshes the fp16 copies itself."""
        self._half_versions = (self.table_params._version, self.sigma_params._version, self.color_params._version,
                               self.table_params.data_ptr(), self._tables_h.data_ptr())

    def _sigma_w(self):
        return self._sigma_h[:2048], self._sigma_h[2048:]

    def split_color(self, flat: torch.Tensor):
        """tcnn's flat parameter layout of color_net [w1 (64, in_pad) | n_hidden - 1 matrices (64, 64) | w3 (16, 64)] as the three
        views the kernels take: first, stacked hidden-to-hidden (empty with one hidden layer), last."""
        a = 64 * self.color_in_pad
        b = a + 4096 * (self.n_hidden_layers_color - 1)
        return flat[:a], flat[a:b], flat[b:]

    def _color_w(self):
        return self.split_color(self._color_h)

    # ------------------------------------------------------------------ reference API
    def _xyzt_seg(self, positions: torch.Tensor, frame_numbers: torch.Tensor):
        fn = frame_numbers.reshape(-1).long()
        xyzt = torch.cat([positions.float() + 0.5,  # humanrf.py:175
                          self.frame_numbers_to_normalized_local_frame_numbers[fn].unsqueeze(-1)], dim=-1).contiguous()
        seg = self.frame_numbers_to_segment_numbers[fn].contiguous()
        return xyzt, seg

    def density(self, query_input: QueryInput) -> QueryOutput:
        """humanrf.py:158-186. Under torch.no_grad() (how prune_samples calls it, volume_rendering.py:41) this is the
        encode + sigma_net pair of kernels. With gradients enabled the density is differentiable with respect to the
        parameters, as in the reference (forward() differentiates through density(), humanrf.py:188-189); the geometry
        features it returns then carry no gradient of their own -- their gradient flows inside forward()."""
        xyzt, seg = self._xyzt_seg(query_input.positions, query_input.frame_numbers)
        if torch.is_grad_enabled() and any(p.requires_grad for p in (self.table_params, self.vectors, self.sigma_params)):
            n = xyzt.shape[0]
            idx = torch.arange(n, device=xyzt.device, dtype=torch.int64)
            dirs = torch.zeros(n, 3, dtype=torch.float32, device=xyzt.device)
            cams = torch.zeros(n, dtype=torch.int32, device=xyzt.device) if self.camera_embedding_dim > 0 else None
            sigma, _, geo = self.field(xyzt, seg, dirs, idx, cams, False)
            return QueryOutput(density=sigma, geometry_features=geo)
        with torch.no_grad():
            sigma, h = self.density_from_xyzt(xyzt, seg)
        return QueryOutput(density=sigma, geometry_features=h[:, 1:1 + self.geometry_feature_dim])

    @torch.no_grad()
    def density_from_xyzt(self, xyzt: torch.Tensor, seg: torch.Tensor):
        self._refresh_half()
        feats, _ = ops.encode4d_fwd(xyzt, seg, self._tables_h, self.vectors.detach(), self._seg_meta,
                                    self.num_segments, save_enc=False)
        sw1, sw2 = self._sigma_w()
        h, sigma = ops.density_mlp_fwd(feats, sw1, sw2, float(self.density_scale))
        return sigma, h

    def field(self, xyzt, seg, ray_dirs, sample_ray, ray_cameras, is_training: bool):
        """sigma (N,), radiance (N,3), geometry_features (N,15); differentiable w.r.t. the parameters."""
        emb = self.camera_embeddings.weight if self.camera_embedding_dim > 0 else None
        use_emb = bool(is_training and self.camera_embedding_dim > 0)  # zeros at eval, humanrf.py:196-204
        return _FieldFn.apply(self, xyzt, seg, ray_dirs, sample_ray, ray_cameras, use_emb, self.table_params,
                              self.vectors, self.sigma_params, self.color_params, emb)

    def forward(self, query_input: QueryInput) -> QueryOutput:
        """humanrf.py:188-208. `directions`/`camera_numbers` are per-sample here, as in the reference."""
        xyzt, seg = self._xyzt_seg(query_input.positions, query_input.frame_numbers)
        n = xyzt.shape[0]
        idx = torch.arange(n, device=xyzt.device, dtype=torch.int64)
        cams = None
        if self.camera_embedding_dim > 0:
            cams = (query_input.camera_numbers.reshape(-1).int().contiguous() if query_input.camera_numbers is not None
                    else torch.zeros(n, dtype=torch.int32, device=xyzt.device))
        sigma, rgb, geo = self.field(xyzt, seg, query_input.directions.float().contiguous(), idx, cams,
                                     query_input.is_training)
        return QueryOutput(density=sigma, geometry_features=geo, radiance=rgb)

    def get_params(self, lr):
        """Same grouping as humanrf.py:210-220: feature grids, sigma_net, color_net, (camera embeddings)."""
        params = [
            {"params": [self.table_params, self.vectors], "lr": lr},
            {"params": [self.sigma_params], "lr": lr},
            {"params": [self.color_params], "lr": lr},
        ]
        if self.camera_embedding_dim > 0:
            params.append({"params": self.camera_embeddings.parameters(), "lr": lr})
        return params

    # ------------------------------------------------------------------ checkpoint interchange
    def reference_state_dict(self) -> Dict[str, torch.Tensor]:
        """State dict with the reference's keys and layouts ('Checkpoint / resume'). Under a data-parallel
        TrainEngine with the sharded exchange this is a COLLECTIVE call (every rank gathers the other ranks' shards of the
        fp32 masters first); so is state_dict()."""
        _call_hook(self._master_sync)
        sd = {}
        names = ("xyz", "xyt", "yzt", "xzt")
        off = 0
        for s, entries in enumerate(self.entries_per_segment):
            sd[f"feature_grids.{s}.vectors"] = self.vectors[s, :, :, :self.total_feature_dim].detach().clone()
            for e, nm in enumerate(names):
                sd[f"feature_grids.{s}.{nm}_encoding.params"] = self.table_params[off * 2:(off + entries) * 2].detach().clone()
                off += entries
        # sigma_net's first matrix is (64, sigma_in_pad) in the reference (tcnn pads 2 n_levels inputs to a multiple of 16), (64, 32)
        # in the kernels' layout
        w1 = self.sigma_params[:2048].detach().reshape(64, 32)[:, :self.sigma_in_pad]
        sd["sigma_net.params"] = torch.cat([w1.reshape(-1), self.sigma_params[2048:].detach()]).clone()
        sd["color_net.params"] = self.color_params.detach().clone()
        if self.camera_embedding_dim > 0:
            sd["camera_embeddings.weight"] = self.camera_embeddings.weight.detach().clone()
        sd["frame_numbers_to_segment_numbers"] = self.frame_numbers_to_segment_numbers.clone()
        sd["frame_numbers_to_normalized_local_frame_numbers"] = self.frame_numbers_to_normalized_local_frame_numbers.clone()
        return sd

    def state_dict(self, *args, **kwargs):
        """Under a data-parallel TrainEngine with the sharded exchange a rank's fp32 masters are current on its own shards only:
        the engine's hook RAISES here when they are stale (call `engine.gather_master_tables()` on EVERY rank first -- then any
        single rank may serialise -- or take the checkpoint through `engine.state_dict()`, which is documented as collective)."""
        _call_hook(self._master_sync)
        return super().state_dict(*args, **kwargs)

    @torch.no_grad()
    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        names = ("xyz", "xyt", "yzt", "xzt")
        off = 0
        for s, entries in enumerate(self.entries_per_segment):
            self.vectors[s].zero_()
            self.vectors[s, :, :, :self.total_feature_dim].copy_(sd[f"feature_grids.{s}.vectors"])
            for e, nm in enumerate(names):
                self.table_params[off * 2:(off + entries) * 2].copy_(sd[f"feature_grids.{s}.{nm}_encoding.params"])
                off += entries
        sp = sd["sigma_net.params"]
        n1 = 64 * self.sigma_in_pad
        self.sigma_params[:2048].zero_()
        self.sigma_params[:2048].view(64, 32)[:, :self.sigma_in_pad].copy_(sp[:n1].reshape(64, self.sigma_in_pad))
        self.sigma_params[2048:].copy_(sp[n1:])
        self.color_params.copy_(sd["color_net.params"])
        if self.camera_embedding_dim > 0:
            self.camera_embeddings.weight.copy_(sd["camera_embeddings.weight"])
        self._half_versions = None
