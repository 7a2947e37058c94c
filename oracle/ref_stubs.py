"""
oracle/ref_stubs.py -- CPU stand-ins for the third-party modules the reference imports. TEST INFRASTRUCTURE ONLY.

The reference's own Python on the hot path (humanrf/volume_rendering.py, scene_representation/humanrf.py,
scene_representation/decomposition4d.py, trainer.py:205-255, input.py, utils/*.py, adaptive_temporal_partitioning.py)
imports and runs in this container once the packages that are NOT in /root/reference exist as modules:

  tinycudann        -> Encoding / Network / NetworkWithInputEncoding (nn.Modules with ONE flat fp32 `params`
                       Parameter, as tcnn's torch binding has; decomposition4d.py:79-122, humanrf.py:123-156)
                       whose arithmetic is oracle/hrf_oracle.py's restatement of tcnn
  nerfacc           -> the three functions volume_rendering.py:75-81,123-141 calls = hrf_oracle's restatement of
                       nerfacc 0.3.1 (A.4)
  humanrf.scene_representation.tensor_composition_native
                    -> compose_tensors_forward / _backward = hrf_oracle's restatement of tensor_composition.cu:9-219
                       (that .cu needs nvcc; its formulas are in the repository and are followed line by line)
  cv2, lpips, tensorboardX, skimage.metrics, simple_parsing -> inert (only imported, never reached on this path)

So what oracle/ref_harness.py executes is REFERENCE CODE for everything the reference repository itself owns
(control flow, tensor plumbing, per-segment masking, jitter, alpha, loss, optimizer wiring, batch merging, frame
tables, segment sizing) on top of OUR restatement of the arithmetic that lives in tcnn / nerfacc / the .cu file.
That pins the oracle's composition against the reference and leaves exactly three things unpinned: tcnn's kernel
arithmetic, nerfacc's scan order, and the CUDA texture unit.

dtype note: tcnn returns torch.half. The tcnn stubs return float32 tensors HOLDING fp16-representable values (the
compose op returns real torch.half, as the reference's does), because
the fp32 cast the reference relies on -- `custom_fwd(cast_inputs=torch.float32)` of truncated_exp
(utils/activation.py:8) under `torch.cuda.amp.autocast()` (trainer.py:145,175) -- only acts on CUDA tensors and CUDA
autocast cannot be enabled here; half -> float is exact, so the values are the ones the reference computes.
"""
from __future__ import annotations

import dataclasses
import math
import sys
import types
from typing import List
from unittest.mock import MagicMock

import numpy as np
import torch

from . import hrf_oracle as O


# ------------------------------------------------------------------------------------------------ tinycudann
class _HashGridEncoding(torch.nn.Module):
    """tcnn.Encoding(n_input_dims=3, {"otype": "HashGrid", ...}) (decomposition4d.py:79-122)."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
        super().__init__()
        if encoding_config.get("otype") != "HashGrid" or n_input_dims != 3:
            raise NotImplementedError("stub: only the 3-D HashGrid the reference instantiates")
        self.n_input_dims = n_input_dims
        self.F = int(encoding_config["n_features_per_level"])
        self.levels = O.hashgrid_levels(int(encoding_config["n_levels"]), int(encoding_config["log2_hashmap_size"]),
                                        int(encoding_config["base_resolution"]), float(encoding_config["per_level_scale"]))
        self.n_output_dims = len(self.levels) * self.F
        n = sum(lv.size for lv in self.levels) * self.F
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter((torch.rand(n, generator=g) * 2.0 - 1.0) * 1e-4)   # A.1 initialisation

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        table = O.round_half(self.params).reshape(-1, self.F)   # tcnn gathers from the fp16 copy of the fp32 masters
        return O.hashgrid_encode(x.float(), table, self.levels)


def Encoding(n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
    return _HashGridEncoding(n_input_dims, encoding_config, seed, dtype)


def _mlp_shapes(n_in_padded: int, n_out: int, network_config: dict) -> List[tuple]:
    if network_config.get("otype") != "FullyFusedMLP" or network_config.get("activation") != "ReLU":
        raise NotImplementedError("stub: only the FullyFusedMLP/ReLU networks the reference instantiates")
    width, hidden = int(network_config["n_neurons"]), int(network_config["n_hidden_layers"])
    out_pad = (n_out + 15) // 16 * 16
    shapes = [(width, n_in_padded)] + [(width, width)] * (hidden - 1) + [(out_pad, width)]
    return shapes


class _Mlp(torch.nn.Module):
    """Flat fp32 `params` = row-major (out, in) matrices, first -> last (A.2); Xavier-uniform initialisation."""

    def _init_mlp(self, n_in_padded: int, n_out: int, network_config: dict, seed: int):
        self.shapes = _mlp_shapes(n_in_padded, n_out, network_config)
        self.n_output_dims = n_out
        self.out_activation = network_config.get("output_activation", "None")
        g = torch.Generator().manual_seed(seed)
        ws = []
        for o, i in self.shapes:
            bound = math.sqrt(6.0 / (i + o))
            ws.append(((torch.rand(o, i, generator=g) * 2.0 - 1.0) * bound).reshape(-1))
        self.params = torch.nn.Parameter(torch.cat(ws))

    def _weights(self):
        ws, off = [], 0
        p = O.round_half(self.params)
        for o, i in self.shapes:
            ws.append(p[off:off + o * i].reshape(o, i))
            off += o * i
        return ws


class Network(_Mlp):
    """tcnn.Network(n_input_dims, n_output_dims, network_config) (humanrf.py:123-133)."""

    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: dict, seed: int = 1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.in_pad = (n_input_dims + 15) // 16 * 16
        self._init_mlp(self.in_pad, n_output_dims, network_config, seed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = O.round_half(x.float())
        if self.in_pad != self.n_input_dims:
            x = torch.cat([x, torch.ones(x.shape[0], self.in_pad - self.n_input_dims)], 1)
        return O.mlp(x, self._weights(), self.out_activation)[:, :self.n_output_dims]


class NetworkWithInputEncoding(_Mlp):
    """tcnn.NetworkWithInputEncoding with Composite[SphericalHarmonics(3 dims, degree 4), Identity] (humanrf.py:135-156)."""

    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: dict, network_config: dict, seed: int = 1337):
        super().__init__()
        nested = encoding_config.get("nested", [])
        ok = (encoding_config.get("otype") == "Composite" and len(nested) == 2
              and nested[0].get("otype") == "SphericalHarmonics" and nested[0].get("n_dims_to_encode") == 3
              and nested[0].get("degree") == 4 and nested[1].get("otype") == "Identity")
        if not ok:
            raise NotImplementedError("stub: only Composite[SphericalHarmonics(3, degree 4), Identity]")
        self.n_input_dims = n_input_dims
        enc_dims = 16 + (n_input_dims - 3)
        self._init_mlp((enc_dims + 15) // 16 * 16, n_output_dims, network_config, seed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.float()
        enc = O.color_net_input01(x[:, :3], x[:, 3:], None)
        return O.mlp(enc, self._weights(), self.out_activation)[:, :self.n_output_dims]


# ------------------------------------------------------------------------------------------------ nerfacc 0.3.1
def render_visibility(alphas, ray_indices, early_stop_eps=1e-4, alpha_thre=0.0, n_rays=None):
    return O.render_visibility(alphas.float(), ray_indices.long(), early_stop_eps, alpha_thre)


def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices, n_rays=None):
    return O.render_weight_from_density(t_starts, t_ends, sigmas, ray_indices.long()).reshape(-1, 1)


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    return O.accumulate_along_rays(weights, ray_indices.long(), None if values is None else values.float(), n_rays)


# ------------------------------------------------------------------------------------------------ tensor_composition_native
def compose_tensors_forward(xyz_f, xyt_f, yzt_f, xzt_f, vectors, xyzt):
    """tensor_composition.cu:120-160 (kernel :22-54). Returns torch.half like the real op: HumanRF.density writes the
    result into a half `features` buffer by boolean-mask assignment (humanrf.py:165-177), which requires equal dtypes."""
    with torch.no_grad():
        return O.compose_tensors(xyz_f.float(), xyt_f.float(), yzt_f.float(), xzt_f.float(), vectors.float(),
                                 xyzt.float()).half()


# The real op allocates its four per-encoding gradients with empty_like(xyz_features): at::Half (tensor_composition.cu:185-188,
# accessors :204-207), so they are ROUNDED TO HALF. The stand-in's encoding outputs are fp32 tensors holding half values (the
# dtype note above) and, until round 4, its four gradients stayed fp32, so the fixtures of rounds 2-3 carried the half rounding
# of the compose OUTPUT's gradient only. HALF_GRAD_OUTPUTS = True restates the real op; ref_render.npz, ref_steps_skip.npz and
# ref_step_weak.npz are generated that way (False reproduces the older fixtures).
HALF_GRAD_OUTPUTS = True


def compose_tensors_backward(xyz_f, xyt_f, yzt_f, xzt_f, vectors, xyzt, d_out):
    """tensor_composition.cu:162-219 (kernel :77-117): d_feat_e = v[pair(e)] * dY; d_vectors taps get feat*dY*(1-w | w)."""
    with torch.no_grad():
        dy = d_out.float()
        vec = vectors.float()
        sv = O.vectors_sample(vec, xyzt.float())
        d_xyz, d_xyt, d_yzt, d_xzt = sv[3] * dy, sv[2] * dy, sv[0] * dy, sv[1] * dy
        if HALF_GRAD_OUTPUTS:
            d_xyz, d_xyt, d_yzt, d_xzt = (g.half().float() for g in (d_xyz, d_xyt, d_yzt, d_xzt))
        feats = [yzt_f.float(), xzt_f.float(), xyt_f.float(), xyz_f.float()]   # pair with vectors 0..3 (:47-54)
        Rv = vec.shape[1]
        d_vec = torch.zeros_like(vec)
        for i in range(4):
            coord = xyzt[:, i].float() * float(Rv) - 0.5
            fl = torch.floor(coord)
            fr = (coord - fl).unsqueeze(1)
            c0 = torch.clamp(fl, 0.0, float(Rv - 1)).long()
            c1 = torch.clamp(fl + 1.0, 0.0, float(Rv - 1)).long()
            dval = feats[i] * dy
            d_vec[i].index_add_(0, c0, dval * (1.0 - fr))
            d_vec[i].index_add_(0, c1, dval * fr)
        return [d_xyz, d_xyt, d_yzt, d_xzt, d_vec]


# ------------------------------------------------------------------------------------------------ installation
def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def _simple_parsing_field(default=dataclasses.MISSING, **kwargs):
    if default is dataclasses.MISSING:
        return dataclasses.field()
    return dataclasses.field(default=default)


def install() -> None:
    """Register the stand-ins in sys.modules (idempotent). Call before importing anything from /root/reference."""
    me = sys.modules[__name__]
    sys.modules.setdefault("tinycudann", _module("tinycudann", Encoding=Encoding, Network=Network,
                                                 NetworkWithInputEncoding=NetworkWithInputEncoding, __stub__=True))
    sys.modules.setdefault("nerfacc", _module("nerfacc", render_visibility=render_visibility,
                                              render_weight_from_density=render_weight_from_density,
                                              accumulate_along_rays=accumulate_along_rays, __stub__=True))
    sys.modules.setdefault("humanrf.scene_representation.tensor_composition_native",
                           _module("humanrf.scene_representation.tensor_composition_native",
                                   compose_tensors_forward=me.compose_tensors_forward,
                                   compose_tensors_backward=me.compose_tensors_backward, __stub__=True))
    sys.modules.setdefault("simple_parsing", _module("simple_parsing", field=_simple_parsing_field,
                                                     ArgumentGenerationMode=MagicMock(), ArgumentParser=MagicMock(),
                                                     NestedMode=MagicMock()))
    for name in ("cv2", "lpips", "tensorboardX", "skimage", "skimage.metrics",
                 "actorshq.dataset.occupancy_grid_native", "actorshq.dataset.ray_sampler_native"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()
    _ = np  # numpy is a dependency of the reference modules; imported here so a missing numpy fails early
