"""
oracle/hrf_oracle.py -- CPU restatement of HumanRF's ray-marching hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (humanrf_amd/) never does. Everything here is plain NumPy / PyTorch-CPU (+ the C sampler in
sampler_oracle.c); every function cites the reference file:line it follows.

PARITY STATUS. The reference has no tests, golden vectors or fixtures. Its own Python on this path DOES run here
(oracle/ref_harness.py imports it from /root/reference over stand-ins for the absent third-party packages), and this
oracle is pinned against it: tests/test_cpu_ref_fixtures.py checks model_features / model_density / model_forward /
prune_samples / render / training_loss (+ autograd + torch.optim.Adam) against outputs of the reference's
HumanRF.density/forward, Decomposition4D.forward, prune_samples, render and Trainer.train_step frozen in
tests/golden/ref_*.npz (values exact, gradients to the noise of the reference's fp16 gradient tensors).
STILL UNPINNED -- the arithmetic that lives in third-party dependencies absent from /root/reference, restated here
from their published algorithms:
  * tiny-cuda-nn (unpinned git HEAD, README.md:19): HashGrid encoding, Composite[SphericalHarmonics,
    Identity] encoding, FullyFusedMLP;
  * nerfacc==0.3.1 (requirements.txt:3): render_visibility, render_weight_from_density,
    accumulate_along_rays (scan order);
  * the CUDA texture unit behind the sampler's occupancy predicate (sampler_oracle.c).
Those are pinned only by analytic known-answer tests and second restatements (tests/test_oracle_kat.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """Load (building if needed) the C part of the oracle."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhrf_oracle.so")
        srcs = [os.path.join(_HERE, f) for f in ("sampler_oracle.c", "occgen_oracle.c", "encode_oracle.c")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libhrf_oracle.so"])
        _LIB = ctypes.CDLL(so)
        _LIB.orc_samples.restype = ctypes.c_int64
        _LIB.orc_tex_gt0.restype = ctypes.c_int
    return _LIB


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------------------------
# Occupancy grids from masks  (actorshq/toolbox/native/occupancy_grid_generation.cu, oracle/occgen_oracle.c)
# ----------------------------------------------------------------------------------------------

def grid_from_masks(masks: np.ndarray, projection_matrices: np.ndarray, landscape_modes: np.ndarray,
                    camera_coverage_threshold: int, grid_resolution: int, width: int, height: int) -> np.ndarray:
    """generate_from_masks (occupancy_grid_generation.cu:82-120). projection_matrices: (C,4,4) as the driver passes
    them, i.e. world->pixel matrices TRANSPOSED so that the flat memory is column-major
    (generate_occupancy_grids_from_masks.py:54-61). -> (G,G,G) uint8 [z][y][x]."""
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    proj = np.ascontiguousarray(projection_matrices, dtype=np.float32)
    land = np.ascontiguousarray(landscape_modes, dtype=np.uint8)
    C = proj.shape[0]
    if masks.shape[1] != width * height:
        raise RuntimeError("The number mask entries per camera has to be equal to width*height!")
    G = int(grid_resolution)
    out = np.empty((G, G, G), dtype=np.uint8)
    _lib().orc_grid_from_masks(_p(masks), _p(proj), _p(land), ctypes.c_int(camera_coverage_threshold), ctypes.c_int(C),
                               ctypes.c_int(G), ctypes.c_int(width), ctypes.c_int(height), _p(out))
    return out


def mask_dilate(masks: np.ndarray, kernel_size: int) -> np.ndarray:
    """cv2.dilate(mask, ones((k,k)), iterations=1) per image; masks (N,H,W) uint8."""
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    n, h, w = masks.shape
    out = np.empty_like(masks)
    _lib().orc_mask_dilate(_p(masks), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(kernel_size), ctypes.c_int64(n), _p(out))
    return out


# ----------------------------------------------------------------------------------------------
# Sampler  (actorshq/dataset/native/ray_sampler.cu)
# ----------------------------------------------------------------------------------------------

def tex_gt0(grid: np.ndarray, x: float, y: float, z: float) -> bool:
    """tex3D<float>(grid, x, y, z) > 0 with the texture descriptor of occupancy_grid.cu:28-35."""
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    return bool(_lib().orc_tex_gt0(_p(grid), ctypes.c_int(grid.shape[0]), ctypes.c_float(x),
                                   ctypes.c_float(y), ctypes.c_float(z)))


def sampler_get_data(
    rgba_u8: np.ndarray,            # (B*P, 4) uint8
    light_mask: Optional[np.ndarray],  # (B*P,) bool
    frame_numbers: np.ndarray,      # (B,) int32
    camera_numbers: np.ndarray,     # (B,) int32
    grids: Optional[Sequence[np.ndarray]],  # B volumes (G,G,G) uint8 (entries may alias), None in aabb mode
    landscape_modes: np.ndarray,    # (B,) bool
    all_ray_indices: np.ndarray,    # (R0,) int64
    inverse_krs: np.ndarray,        # (B,3,3) float32, "column-major" (already transposed, data_loader.py:194-207)
    camera_origins: np.ndarray,     # (B,3) float32
    aabb: np.ndarray,               # (2,3) float32
    grid_resolution: int,
    image_width: int,
    image_height: int,
    raymarching_step_size: float,
    filter_light_bloom: bool,
    occupancy: bool,
    get_samples: bool,
):
    """get_data<kOccupancyMinmax, kGetSamples> (ray_sampler.cu:196-325) including the host-side
    compaction semantics (:254-290, :314-324). Returns the same 9 arrays in the same order."""
    lib = _lib()
    f32 = np.float32
    idx = np.ascontiguousarray(all_ray_indices, dtype=np.int64)
    R0 = idx.shape[0]
    ikr = np.ascontiguousarray(inverse_krs, dtype=f32)
    org = np.ascontiguousarray(camera_origins, dtype=f32)
    ab = np.ascontiguousarray(aabb, dtype=f32)
    land = np.ascontiguousarray(landscape_modes).astype(np.uint8)
    dirs = np.empty((R0, 3), f32)
    mm = np.empty((R0, 2), f32)
    mask = np.empty((R0,), np.uint8)
    gptrs = None
    if occupancy:
        gl = [np.ascontiguousarray(g, dtype=np.uint8) for g in grids]
        gptrs = (ctypes.c_void_p * len(gl))(*[g.ctypes.data for g in gl])
    lib.orc_minmax(_p(ikr), _p(org), _p(land), _p(idx), gptrs, _p(ab), ctypes.c_int64(R0),
                   ctypes.c_int(grid_resolution), ctypes.c_int(image_width), ctypes.c_int(image_height),
                   ctypes.c_int(1 if occupancy else 0), _p(dirs), _p(mm), _p(mask))
    ray_mask = mask.astype(bool)
    if filter_light_bloom:
        ray_mask = ray_mask & ~np.asarray(light_mask).reshape(-1)[idx]         # :254-257
    ray_indices = idx[ray_mask]                                               # :258
    s_dirs = dirs[ray_mask]
    s_mm = mm[ray_mask]
    s_rgba = rgba_u8.reshape(-1, 4)[ray_indices].astype(f32) / f32(255.0)     # :262
    image_numbers = ray_indices // (image_width * image_height)               # :263
    s_org = org[image_numbers]
    s_frames = np.asarray(frame_numbers, dtype=np.int32)[image_numbers]
    s_cams = np.asarray(camera_numbers, dtype=np.int32)[image_numbers]
    if not get_samples:
        return (s_org, s_dirs, s_rgba, s_frames, s_cams, s_mm, ray_mask,
                np.empty((0,), f32), np.empty((0,), np.int32))
    R = ray_indices.shape[0]
    counts = np.empty((R,), np.int32)
    s_mm = np.ascontiguousarray(s_mm)
    lib.orc_counts(_p(s_mm), ctypes.c_int64(R), ctypes.c_float(raymarching_step_size), _p(counts))
    s_org_c = np.ascontiguousarray(s_org)
    s_dirs_c = np.ascontiguousarray(s_dirs)
    rgrids = None
    if occupancy:
        rgrids = (ctypes.c_void_p * max(R, 1))(*[gl[i].ctypes.data for i in image_numbers])
    n = lib.orc_samples(_p(s_org_c), _p(s_dirs_c), _p(s_mm), _p(counts), rgrids, ctypes.c_int64(R),
                        ctypes.c_int(grid_resolution), ctypes.c_float(raymarching_step_size), None, None)
    t = np.empty((n,), f32)
    ray = np.empty((n,), np.int32)
    lib.orc_samples(_p(s_org_c), _p(s_dirs_c), _p(s_mm), _p(counts), rgrids, ctypes.c_int64(R),
                    ctypes.c_int(grid_resolution), ctypes.c_float(raymarching_step_size), _p(t), _p(ray))
    return (s_org, s_dirs, s_rgba, s_frames, s_cams, s_mm, ray_mask, t, ray)


# ----------------------------------------------------------------------------------------------
# tcnn HashGrid  (call sites decomposition4d.py:79-122; semantics SURVEY.md A.1 [UPSTREAM-KNOWLEDGE])
# ----------------------------------------------------------------------------------------------

@dataclass
class Level:
    scale: float     # fp32 value
    res: int
    size: int        # entries in this level
    offset: int      # first entry of this level inside the encoding's table
    hashed: bool


def hashgrid_levels(n_levels: int, log2_hashmap_size: int, base_resolution: int,
                    per_level_scale: float) -> List[Level]:
    """Per-level geometry of a tcnn HashGrid (A.1). All float steps in fp32, like tcnn."""
    f32 = np.float32
    log2_pls = np.log2(f32(per_level_scale)).astype(f32)
    levels: List[Level] = []
    offset = 0
    for l in range(n_levels):
        scale = f32(np.exp2(f32(f32(l) * log2_pls)).astype(f32) * f32(base_resolution) - f32(1.0))
        res = int(np.ceil(scale)) + 1
        n = min(res ** 3, 2 ** 31 - 1)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        levels.append(Level(float(scale), res, n, offset, res ** 3 > n))
        offset += n
    return levels


_PRIMES = (1, 2654435761, 805459861)


def hashgrid_indices(x: torch.Tensor, lv: Level) -> Tuple[torch.Tensor, torch.Tensor]:
    """Corner entry indices (N,8) int64 (level-local) and trilinear weights (N,8) fp32 for one level."""
    scale = torch.tensor(lv.scale, dtype=torch.float32)
    # pos = fmaf(x, scale, 0.5): emulate the single rounding in float64 (exact product of two fp32 fits)
    pos = (x.double() * scale.double() + 0.5).float()
    g = torch.floor(pos)
    w = pos - g
    gi = g.long()
    idxs, wts = [], []
    for c in range(8):
        weight = torch.ones_like(w[:, 0])
        cc = []
        for d in range(3):
            if (c >> d) & 1:
                weight = weight * w[:, d]
                cc.append(gi[:, d] + 1)
            else:
                weight = weight * (1.0 - w[:, d])
                cc.append(gi[:, d])
        if lv.hashed:
            h = torch.zeros_like(cc[0])
            for d in range(3):
                h = h ^ ((cc[d] * _PRIMES[d]) & 0xFFFFFFFF)
            index = h % lv.size
        else:
            index = ((cc[0] + cc[1] * lv.res + cc[2] * lv.res * lv.res) & 0xFFFFFFFF) % lv.size
        idxs.append(index)
        wts.append(weight)
    return torch.stack(idxs, 1), torch.stack(wts, 1)


def hashgrid_encode(x: torch.Tensor, table: torch.Tensor, levels: Sequence[Level], accumulate: str = "fp32") -> torch.Tensor:
    """x (N,3) fp32 in [0,1]; table (entries, F) holding the fp16-representable parameter values.
    Returns (N, L*F) rounded to half precision (tcnn writes __half outputs), level-major features.

    accumulate="fp32" (the definition the HIP kernels are held to, enc_gather in csrc/encode_common.h): fp32 weights, the eight
    corners summed in fp32 in corner order, ONE rounding to half at the end.
    accumulate="fp16" / "fp16_legacy" [UPSTREAM-KNOWLEDGE: tiny-cuda-nn is not in /root/reference, SURVEY.md Appendix A]: what
    tcnn's kernel_grid does with T = __half parameters -- a HALF accumulator, eight roundings. "fp16": the current form,
    `result = fma((T)weight, val, result)` (weight rounded to half, one rounding per fused multiply-add); "fp16_legacy": the
    earlier form, `result += (T)(weight * (float)val)` (fp32 product rounded to half, then a half add). No gradient; these
    modes exist to BOUND what the fp32 definition deviates from the upstream arithmetic by
    (tests/test_oracle_kat.py::test_half_accumulate_hashgrid_bounds_the_fp32_accumulate_deviation)."""
    if accumulate not in ("fp32", "fp16", "fp16_legacy"):
        raise ValueError("accumulate must be 'fp32', 'fp16' or 'fp16_legacy'")
    outs = []
    for lv in levels:
        idx, w = hashgrid_indices(x, lv)
        vals = table[lv.offset + idx]                     # (N,8,F)
        if accumulate == "fp32":
            acc = torch.zeros(x.shape[0], table.shape[1], dtype=torch.float32)
            for c in range(8):                                 # fp32 accumulate in corner order
                acc = acc + w[:, c:c + 1] * vals[:, c].float()
            outs.append(acc)
            continue
        # half accumulator; every step is computed exactly in float64 (products of two halves and their sum with a half fit)
        # and rounded once by numpy's correctly rounding float64 -> float16 conversion
        v64 = vals.detach().numpy().astype(np.float16).astype(np.float64)
        w32 = w.detach().numpy().astype(np.float32)
        acc16 = np.zeros((x.shape[0], table.shape[1]), dtype=np.float16)
        for c in range(8):
            if accumulate == "fp16":
                wh = w32[:, c].astype(np.float16).astype(np.float64)[:, None]
                acc16 = (wh * v64[:, c] + acc16.astype(np.float64)).astype(np.float16)
            else:
                prod = (w32[:, c:c + 1] * v64[:, c].astype(np.float32)).astype(np.float32).astype(np.float16)
                acc16 = (prod.astype(np.float64) + acc16.astype(np.float64)).astype(np.float16)
        outs.append(torch.from_numpy(acc16.astype(np.float32)))
    return round_half(torch.cat(outs, 1))


class _RoundHalf(torch.autograd.Function):
    """fp16 rounding with an exact straight-through gradient. (`x.half().float()` would also round the GRADIENT to
    fp16 on the way back -- unscaled gradients of ~1e-6 land in the subnormal range and pick up percent-level
    noise, which is an artefact of the emulation, not of the path being restated.)"""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def round_half(x: torch.Tensor) -> torch.Tensor:
    """Round to fp16 and come back to fp32 (gradient passes straight through, in fp32)."""
    return _RoundHalf.apply(x)


class _HalfGradient(torch.autograd.Function):
    """Identity whose GRADIENT is what a torch.half tensor would carry. In the reference the compose op's output and its four
    per-encoding inputs' gradients are half tensors (decomposition4d.py:8-39: the autograd Function returns what
    compose_tensors_backward writes, tensor_composition.cu:85-117, and its own output is half, so autograd hands dL/d(output)
    over in half), at whatever scale the caller's GradScaler gave the loss. `scale` is that factor when this oracle
    differentiates an UNSCALED loss: g -> half(g * scale) / scale. Contributions below 2^-25 / scale vanish, as they do there."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = float(scale)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g * ctx.scale).half().float() / ctx.scale, None


def half_gradient(x: torch.Tensor, scale: float) -> torch.Tensor:
    return _HalfGradient.apply(x, scale) if scale else x


class _RoundBF16(torch.autograd.Function):
    """bf16 rounding (nearest even), straight-through gradient: the arithmetic of the product's mlp_precision="bf16"
    variant (BASELINE.json configs[4]). The reference has no bf16 configuration, so this mode is pinned by nothing but
    its definition -- the fp16 mode is the one the reference fixtures pin."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def round_bf16(x: torch.Tensor) -> torch.Tensor:
    return _RoundBF16.apply(x)


# ----------------------------------------------------------------------------------------------
# compose_tensors  (humanrf/scene_representation/native/tensor_composition.cu:22-54)
# ----------------------------------------------------------------------------------------------

def vectors_sample(vectors: torch.Tensor, xyzt: torch.Tensor) -> torch.Tensor:
    """vectors (4,Rv,F) fp32, xyzt (N,4) in [0,1] -> (4,N,F): linear interpolation with the
    `coord*Rv - 0.5`, clamped-tap convention of tensor_composition.cu:37-45."""
    Rv = vectors.shape[1]
    out = []
    for i in range(4):
        coord = xyzt[:, i] * float(Rv) - 0.5
        fl = torch.floor(coord)
        fr = (coord - fl).unsqueeze(1)
        # The reference clamps c0 only from below and c1 only from above (tensor_composition.cu:41-42) and
        # reads out of bounds for coordinates outside [0,1]; the build clamps both taps to [0, Rv-1].
        c0 = torch.clamp(fl, min=0.0, max=float(Rv - 1)).long()
        c1 = torch.clamp(fl + 1.0, min=0.0, max=float(Rv - 1)).long()
        v0 = vectors[i][c0]
        v1 = vectors[i][c1]
        out.append(v0 + fr * (v1 - v0))
    return torch.stack(out, 0)


def compose_tensors(xyz_f, xyt_f, yzt_f, xzt_f, vectors, xyzt) -> torch.Tensor:
    """tensor_composition.cu:47-54: xyz*v_t + xyt*v_z + yzt*v_x + xzt*v_y, rounded to half."""
    sv = vectors_sample(vectors, xyzt)
    res = xyz_f * sv[3] + xyt_f * sv[2] + yzt_f * sv[0] + xzt_f * sv[1]
    return round_half(res)


class _OrcLevel(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_float), ("res", ctypes.c_int32), ("size", ctypes.c_int64), ("offset", ctypes.c_int64),
                ("hashed", ctypes.c_int32)]


# > 0: decomposition4d() calls made with grad mode OFF (the pruning pass) run oracle/encode_oracle.c on that many OpenMP threads -- the
# same values bit for bit (tests/test_oracle_kat.py); bench.py's cpu_baseline sets it for its second, all-cores timing. 0: torch only.
C_ENCODE_THREADS = 0


def decomposition4d_c(xyzt: torch.Tensor, tables: Sequence[torch.Tensor], vectors: torch.Tensor, levels: Sequence[Level],
                      threads: int) -> torch.Tensor:
    """oracle/encode_oracle.c: Decomposition4D.forward on `threads` OpenMP threads (no autograd)."""
    x = np.ascontiguousarray(xyzt.detach().numpy(), dtype=np.float32)
    tabs = [np.ascontiguousarray(t.detach().numpy(), dtype=np.float32) for t in tables]
    vec = np.ascontiguousarray(vectors.detach().numpy(), dtype=np.float32)
    lv = (_OrcLevel * len(levels))(*[_OrcLevel(l.scale, l.res, l.size, l.offset, int(l.hashed)) for l in levels])
    ptrs = (ctypes.c_void_p * 4)(*[t.ctypes.data for t in tabs])
    out = np.empty((x.shape[0], 2 * len(levels)), dtype=np.float32)
    _lib().orc_decomposition4d_fwd(_p(x), ptrs, _p(vec), lv, ctypes.c_int64(x.shape[0]), len(levels), vec.shape[1], _p(out), int(threads))
    return torch.from_numpy(out)


def decomposition4d(xyzt: torch.Tensor, tables: Sequence[torch.Tensor], vectors: torch.Tensor,
                    levels: Sequence[Level], half_gradient_scale: float = 0.0) -> torch.Tensor:
    """Decomposition4D.forward (decomposition4d.py:124-135). tables = (xyz, xyt, yzt, xzt).
    half_gradient_scale > 0: the gradients of the compose op's output and of its four per-encoding inputs pass through half
    at that scale, as the reference's half tensors make them (see _HalfGradient); 0: fp32 gradients throughout."""
    hg = half_gradient_scale
    if C_ENCODE_THREADS > 0 and not torch.is_grad_enabled() and len(levels) * 2 == vectors.shape[2]:
        return decomposition4d_c(xyzt, tables, vectors, levels, C_ENCODE_THREADS)
    xyz_f = half_gradient(hashgrid_encode(xyzt[:, [0, 1, 2]], tables[0], levels), hg)
    xyt_f = half_gradient(hashgrid_encode(xyzt[:, [0, 1, 3]], tables[1], levels), hg)
    yzt_f = half_gradient(hashgrid_encode(xyzt[:, [1, 2, 3]], tables[2], levels), hg)
    xzt_f = half_gradient(hashgrid_encode(xyzt[:, [0, 2, 3]], tables[3], levels), hg)
    return half_gradient(compose_tensors(xyz_f, xyt_f, yzt_f, xzt_f, vectors, xyzt), hg)


# ----------------------------------------------------------------------------------------------
# tcnn FullyFusedMLP / Composite encoding  (humanrf.py:123-156)
# ----------------------------------------------------------------------------------------------

def mlp(x: torch.Tensor, weights: Sequence[torch.Tensor], out_activation: str, precision: str = "fp16",
        accumulate: str = "fp32") -> torch.Tensor:
    """Bias-free MLP, weights[i] (out,in) holding fp16-representable values, ReLU between layers, activations rounded to
    half after every layer (A.2).
    accumulate="fp32" (default): products summed in fp32 -- what the gfx950 kernels do on the matrix cores
      (v_mfma_f32_16x16x16_f16). accumulate="fp16": tcnn's FullyFusedMLP keeps its accumulator fragments in __half
      (wmma::fragment<accumulator, 16, 16, 16, __half>, fully_fused_mlp.cu): the sum over each block of 16 inputs is formed
      exactly and ADDED INTO A HALF accumulator, i.e. rounded to half after every block. (What happens inside one
      16-deep tensor-core instruction is not specified by NVIDIA; this is its documented contract.) The mode exists to
      BOUND the deviation the fp32 accumulation introduces (tests/test_oracle_kat.py), fp16 precision only.
    precision="bf16": the same network with every rounding going to bf16 (input included: it arrives as half values)
    and bf16-representable weights; the output is a bf16 value stored in the half tensor the next stage reads."""
    if accumulate not in ("fp32", "fp16") or (accumulate == "fp16" and precision != "fp16"):
        raise ValueError("accumulate must be 'fp32' or 'fp16' (fp16 only with precision='fp16')")
    rnd = round_half if precision == "fp16" else round_bf16
    h = x if precision == "fp16" else round_bf16(x)
    for i, w in enumerate(weights):
        if accumulate == "fp16":
            acc = torch.zeros(h.shape[0], w.shape[0], dtype=h.dtype)
            for k0 in range(0, w.shape[1], 16):
                acc = round_half(acc + h[:, k0:k0 + 16] @ w[:, k0:k0 + 16].t())
            h = acc
        else:
            h = h @ w.t()
        if i + 1 < len(weights):
            h = rnd(torch.relu(h))
    if out_activation == "Sigmoid":
        h = torch.sigmoid(h)
    elif out_activation != "None":
        raise ValueError(out_activation)
    return round_half(rnd(h)) if precision != "fp16" else round_half(h)


_SH = dict(c0=0.28209479177387814, c1=0.48860251190291987, c2a=1.0925484305920792,
           c2b=0.94617469575755997, c2c=0.31539156525251999, c2d=0.54627421529603959,
           c3a=0.59004358992664352, c3b=2.8906114426405538, c3c=0.45704579946446572,
           c3d=0.3731763325901154, c3e=1.4453057213202769)


def sh16(d01: torch.Tensor) -> torch.Tensor:
    """tcnn SphericalHarmonics degree 4 on inputs in [0,1] (mapped to 2x-1), A.3. -> (N,16) fp32."""
    v = d01 * 2.0 - 1.0
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    s = _SH
    out = [
        torch.full_like(x, s["c0"]),
        -s["c1"] * y, s["c1"] * z, -s["c1"] * x,
        s["c2a"] * xy, -s["c2a"] * yz, s["c2b"] * z2 - s["c2c"], -s["c2a"] * xz, s["c2d"] * (x2 - y2),
        s["c3a"] * y * (-3.0 * x2 + y2), s["c3b"] * xy * z, s["c3c"] * y * (1.0 - 5.0 * z2),
        s["c3d"] * z * (5.0 * z2 - 3.0), s["c3c"] * x * (1.0 - 5.0 * z2), s["c3e"] * z * (x2 - y2),
        s["c3a"] * x * (-x2 + 3.0 * y2),
    ]
    return torch.stack(out, 1)


def color_net_input01(d01: torch.Tensor, geo: torch.Tensor, cam_emb: Optional[torch.Tensor],
                      precision: str = "fp16") -> torch.Tensor:
    """Composite[SH(3 dims), Identity(rest)] padded with ones to a multiple of 16 (A.3), rounded to half.
    d01: the first three input dims as tcnn receives them, in [0,1]."""
    parts = [sh16(d01), geo.float()]
    if cam_emb is not None:
        parts.append(cam_emb.float())
    enc = torch.cat(parts, 1)
    pad = (-enc.shape[1]) % 16
    if pad:
        enc = torch.cat([enc, torch.ones(enc.shape[0], pad)], 1)
    return round_half(enc) if precision == "fp16" else round_bf16(enc)


def color_net_input(directions: torch.Tensor, geo: torch.Tensor, cam_emb: Optional[torch.Tensor],
                    precision: str = "fp16") -> torch.Tensor:
    """directions in [-1,1]; humanrf.py:192 maps them to [0,1] before the colour network's encoding."""
    return color_net_input01((directions + 1.0) * 0.5, geo, cam_emb, precision)


def truncated_exp(x: torch.Tensor) -> torch.Tensor:
    """humanrf/utils/activation.py:6-21: exp forward, backward uses exp(clamp(x, -15, 15))."""
    return _TruncExp.apply(x)


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return dy * torch.exp(x.clamp(-15.0, 15.0))


# ----------------------------------------------------------------------------------------------
# Model container (HumanRF, humanrf/scene_representation/humanrf.py)
# ----------------------------------------------------------------------------------------------

@dataclass
class OracleModel:
    """Parameters of a HumanRF model in oracle form. tables[s][e] is (entries,2) with fp16-representable
    values, e in (xyz, xyt, yzt, xzt); vectors[s] is (4,Rv,32) fp32; MLP weights hold fp16 values."""
    levels: List[List[Level]]                 # per segment
    tables: List[List[torch.Tensor]]          # per segment x 4
    vectors: List[torch.Tensor]               # per segment
    sigma_w: List[torch.Tensor]               # [(64,32), (16,64)]
    color_w: List[torch.Tensor]               # [(64,32|48), (64,64), (16,64)]
    frame_to_segment: torch.Tensor            # (max_frame+1,) int
    frame_to_local: torch.Tensor              # (max_frame+1,) fp32
    density_scale: float = 100.0
    camera_embeddings: Optional[torch.Tensor] = None   # (160,E)
    mlp_precision: str = "fp16"                         # "bf16": see mlp()
    mlp_accumulate: str = "fp32"                        # "fp16": tcnn's half accumulator fragments, see mlp()
    half_gradient_scale: float = 0.0                    # > 0: the reference's half gradient tensors around the compose op, see decomposition4d()

    def parameters(self) -> List[torch.Tensor]:
        ps = [t for seg in self.tables for t in seg] + list(self.vectors) + self.sigma_w + self.color_w
        if self.camera_embeddings is not None:
            ps.append(self.camera_embeddings)
        return ps


def model_features(m: OracleModel, positions: torch.Tensor, frame_numbers: torch.Tensor) -> torch.Tensor:
    """Feature part of HumanRF.density (humanrf.py:159-179): per-segment Decomposition4D of
    (positions + 0.5, normalized local frame number)."""
    fn = frame_numbers.reshape(-1).long()
    seg = m.frame_to_segment[fn]
    feats = torch.zeros(positions.shape[0], 32)
    for s in range(len(m.tables)):
        sel = (seg == s).nonzero().reshape(-1)
        if sel.numel() == 0:
            continue
        xyz = positions[sel] + 0.5
        t = m.frame_to_local[fn[sel]].unsqueeze(1)
        f = decomposition4d(torch.cat([xyz, t], 1), m.tables[s], m.vectors[s], m.levels[s], m.half_gradient_scale)
        feats = feats.index_put((sel,), f)
    return feats


def model_density(m: OracleModel, positions, frame_numbers):
    """HumanRF.density (humanrf.py:158-186) -> (sigma fp32 (N,), geometry_features half-valued (N,15), h)."""
    feats = model_features(m, positions, frame_numbers)
    h = mlp(feats, m.sigma_w, "None", m.mlp_precision, m.mlp_accumulate)
    sigma = truncated_exp(h[:, 0]) * m.density_scale
    return sigma, h[:, 1:], feats


def model_forward(m: OracleModel, positions, directions, frame_numbers, camera_numbers, is_training: bool):
    """HumanRF.forward (humanrf.py:188-208) -> (sigma (N,), radiance (N,3))."""
    sigma, geo, _ = model_density(m, positions, frame_numbers)
    emb = None
    if m.camera_embeddings is not None:
        if is_training:
            emb = m.camera_embeddings[camera_numbers.reshape(-1).long()]
        else:
            emb = torch.zeros(positions.shape[0], m.camera_embeddings.shape[1])
    x = color_net_input(directions, geo, emb, m.mlp_precision)
    rgb = mlp(x, m.color_w, "Sigmoid", m.mlp_precision, m.mlp_accumulate)[:, :3]
    return sigma, rgb


# ----------------------------------------------------------------------------------------------
# nerfacc 0.3.1 (humanrf/volume_rendering.py:75-81, 123-141; SURVEY.md A.4 [UPSTREAM-KNOWLEDGE])
# ----------------------------------------------------------------------------------------------

def render_visibility(alphas: torch.Tensor, ray_indices: torch.Tensor, early_stop_eps: float,
                      alpha_thre: float) -> torch.Tensor:
    a = np.ascontiguousarray(alphas.detach().reshape(-1).numpy(), dtype=np.float32)
    ri = np.ascontiguousarray(ray_indices.reshape(-1).numpy(), dtype=np.int64)
    vis = np.empty(a.shape[0], np.uint8)
    _lib().orc_visibility(_p(a), _p(ri), ctypes.c_int64(a.shape[0]), ctypes.c_float(early_stop_eps),
                          ctypes.c_float(alpha_thre), _p(vis))
    return torch.from_numpy(vis.astype(bool))


def _exclusive_segment_cumsum(x: torch.Tensor, ray_indices: torch.Tensor) -> torch.Tensor:
    """Exclusive cumulative sum of x restarting at every new ray (ray_indices sorted)."""
    cs = torch.cumsum(x.double(), 0)
    excl = cs - x.double()
    first = torch.ones_like(ray_indices, dtype=torch.bool)
    first[1:] = ray_indices[1:] != ray_indices[:-1]
    start_idx = torch.cummax(torch.where(first, torch.arange(x.shape[0]), torch.zeros_like(ray_indices)), 0).values
    return (excl - excl[start_idx]).float()


def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices) -> torch.Tensor:
    """w_i = T_i (1 - exp(-sigma_i dt_i)), T_i = exp(-sum_{j<i} sigma_j dt_j); dt = t_end - t_start."""
    sdt = sigmas.reshape(-1) * (t_ends.reshape(-1) - t_starts.reshape(-1))
    T = torch.exp(-_exclusive_segment_cumsum(sdt, ray_indices))
    return T * (1.0 - torch.exp(-sdt))


def accumulate_along_rays(weights, ray_indices, values, n_rays) -> torch.Tensor:
    src = weights.reshape(-1, 1) * values if values is not None else weights.reshape(-1, 1)
    out = torch.zeros(n_rays, src.shape[1], dtype=src.dtype)
    return out.index_add(0, ray_indices, src)


# ----------------------------------------------------------------------------------------------
# prune_samples / render / loss  (humanrf/volume_rendering.py:42-150, humanrf/trainer.py:205-255)
# ----------------------------------------------------------------------------------------------

def prune_samples(m: OracleModel, ray_origins, ray_directions, frame_numbers, sample_distances, ray_indices,
                  jitter: Optional[torch.Tensor], step: float = 4e-4):
    """volume_rendering.py:42-84. `jitter` stands for torch.rand_like(sample_distances) (None: eval).
    Returns (sample_distances (N,1), ray_indices, visibility_mask, sigma) with jitter applied."""
    with torch.no_grad():
        t = sample_distances.reshape(-1, 1)
        if jitter is not None:
            t = t + jitter.reshape(-1, 1) * step
        pos = ray_origins[ray_indices] + t * ray_directions[ray_indices]
        sigma, _, _ = model_density(m, pos, frame_numbers.reshape(-1)[ray_indices])
        alphas = 1.0 - torch.exp(-sigma * step)
        vis = render_visibility(alphas, ray_indices, 1e-4, 1e-4)
    return t, ray_indices, vis, sigma


def render(m: OracleModel, ray_origins, ray_directions, frame_numbers, camera_numbers, sample_distances,
           ray_indices, background_rgb, is_training: bool, step: float = 4e-4):
    """volume_rendering.py:87-150 -> (color (R,3), weights_sum (R,1))."""
    n_rays = ray_origins.shape[0]
    d = ray_directions[ray_indices]
    t = sample_distances.reshape(-1, 1)
    pos = ray_origins[ray_indices] + t * d
    sigma, rgb = model_forward(m, pos, d, frame_numbers.reshape(-1)[ray_indices],
                               camera_numbers.reshape(-1)[ray_indices], is_training)
    w = render_weight_from_density(t, t + step, sigma, ray_indices)
    color = accumulate_along_rays(w, ray_indices, rgb, n_rays)
    acc = accumulate_along_rays(w, ray_indices, None, n_rays)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - acc)
    return color, acc


def training_loss(color, acc, rgba, background_rgb, bce_weight: float = 1e-3):
    """trainer.py:205-247: gt blend with the background, Huber(delta=0.01, mean) + w * mean BCE."""
    gt_mask = rgba[:, 3:4]
    gt_rgb = rgba[:, 0:3] * gt_mask + background_rgb * (1 - gt_mask)
    photometric = torch.nn.functional.huber_loss(color, gt_rgb, reduction="mean", delta=0.01)
    p = torch.clamp(acc, 0, 1)
    bce = -(gt_mask * torch.log(p + 1e-10) + (1 - gt_mask) * torch.log(1 - p + 1e-10))
    return photometric + bce.mean() * bce_weight, photometric


def psnr(color, gt_rgb) -> float:
    """trainer.py:218-223 / actorshq/evaluation/evaluate.py:80-85."""
    mse = torch.square(color - gt_rgb).mean().item()
    return -10.0 * math.log10(mse)
