"""The three nerfacc 0.3.1 functions humanrf/volume_rendering.py calls (:75-81, :123-141), on the gfx950 kernels:
`import humanrf_amd.compat.nerfacc as nerfacc` keeps code written against them working. `ray_indices` must be sorted
(contiguous run per ray), which is what the sampler emits and what nerfacc's packed form assumes as well.
The training step itself does not go through these: it uses the fused composite kernels."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib, ops
from .._lib import check, ptr, stream_ptr


def _ray_start(ray_indices: torch.Tensor, n_rays: int) -> torch.Tensor:
    return ops.ray_offsets(ray_indices.contiguous(), n_rays)


@torch.no_grad()
def render_visibility(alphas: torch.Tensor, *, ray_indices: torch.Tensor, early_stop_eps: float = 1e-4,
                      alpha_thre: float = 0.0, n_rays: Optional[int] = None) -> torch.Tensor:
    """visible_i = (T_i >= early_stop_eps) & (alpha_i >= alpha_thre), T_i = prod_{j<i} (1 - alpha_j)  -> bool (N,)."""
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    a = alphas.reshape(-1).float().contiguous()
    vis, _ = ops.visibility(a, None, _ray_start(ray_indices, n_rays), n_rays, early_stop_eps, alpha_thre)
    return vis.bool()


class _Weights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, t_starts, t_ends, ray_start, n_rays):
        s = sigmas.reshape(-1).float().contiguous()
        t0, t1 = t_starts.reshape(-1).float().contiguous(), t_ends.reshape(-1).float().contiguous()
        w = torch.empty_like(s)
        check(_lib.lib().hrf_weights_fwd(ptr(s), ptr(t0), ptr(t1), ptr(ray_start), n_rays, ptr(w), stream_ptr()))
        ctx.save_for_backward(s, t0, t1, ray_start)
        ctx.n_rays, ctx.shape = n_rays, sigmas.shape
        return w.view(-1, 1)

    @staticmethod
    def backward(ctx, d_w):
        s, t0, t1, ray_start = ctx.saved_tensors
        g = d_w.reshape(-1).float().contiguous()
        d_s = torch.zeros_like(s)
        check(_lib.lib().hrf_weights_bwd(ptr(s), ptr(t0), ptr(t1), ptr(ray_start), ptr(g), ctx.n_rays, ptr(d_s), stream_ptr()))
        return d_s.view(ctx.shape), None, None, None, None


def render_weight_from_density(t_starts: torch.Tensor, t_ends: torch.Tensor, sigmas: torch.Tensor, *,
                               ray_indices: torch.Tensor, n_rays: Optional[int] = None) -> torch.Tensor:
    """w_i = T_i (1 - exp(-sigma_i (t_end_i - t_start_i))), differentiable in sigmas -> (N, 1)."""
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    return _Weights.apply(sigmas, t_starts, t_ends, _ray_start(ray_indices, n_rays), n_rays)


class _Accumulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, values, ray_indices, ray_start, n_rays):
        w = weights.reshape(-1).float().contiguous()
        v = values.float().contiguous() if values is not None else None
        D = v.shape[1] if v is not None else 1
        out = torch.empty(n_rays, D, dtype=torch.float32, device=w.device)
        check(_lib.lib().hrf_accumulate_fwd(ptr(w), ptr(v), D, ptr(ray_start), n_rays, ptr(out), stream_ptr()))
        ctx.save_for_backward(w, v, ray_indices)
        ctx.D, ctx.wshape, ctx.vdtype = D, weights.shape, values.dtype if values is not None else None
        return out

    @staticmethod
    def backward(ctx, d_out):
        w, v, ray_indices = ctx.saved_tensors
        g = d_out.float().contiguous()
        d_w = torch.empty_like(w)
        d_v = torch.empty_like(v) if v is not None else None
        check(_lib.lib().hrf_accumulate_bwd(ptr(w), ptr(v), ctx.D, ptr(ray_indices), ptr(g), w.numel(), ptr(d_w), ptr(d_v),
                                            stream_ptr()))
        return d_w.view(ctx.wshape), (d_v.to(ctx.vdtype) if d_v is not None else None), None, None, None


def accumulate_along_rays(weights: torch.Tensor, ray_indices: torch.Tensor, values: Optional[torch.Tensor] = None,
                          n_rays: Optional[int] = None) -> torch.Tensor:
    """out[r] = sum_{i in ray r} weights_i * values_i  ((n_rays, D); values None -> (n_rays, 1) sum of weights)."""
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    ri = ray_indices.contiguous()
    return _Accumulate.apply(weights, values, ri, _ray_start(ri, n_rays), n_rays)
