"""prune_samples / render / RenderOutput with the operator surface of humanrf/volume_rendering.py:14-150,
running on the gfx950 kernels: the nerfacc calls (render_visibility, render_weight_from_density,
accumulate_along_rays) and the torch glue around them are replaced by wavefront-per-ray kernels that rely on
`ray_indices` being sorted (one contiguous run per ray, as the sampler emits them)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch

from . import ops
from .dataset.input_batch import InputBatch
from .scene_representation.humanrf import HumanRF
from .scene_representation.query_io import QueryInput


@dataclass
class RenderOutput:
    """Per-ray outputs (volume_rendering.py:14-39)."""

    color: torch.Tensor = None        # (#rays, 3) float
    weights_sum: torch.Tensor = None  # (#rays, 1) float

    @classmethod
    @torch.no_grad()
    def merge_render_outputs(cls, render_outputs: List["RenderOutput"]) -> "RenderOutput":
        final = RenderOutput()
        for key, val in vars(render_outputs[0]).items():
            if val is None:
                setval = None
            elif isinstance(val, torch.Tensor):
                setval = torch.cat([getattr(r, key) for r in render_outputs], dim=0)
            else:
                raise RuntimeError("Unknown data type in the input_batches!")
            setattr(final, key, setval)
        return final


class _CompositeFn(torch.autograd.Function):
    """render_weight_from_density + accumulate_along_rays (rgb and weights) + background blend
    (volume_rendering.py:123-145) as one kernel forward and one backward."""

    @staticmethod
    def forward(ctx, sigma, rgb, t, ray_start, background, num_rays, step):
        rgb_h = rgb.half().contiguous()
        sigma = sigma.float().contiguous()
        color, acc = ops.composite_fwd(sigma, rgb_h, t, ray_start, background, num_rays, step)
        ctx.save_for_backward(sigma, rgb_h, t, ray_start, background)
        ctx.num_rays, ctx.step = num_rays, step
        return color, acc

    @staticmethod
    def backward(ctx, d_color, d_acc):
        sigma, rgb_h, t, ray_start, background = ctx.saved_tensors
        d_color = d_color.float().contiguous()
        d_acc = d_acc.float().contiguous() if d_acc is not None else None
        d_sigma, d_rgb = ops.composite_bwd(sigma, rgb_h, t, ray_start, background, d_color, d_acc, ctx.num_rays, ctx.step)
        return d_sigma, d_rgb, None, None, None, None, None


def _background_tensor(background_rgb, num_rays: int, device) -> torch.Tensor:
    if background_rgb is None:
        return None
    if not torch.is_tensor(background_rgb):
        background_rgb = torch.full((num_rays, 3), float(background_rgb), dtype=torch.float32, device=device)
    bg = background_rgb.to(device=device, dtype=torch.float32)
    if bg.shape != (num_rays, 3):
        bg = bg.expand(num_rays, 3)
    return bg.contiguous()


FUSED_PRUNE = True  # False: the unfused kernel sequence (encode -> sigma_net -> visibility); same results


@torch.no_grad()
def prune_samples(input_batch: InputBatch, scene_representation, is_training: bool,
                  render_step_size: float = 4e-4) -> None:
    """In-place pruning of samples whose weight is negligible (volume_rendering.py:42-84):
    jitter (training), density of every sample, alpha = 1 - exp(-sigma*step),
    visible = (T >= 1e-4) & (alpha >= 1e-4), boolean-mask compaction of sample_distances / ray_indices.
    With a HumanRF model the whole body is one fused march kernel with per-ray early termination."""
    ib = input_batch
    n = ib.num_samples
    if n == 0:
        return
    t = ib.sample_distances.reshape(-1).contiguous()
    ray_idx = ib.ray_indices.contiguous()
    jitter = torch.rand_like(t) if is_training else None  # volume_rendering.py:63-64
    ray_start = ops.ray_offsets(ray_idx, ib.num_rays)
    if isinstance(scene_representation, HumanRF) and FUSED_PRUNE:
        m = scene_representation
        m._refresh_half()
        t_stage, _, ray_cnt, ray_eval = ops.prune_march(
            ib.ray_origins.contiguous(), ib.ray_directions.contiguous(), ib.frame_numbers.reshape(-1).contiguous(),
            ray_start, t, jitter, m, 1e-4, 1e-4, render_step_size, want_evaluated=True)
        off = torch.zeros(ib.num_rays + 1, dtype=torch.int32, device=t.device)
        torch.cumsum(ray_cnt, 0, out=off[1:])
        n_keep = int(off[-1].item())
        new_t, new_ray = ops.pack_runs(ray_start, ray_cnt, off, t_stage, n_keep)
        ib.sample_distances = new_t.view(-1, 1)
        ib.ray_indices = new_ray
        ib._num_evaluated = ray_eval  # device tensor: samples actually encoded per ray (statistics only)
        return
    if isinstance(scene_representation, HumanRF):
        m = scene_representation
        xyzt, seg = ops.query_prep(ib.ray_origins.contiguous(), ib.ray_directions.contiguous(),
                                   ib.frame_numbers.reshape(-1).contiguous(), ray_idx, t, jitter,
                                   m.frame_numbers_to_segment_numbers, m.frame_numbers_to_normalized_local_frame_numbers,
                                   render_step_size)
        sigma, _ = m.density_from_xyzt(xyzt, seg)
    else:  # any object with the reference's density(QueryInput) method
        if jitter is not None:
            t = t + jitter * render_step_size
        qi = QueryInput(is_training=is_training,
                        positions=ib.ray_origins[ray_idx] + t.unsqueeze(-1) * ib.ray_directions[ray_idx],
                        frame_numbers=ib.frame_numbers[ray_idx], unique_frame_numbers=ib.unique_frame_numbers)
        sigma = scene_representation.density(qi).density.reshape(-1).float().contiguous()
    vis, _ = ops.visibility(None, sigma, ray_start, ib.num_rays, 1e-4, 1e-4, render_step_size)  # alpha in-kernel
    slot = ops.scan_exclusive(vis)
    n_keep = int(slot[n].item())
    new_t, new_ray = ops.compact_samples(vis, slot, t, ray_idx, n_keep)
    ib.sample_distances = new_t.view(-1, 1)
    ib.ray_indices = new_ray


def render(input_batch: InputBatch, scene_representation, background_rgb, is_training: bool,
           render_step_size: float = 4e-4) -> RenderOutput:
    """Weights per sample and per-ray accumulation (volume_rendering.py:87-150)."""
    ib = input_batch
    dev = ib.ray_origins.device
    t = ib.sample_distances.reshape(-1).contiguous()
    ray_idx = ib.ray_indices.contiguous()
    if isinstance(scene_representation, HumanRF):
        m = scene_representation
        xyzt, seg = ops.query_prep(ib.ray_origins.contiguous(), ib.ray_directions.contiguous(),
                                   ib.frame_numbers.reshape(-1).contiguous(), ray_idx, t, None,
                                   m.frame_numbers_to_segment_numbers, m.frame_numbers_to_normalized_local_frame_numbers,
                                   render_step_size)
        cams = ib.camera_numbers.reshape(-1).contiguous() if ib.camera_numbers is not None else None
        sigma, rgb, _ = m.field(xyzt, seg, ib.ray_directions.contiguous(), ray_idx, cams, is_training)
    else:
        d = ib.ray_directions[ray_idx]
        qi = QueryInput(is_training=is_training, positions=ib.ray_origins[ray_idx] + t.unsqueeze(-1) * d, directions=d,
                        frame_numbers=ib.frame_numbers[ray_idx], unique_frame_numbers=ib.unique_frame_numbers,
                        camera_numbers=ib.camera_numbers[ray_idx])
        q = scene_representation(qi)
        sigma, rgb = q.density.reshape(-1), q.radiance
    ray_start = ops.ray_offsets(ray_idx, ib.num_rays)
    bg = _background_tensor(background_rgb, ib.num_rays, dev)
    color, acc = _CompositeFn.apply(sigma, rgb, t, ray_start, bg, ib.num_rays, render_step_size)
    return RenderOutput(color=color, weights_sum=acc)
