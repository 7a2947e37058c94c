"""Inference form of the hot path: full-image rendering for validation / test (humanrf/trainer.py:257-370, 517-526).

Same kernels as training, driven the way Trainer.validate drives them: the loader yields consecutive pixel ranges of one
image (data_loader.py:576-624), every batch goes through prune_samples(is_training=False) (no jitter) and
render(background 0, is_training=False) (zero camera embedding, humanrf.py:196-204), the partial outputs are merged
(merge_input_batches / RenderOutput.merge_render_outputs) and scattered into the image through `ray_masks`
(combine_rays_to_image). PSNR follows trainer.py:218-223 as evaluate_one_image applies it (trainer.py:372-389): mean
squared error over the rendered, i.e. ray-masked, rays against the ground truth blended onto the background.
Image files, LPIPS and SSIM (trainer.py:404-416) are out of scope."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Tuple

import torch

from .dataset.input_batch import InputBatch
from .input import merge_input_batches
from .volume_rendering import RenderOutput, prune_samples, render


@torch.no_grad()
def combine_rays_to_image(full_image_batch: InputBatch, full_render_output: RenderOutput, background_rgb) -> torch.Tensor:
    """trainer.py:517-526: (1, H, W, 3) image, rays without samples in the occupancy grid keep the background."""
    image_width, image_height = full_image_batch.width, full_image_batch.height
    color = full_render_output.color
    full_pred_rgb = torch.full((image_width * image_height, 3), float(background_rgb), dtype=torch.float, device=color.device)
    full_pred_rgb[full_image_batch.ray_masks.squeeze(1)] = color
    return full_pred_rgb.view(1, image_height, image_width, 3)


@torch.no_grad()
def psnr_of_rendered_rays(full_render_output: RenderOutput, rgba: torch.Tensor, background_rgb) -> float:
    """trainer.py:372-389 + 218-223: gt = rgb*mask + bg*(1-mask) over the rendered rays, psnr = -10 log10(mse)."""
    gt_rgb, gt_mask = rgba[..., 0:3], rgba[..., 3:4]
    gt_rgb = gt_rgb * gt_mask + background_rgb * (1 - gt_mask)
    mse = torch.square(full_render_output.color - gt_rgb).mean().item()
    return -10.0 * math.log10(max(mse, 1e-20))


@torch.no_grad()
def render_image(model, batches: Iterable[InputBatch], background_rgb: float = 0.0) -> Tuple[InputBatch, RenderOutput]:
    """The loop body of Trainer.validate for one image (trainer.py:283-315): -> (merged image batch holding ray_masks /
    rgba / width / height, merged RenderOutput)."""
    partial_batches: List[InputBatch] = []
    partial_outputs: List[RenderOutput] = []
    for current_batch in batches:
        partial_batches.append(InputBatch(ray_masks=current_batch.ray_masks, rgba=current_batch.rgba,
                                          width=current_batch.width, height=current_batch.height))
        if current_batch.num_rays == 0:   # nothing of this pixel range hits the occupancy grid
            dev = current_batch.ray_masks.device
            partial_outputs.append(RenderOutput(color=torch.zeros(0, 3, device=dev), weights_sum=torch.zeros(0, 1, device=dev)))
            continue
        prune_samples(current_batch, model, False)
        partial_outputs.append(render(current_batch, model, background_rgb, False))
    full = InputBatch(ray_masks=torch.cat([b.ray_masks for b in partial_batches], 0),
                      rgba=torch.cat([b.rgba for b in partial_batches], 0),
                      width=partial_batches[0].width, height=partial_batches[0].height)
    return full, RenderOutput.merge_render_outputs(partial_outputs)


@torch.no_grad()
def validate(model, loader, camera_frame_pairs: Sequence[Tuple[int, int]], rays_batch_size: int = 8192,
             background_rgb: float = 0.0, return_images: bool = False, world_size: int = 1, rank: int = 0,
             group=None) -> Dict[str, object]:
    """Trainer.validate's metric loop (trainer.py:257-370) over `camera_frame_pairs`: per-image PSNR and their mean.
    `loader.validation_batches(camera, frame, batch)` must yield the image's pixel ranges in order.
    world_size > 1 (the reference validates on its one GPU): the images are dealt out to the ranks round-robin --
    rank r renders pairs r, r + N, ... on its replica -- and the per-image PSNRs are gathered, so every rank returns the full list in
    the order of `camera_frame_pairs`. COLLECTIVE: every rank calls it with the same pairs. `images` holds this rank's share only."""
    pairs = list(camera_frame_pairs)
    mine = list(range(rank, len(pairs), max(world_size, 1)))
    psnrs, images = [], []
    was_training = model.training
    model.eval()
    try:
        for i in mine:
            cam, frame = pairs[i]
            full_batch, full_out = render_image(model, loader.validation_batches(cam, frame, rays_batch_size), background_rgb)
            psnrs.append(psnr_of_rendered_rays(full_out, full_batch.rgba, background_rgb))
            if return_images:
                images.append(combine_rays_to_image(full_batch, full_out, background_rgb))
    finally:
        model.train(was_training)
    if world_size > 1:
        import torch.distributed as dist
        dev = next(model.parameters()).device
        if str(dist.get_backend(group)) == "gloo":
            dev = torch.device("cpu")
        buf = torch.zeros(len(pairs), dtype=torch.float64, device=dev)
        if mine:
            buf[torch.tensor(mine, device=dev)] = torch.tensor(psnrs, dtype=torch.float64, device=dev)
        dist.all_reduce(buf, group=group)           # every image was rendered by exactly one rank: the sum is the gather
        psnrs = buf.tolist()
    out: Dict[str, object] = {"psnr": psnrs, "psnr_mean": sum(psnrs) / max(len(psnrs), 1), "images_rendered_here": len(mine)}
    if return_images:
        out["images"] = images
    return out


__all__ = ["combine_rays_to_image", "psnr_of_rendered_rays", "render_image", "validate", "merge_input_batches"]
