"""Drop-in for actorshq.dataset.ray_sampler_native (ray_sampler.cu:196-333): the four entry points
get_{rays,samples}_{aabb,occupancy}_minmax with the reference's 15-argument signature and 9 outputs.

Differences from the reference, all on the host side of the C ABI:
  * rgba / light_mask pools may live in HBM. CPU pools, which is what the reference's
    data loader passes (data_loader.py:258-271), are accepted and gathered on the CPU exactly like
    ray_sampler.cu:256,262 do.
  * one host synchronisation per call (the two sizes R and N0 are read back together) instead of the
    reference's >= 4 (ray_sampler.cu:256,258-262,286,322).
"""
from __future__ import annotations

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..ops import scan_exclusive


def _check(t, name, dtypes, cuda=True):
    """Contiguity / device as actorshq/toolbox/native/utils.cuh:5-19; element type as the typed
    packed_accessor<T> of ray_sampler.cu:214-231 would (it throws on a mismatch)."""
    if not t.is_contiguous():
        raise RuntimeError(f"Tensor not contiguous: {name}")
    if cuda is not None and cuda != t.is_cuda:
        raise RuntimeError(f"Tensor is not on the expected device: {name}")
    if t.dtype not in dtypes:
        raise RuntimeError(f"expected scalar type {' or '.join(str(d) for d in dtypes)} but found {t.dtype}: {name}")


def _get_data(occupancy: bool, get_samples: bool, rgba, light_mask, frame_numbers, camera_numbers,
              grid_texture_objects, landscape_modes, all_ray_indices, inverse_krs, camera_origins, aabb,
              grid_resolution, image_width, image_height, raymarching_step_size, filter_light_bloom):
    L = _lib.lib()
    i32, i64, f32, u8b = (torch.int32,), (torch.int64,), (torch.float32,), (torch.bool, torch.uint8)
    for t, nm, dt in ((frame_numbers, "frame_numbers", i32), (camera_numbers, "camera_numbers", i32),
                      (landscape_modes, "landscape_modes", u8b), (all_ray_indices, "all_ray_indices", i64),
                      (inverse_krs, "inverse_krs", f32), (camera_origins, "camera_origins", f32), (aabb, "aabb", f32)):
        _check(t, nm, dt)
    if occupancy:
        _check(grid_texture_objects, "grid_texture_objects", i64)
    _check(rgba, "rgba", (torch.uint8,), cuda=None)          # the pool may live on either side (see module docstring)
    _check(light_mask, "light_mask", u8b, cuda=None)
    dev = aabb.device
    stream = stream_ptr()
    R0 = all_ray_indices.shape[0]
    P = int(image_width) * int(image_height)
    step = float(raymarching_step_size)
    land_u8 = landscape_modes.view(torch.uint8) if landscape_modes.dtype == torch.bool else landscape_modes

    lm_dev = None
    if filter_light_bloom:
        lm = light_mask.reshape(-1)
        if lm.is_cuda:
            lm_dev = lm.view(torch.uint8) if lm.dtype == torch.bool else lm
        else:  # reference path: CPU gather + H2D (ray_sampler.cu:256); indices are needed on the host anyway
            lm_dev = None

    dirs_all = torch.empty(R0, 3, dtype=torch.float32, device=dev)
    mm_all = torch.empty(R0, 2, dtype=torch.float32, device=dev)
    mask = torch.empty(R0, dtype=torch.uint8, device=dev)
    count_all = torch.empty(R0, dtype=torch.int32, device=dev)
    check(L.hrf_sampler_rays(ptr(inverse_krs), ptr(camera_origins), ptr(land_u8), ptr(all_ray_indices),
                             ptr(grid_texture_objects) if occupancy else None, ptr(aabb), ptr(lm_dev), R0,
                             int(grid_resolution), int(image_width), int(image_height), step, 1 if occupancy else 0,
                             ptr(dirs_all), ptr(mm_all), ptr(mask), ptr(count_all),
                             ptr(torch.empty(R0 + 1, dtype=torch.int32, device=dev)) if occupancy else None, stream))
    if filter_light_bloom and lm_dev is None:
        sel = light_mask.reshape(-1)[all_ray_indices.cpu()].to(dev)
        mask = (mask.bool() & ~sel).to(torch.uint8)
        count_all = torch.where(mask.bool(), count_all, torch.zeros_like(count_all))
    slot = scan_exclusive(mask)

    rgba_dev = rgba.reshape(-1, 4) if rgba.is_cuda else None
    org = torch.empty(R0, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(R0, 3, dtype=torch.float32, device=dev)
    srgba = torch.empty(R0, 4, dtype=torch.float32, device=dev)
    frames = torch.empty(R0, dtype=torch.int32, device=dev)
    cams = torch.empty(R0, dtype=torch.int32, device=dev)
    mm = torch.empty(R0, 2, dtype=torch.float32, device=dev)
    cnt = torch.empty(R0, dtype=torch.int32, device=dev)
    ridx = torch.empty(R0, dtype=torch.int64, device=dev)
    check(L.hrf_sampler_compact_rays(ptr(all_ray_indices), ptr(mask), ptr(slot), ptr(dirs_all), ptr(mm_all),
                                     ptr(count_all), ptr(rgba_dev), ptr(camera_origins), ptr(frame_numbers),
                                     ptr(camera_numbers), R0, P, ptr(org), ptr(dirs),
                                     ptr(srgba) if rgba_dev is not None else None, ptr(frames), ptr(cams), ptr(mm),
                                     ptr(cnt), ptr(ridx), None, None, stream))
    ray_mask = mask.view(torch.bool)
    if not get_samples:
        R = int(slot[R0].item())
        out_rgba = srgba[:R] if rgba_dev is not None else (rgba.reshape(-1, 4)[ridx[:R].cpu()] / 255.0).to(dev)
        return [org[:R], dirs[:R], out_rgba, frames[:R], cams[:R], mm[:R], ray_mask,
                torch.empty(0, dtype=torch.float32, device=dev), torch.empty(0, dtype=torch.int32, device=dev)]

    # sample stage. R is not known on the host yet: pass 1 runs over all R0 slots and skips those beyond the
    # device-side count.
    n_rays_dev = slot[R0:R0 + 1]  # device-side R: the kernels skip slots beyond it
    kept = torch.empty(R0, dtype=torch.int32, device=dev)
    check(L.hrf_sampler_samples(ptr(ridx), ptr(grid_texture_objects) if occupancy else None, ptr(org), ptr(dirs),
                                ptr(mm), ptr(cnt), None, R0, ptr(n_rays_dev), P, int(grid_resolution), step,
                                1 if occupancy else 0, ptr(kept), None, None, 0, stream))
    offsets = scan_exclusive(kept)
    sizes = torch.stack([slot[R0], offsets[R0]]).cpu()  # the single host sync of the call
    R, N = int(sizes[0]), int(sizes[1])
    t = torch.empty(N, dtype=torch.float32, device=dev)
    ray = torch.empty(N, dtype=torch.int32, device=dev)
    if N > 0:
        check(L.hrf_sampler_samples(ptr(ridx), ptr(grid_texture_objects) if occupancy else None, ptr(org), ptr(dirs),
                                    ptr(mm), ptr(cnt), ptr(offsets), R, None, P, int(grid_resolution), step,
                                    1 if occupancy else 0, None, ptr(t), ptr(ray), N, stream))
    out_rgba = srgba[:R] if rgba_dev is not None else (rgba.reshape(-1, 4)[ridx[:R].cpu()] / 255.0).to(dev)
    return [org[:R], dirs[:R], out_rgba, frames[:R], cams[:R], mm[:R], ray_mask, t, ray]


def get_rays_aabb_minmax(*args):
    return _get_data(False, False, *args)


def get_rays_occupancy_minmax(*args):
    return _get_data(True, False, *args)


def get_samples_aabb_minmax(*args):
    return _get_data(False, True, *args)


def get_samples_occupancy_minmax(*args):
    return _get_data(True, True, *args)
