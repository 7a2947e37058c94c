"""Device-resident batch collection for the training step (trainer.py:138-172 of the reference).

Same control flow and results as the reference-shaped sequence
    next(loader) -> prune_samples -> ... -> merge_input_batches
but organised so that one iteration of the batch-growing loop costs ONE host synchronisation and no tensor
re-packing: the sampler stages, the fused prune march and the scans run back to back on the stream with
device-side counts (the kernels take upper bounds from the host and the true counts from device memory), the three
sizes (R, N0, N1) are read back together, and the survivors are packed straight into step-level ray / sample
buffers at the running offsets, which is what merge_input_batches' concatenation + re-basing would produce.
The reference pays >= 6 synchronisations and ~20 boolean-mask / cat kernels per iteration here
(ray_sampler.cu:256-323, data_loader.py:631-660, volume_rendering.py:83-84, input.py:10-55)."""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .dataset.input_batch import InputBatch

STEP = 4e-4


class StepCollector:
    def __init__(self, model, loader, samples_max: int, rays_initial: int, cap_rays: int = 1 << 18, cap_pre: int = 1 << 22):
        self.model, self.loader = model, loader
        self.samples_max, self.rays_initial = samples_max, rays_initial
        self.dev = model.table_params.device
        self.cap_rays = cap_rays
        self.cap_samples = int(samples_max * 1.1) + samples_max  # one overshooting iteration still fits
        self.cap_pre = cap_pre
        self.cap_r0 = 0
        self.evaluated = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self._alloc_step()
        self._alloc_iter(max(rays_initial, 1 << 15))

    # ------------------------------------------------------------------ buffers
    def _alloc_step(self):
        d, R, N = self.dev, self.cap_rays, self.cap_samples
        f32, i32 = torch.float32, torch.int32
        self.origins = torch.empty(R, 3, dtype=f32, device=d)
        self.dirs = torch.empty(R, 3, dtype=f32, device=d)
        self.rgba = torch.empty(R, 4, dtype=f32, device=d)
        self.frames = torch.empty(R, dtype=i32, device=d)
        self.cams = torch.empty(R, dtype=i32, device=d)
        self.minmax = torch.empty(R, 2, dtype=f32, device=d)
        self.count = torch.empty(R, dtype=i32, device=d)
        self.ridx = torch.empty(R, dtype=torch.int64, device=d)
        self.t = torch.empty(N, dtype=f32, device=d)
        self.ray = torch.empty(N, dtype=torch.int64, device=d)

    def _grow_rays(self, new_cap: int, keep: int):
        """Enlarge the per-ray step buffers, preserving the first `keep` rows (rays of earlier iterations)."""
        old = (self.origins, self.dirs, self.rgba, self.frames, self.cams, self.minmax, self.count, self.ridx)
        self.cap_rays = new_cap
        t, r = self.t, self.ray
        self._alloc_step()
        self.t, self.ray = t, r
        for dst, src in zip((self.origins, self.dirs, self.rgba, self.frames, self.cams, self.minmax, self.count, self.ridx), old):
            dst[:keep].copy_(src[:keep])

    def _alloc_iter(self, r0: int):
        d = self.dev
        f32, i32 = torch.float32, torch.int32
        self.cap_r0 = r0
        self.dirs_all = torch.empty(r0, 3, dtype=f32, device=d)
        self.mm_all = torch.empty(r0, 2, dtype=f32, device=d)
        self.mask = torch.empty(r0, dtype=torch.uint8, device=d)
        self.count_all = torch.empty(r0, dtype=i32, device=d)
        self.slot = torch.empty(r0 + 1, dtype=i32, device=d)
        self.kept = torch.empty(r0, dtype=i32, device=d)
        self.offsets = torch.empty(r0 + 1, dtype=i32, device=d)
        self.ray_cnt = torch.empty(r0, dtype=i32, device=d)
        self.ray_eval = torch.empty(r0, dtype=i32, device=d)
        self.out_off = torch.empty(r0 + 1, dtype=i32, device=d)
        self.scan_ws = torch.empty(2 * ((r0 + 4095) // 4096) + 1, dtype=i32, device=d)
        self.sizes = torch.empty(3, dtype=i32, device=d)
        self.idx = torch.empty(r0, dtype=torch.int64, device=d)
        self.order = torch.empty(r0, dtype=i32, device=d)
        self.order_ws = torch.empty(2 * max(self.model.num_segments, min(self.model.num_frames, 1024)), dtype=i32, device=d)
        self._alloc_pre(self.cap_pre)

    def _alloc_pre(self, n: int):
        d = self.dev
        self.cap_pre = n
        self.t0 = torch.empty(n, dtype=torch.float32, device=d)
        self.ray0 = torch.empty(n, dtype=torch.int32, device=d)
        self.t_stage = torch.empty(n, dtype=torch.float32, device=d)

    def _scan(self, x, is_u8, n, out):
        ws = self.scan_ws if n > 8192 else None
        check(_lib.lib().hrf_scan_exclusive(ptr(x), 1 if is_u8 else 0, n, ptr(out), ptr(ws), stream_ptr()))

    # ------------------------------------------------------------------ one iteration of the batch-growing loop
    def _iteration(self, r0: int, ray_base: int, samp_base: int) -> Tuple[int, int, int]:
        L, ld, m, st = _lib.lib(), self.loader, self.model, stream_ptr()
        if r0 > self.cap_r0:
            self._alloc_iter(int(r0 * 1.25))
        if ray_base + r0 > self.cap_rays:
            self._grow_rays(int((ray_base + r0) * 1.5), ray_base)
        width, height = ld.resolution
        P = width * height
        idx = ld.draw_ray_indices(r0, out=self.idx)                                # data_loader.py:540-546
        occ = 1 if ld.occupancy else 0
        G = int(ld.occupancy_grid_resolution)
        land = ld.landscape_mode_cuda.view(torch.uint8)
        tex = ld.grid_texture_objects_cuda if occ else None
        with ops._span("sampler_kernels", r0):
            check(L.hrf_sampler_rays(ptr(ld.inverse_krs_cuda), ptr(ld.camera_origins_cuda), ptr(land), ptr(idx), ptr(tex),
                                     ptr(ld.aabb), None, r0, G, width, height, STEP, occ, ptr(self.dirs_all),
                                     ptr(self.mm_all), ptr(self.mask), ptr(self.count_all), st))
            self._scan(self.mask, True, r0, self.slot)
            rb = ray_base
            check(L.hrf_sampler_compact_rays(ptr(idx), ptr(self.mask), ptr(self.slot), ptr(self.dirs_all), ptr(self.mm_all),
                                             ptr(self.count_all), ptr(ld.pixel_colors), ptr(ld.camera_origins_cuda),
                                             ptr(ld.frame_numbers_cuda), ptr(ld.camera_numbers_cuda), r0, P,
                                             ptr(self.origins[rb:]), ptr(self.dirs[rb:]), ptr(self.rgba[rb:]),
                                             ptr(self.frames[rb:]), ptr(self.cams[rb:]), ptr(self.minmax[rb:]),
                                             ptr(self.count[rb:]), ptr(self.ridx[rb:]), st))
            n_dev = self.slot[r0:]
            check(L.hrf_sampler_samples(ptr(self.ridx[rb:]), ptr(tex), ptr(self.origins[rb:]), ptr(self.dirs[rb:]),
                                        ptr(self.minmax[rb:]), ptr(self.count[rb:]), None, r0, ptr(n_dev), P, G, STEP, occ,
                                        ptr(self.kept), None, None, self.cap_pre, st))
            self._scan(self.kept, False, r0, self.offsets)
            check(L.hrf_sampler_samples(ptr(self.ridx[rb:]), ptr(tex), ptr(self.origins[rb:]), ptr(self.dirs[rb:]),
                                        ptr(self.minmax[rb:]), ptr(self.count[rb:]), ptr(self.offsets), r0, ptr(n_dev), P, G,
                                        STEP, occ, None, ptr(self.t0), ptr(self.ray0), self.cap_pre, st))
        jitter = torch.rand(self.cap_pre, dtype=torch.float32, device=self.dev)  # volume_rendering.py:63-64
        m._refresh_half()
        sw1, sw2 = m._sigma_w()
        order = None
        if m.num_frames > 1:  # schedule only: rays by frame, one eighth per XCD
            order = ops.ray_segment_order(self.frames[rb:rb + r0], m, n_dev, out=self.order, workspace=self.order_ws)
        with ops._span("prune_march", 1):
            check(L.hrf_prune_march(ptr(self.origins[rb:]), ptr(self.dirs[rb:]), ptr(self.frames[rb:]), ptr(self.offsets),
                                    ptr(self.t0), ptr(jitter), STEP, 1e-4, 1e-4, ptr(m.frame_numbers_to_segment_numbers),
                                    ptr(m.frame_numbers_to_normalized_local_frame_numbers), ptr(m._tables_h),
                                    ptr(m.vectors), ptr(m._seg_meta), m.num_segments, m.vec_res, ptr(sw1), ptr(sw2),
                                    float(m.density_scale), r0, ptr(n_dev), self.cap_pre, ptr(self.t_stage), None,
                                    ptr(self.ray_cnt), ptr(self.ray_eval), ptr(order), st))
        self._scan(self.ray_cnt, False, r0, self.out_off)
        torch.stack([self.slot[r0], self.offsets[r0], self.out_off[r0]], out=self.sizes)
        R, n0, n1 = (int(v) for v in self.sizes.cpu())                  # the single host sync of the iteration
        if n0 > self.cap_pre:                                            # rare: grow and redo (kernels guard the bound)
            self._alloc_pre(int(n0 * 1.5))
            return self._iteration(r0, ray_base, samp_base)
        if samp_base + n1 > self.cap_samples:
            raise RuntimeError("StepCollector: sample capacity exceeded")
        self.evaluated += self.ray_eval[:r0].sum()
        check(L.hrf_pack_runs(ptr(self.offsets), ptr(self.ray_cnt), ptr(self.out_off), ptr(self.t_stage), R, None, ray_base,
                              ptr(self.t[samp_base:]), ptr(self.ray[samp_base:]), st))
        return R, n0, n1

    # ------------------------------------------------------------------ trainer.py:138-172
    def collect(self):
        """-> (InputBatch of views into the step buffers, rays drawn, pre-prune samples)."""
        r0 = self.rays_initial
        total_rays = total_samples = 0
        ray_base = samp_base = n_pre = 0
        while True:
            R, n0, n1 = self._iteration(r0, ray_base, samp_base)
            ray_base += R
            samp_base += n1
            n_pre += n0
            total_rays += r0
            total_samples += n1
            if total_samples < 0.9 * self.samples_max:
                avg = total_samples / total_rays
                assert avg > 0, "There is probably a problem with the predicted geometry."
                r0 = int((self.samples_max - total_samples) / avg)
            else:
                break
        n_rays, n_samples = ray_base, samp_base
        max_num = int(self.samples_max * 1.1)
        if n_samples > max_num:                                          # humanrf/input.py:33-47
            cutoff = int(self.ray[max_num].item())
            n_samples = int(torch.searchsorted(self.ray[:n_samples], cutoff).item())
            n_rays = cutoff
        ib = InputBatch(ray_origins=self.origins[:n_rays], ray_directions=self.dirs[:n_rays], minmaxes=self.minmax[:n_rays],
                        rgba=self.rgba[:n_rays], frame_numbers=self.frames[:n_rays].view(-1, 1),
                        camera_numbers=self.cams[:n_rays].view(-1, 1), sample_distances=self.t[:n_samples].view(-1, 1),
                        ray_indices=self.ray[:n_samples], width=self.loader.resolution[0], height=self.loader.resolution[1])
        return ib, total_rays, n_pre
