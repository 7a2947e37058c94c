"""Device-resident batch collection for the training step (trainer.py:138-172 of the reference).

Same control flow and results as the reference-shaped sequence
    next(loader) -> prune_samples -> ... -> merge_input_batches
but organised so that one iteration of the batch-growing loop costs ONE host synchronisation and no tensor
re-packing: the sampler stages, the fused prune march and the scans run back to back on the stream with
device-side counts (the kernels take upper bounds from the host and the true counts from device memory), the
sizes are read back together, and the survivors are packed straight into step-level ray / sample buffers at the
running offsets, which is what merge_input_batches' concatenation + re-basing would produce.
The reference pays >= 6 synchronisations and ~20 boolean-mask / cat kernels per iteration here
(ray_sampler.cu:256-323, data_loader.py:631-660, volume_rendering.py:83-84, input.py:10-55).

Pipelining (`pipelined=True`): the sampler stages do not depend on the model, only the prune march does. While
step n is collected, the sampler stages of step n+1 run on a second HIP stream over a predicted number of drawn
rays (rays_initial + 1.15 x what step n-1 needed); step n+1's batch-growing iterations then consume PREFIXES of that
set -- sampler outputs are per drawn ray and compacted in draw order, so a prefix of the drawn rays is a prefix of
every derived array -- and only march. If the prediction falls short, the remaining iterations run the classic,
un-overlapped way. The draws are i.i.d. uniform either way (data_loader.py:540-546)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .dataset.input_batch import InputBatch

STEP = 4e-4


class _RaySet:
    """Sampler-stage outputs for one set of drawn rays: per drawn ray (mask, slot, ...), per compacted ray (origins,
    ...; also the per-ray arrays of the training batch built from it) and the staged samples (t0)."""

    def __init__(self, dev, cap_draw: int, cap_pre: int):
        self.dev = dev
        self.n_drawn = 0                      # drawn rays the sampler stages have been run for (prefetch)
        self.ready: Optional[torch.cuda.Event] = None
        self._alloc_draw(cap_draw)
        self._alloc_compact(cap_draw)
        self._alloc_pre(cap_pre)

    def _alloc_draw(self, n: int):
        d, f32, i32 = self.dev, torch.float32, torch.int32
        self.cap_draw = n
        self.idx = torch.empty(n, dtype=torch.int64, device=d)
        self.dirs_all = torch.empty(n, 3, dtype=f32, device=d)
        self.mm_all = torch.empty(n, 2, dtype=f32, device=d)
        self.mask = torch.empty(n, dtype=torch.uint8, device=d)
        self.count_all = torch.empty(n, dtype=i32, device=d)
        self.slot = torch.empty(n + 1, dtype=i32, device=d)
        self.cand_all = torch.empty(n + 1, dtype=i32, device=d)   # exclusive scan of count_all (candidates per drawn ray)
        self.scan_ws = torch.zeros(2 * ((n + 4095) // 4096) + 8, dtype=i32, device=d)
        self.pre_ws = torch.empty(n + 1, dtype=i32, device=d)      # hrf_sampler_rays: counter + ids of the rays that take the exact march

    def _alloc_compact(self, n: int, keep: int = 0):
        d, f32, i32 = self.dev, torch.float32, torch.int32
        old = getattr(self, "_compact", None)
        self.cap_rays = n
        self.origins = torch.empty(n, 3, dtype=f32, device=d)
        self.dirs = torch.empty(n, 3, dtype=f32, device=d)
        self.rgba = torch.empty(n, 4, dtype=f32, device=d)
        self.frames = torch.empty(n, dtype=i32, device=d)
        self.cams = torch.empty(n, dtype=i32, device=d)
        self.minmax = torch.empty(n, 2, dtype=f32, device=d)
        self.count = torch.empty(n, dtype=i32, device=d)
        self.ridx = torch.empty(n, dtype=torch.int64, device=d)
        self.offsets = torch.empty(n, dtype=i32, device=d)    # first staged slot of the ray (= its candidates' offset)
        self.kept = torch.empty(n, dtype=i32, device=d)       # staged samples of the ray (passed the occupancy predicate)
        self._compact = (self.origins, self.dirs, self.rgba, self.frames, self.cams, self.minmax, self.count, self.ridx,
                         self.offsets, self.kept)
        if old is not None and keep > 0:  # rays of earlier iterations of the step
            for dst, src in zip(self._compact, old):
                dst[:keep].copy_(src[:keep])

    def _alloc_pre(self, n: int):
        self.cap_pre = n
        self.t0 = torch.empty(n, dtype=torch.float32, device=self.dev)


class StepCollector:
    def __init__(self, model, loader, samples_max: int, rays_initial: int, cap_rays: int = 1 << 18, cap_pre: int = 1 << 24,
                 pipelined: bool = True, seed: int = 0x5eed, spec_margin: float = 1.03, spec_history: int = 1,
                 sort_batch: bool = True):
        self.model, self.loader = model, loader
        self.samples_max, self.rays_initial = samples_max, rays_initial
        self.dev = model.table_params.device
        self.pipelined = pipelined
        self.iterations_prefetched = self.iterations_classic = 0   # statistics
        self.margin = 1.15   # drawn rays prefetched for the next step / drawn rays this step needed beyond rays_initial
        self.auto_prefetch = True                # collect() issues the prefetch itself (see collect)
        self.cap_samples = int(samples_max * 1.1) + samples_max  # one overshooting iteration still fits
        # [0] samples handed to the pruning pass, [1] samples it encoded -- accumulated by the march itself
        self.totals = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self._jitter_stream = (int(seed) * 2654435761) & 0xFFFFFFFF   # one counter-based stream per march launch
        cap_draw = max(cap_rays, rays_initial, 1 << 15)
        self.sets = [_RaySet(self.dev, cap_draw, cap_pre), _RaySet(self.dev, cap_draw, cap_pre)]
        self.tmp = _RaySet(self.dev, max(rays_initial, 1 << 15), cap_pre)  # drawn-level scratch of classic iterations
        self.cur = 0
        self._pending: Optional[int] = None      # drawn rays to prefetch for the next step
        self.side = torch.cuda.Stream(device=self.dev) if pipelined else None
        d, i32 = self.dev, torch.int32
        self.t = torch.empty(self.cap_samples, dtype=torch.float32, device=d)
        self.ray = torch.empty(self.cap_samples, dtype=torch.int64, device=d)
        self.sizes = torch.empty(3, dtype=i32, device=d)
        self.plan = torch.empty(16, dtype=torch.int64, device=d)
        self.plan_host = torch.empty(16, dtype=torch.int64).pin_memory() if d.type == "cuda" else torch.empty(16, dtype=torch.int64)
        self.speculate = True          # march all the rays a step is expected to need in one launch (see collect)
        # issue the next step's sampler stages behind the march (True: the march has the CUs to itself, the sampler kernels fill the time
        # around the plan read-back) or next to it (False). Round 6 measured the two at frozen model states three times: next to the march
        # -0.8 % and -0.9 % (two-process means; rounds alternating A B A B) and +0.8 % (alternating A B B A on another checkpoint): no robust
        # difference (profiles/r06_stepbench_ab.txt); the round-2 placement stays.
        self.prefetch_after_march = True
        self._prefetch_due = False
        self._predicted_total = 0      # drawn rays the previous steps' batch-growing loops used (max of the last few)
        self._recent_totals = []
        # speculation margin over that prediction: a shortfall costs a second, badly filled march launch + plan + sync
        # (~1 ms when it happens), every per cent of margin ~10 us of march
        # (measured on MI355X, profiles/r02 notes in DESIGN.md: 3 % -> 1.30 march launches per step, 8 % -> 1.09, 12 % -> 1.02,
        # step time unchanged within noise)
        self.spec_margin = float(spec_margin)
        self.spec_history = int(spec_history)
        self.march_launch_rays = 0     # drawn rays marched speculatively (statistics: waste = this - rays the loops used)
        self.march_launches = 0        # prune-march launches (statistics)
        self.rays_used = 0             # drawn rays the loops used (statistics)
        self.n_dev = torch.empty(1, dtype=i32, device=d)
        keys = max(model.num_segments, min(model.num_frames, 1024))
        self.order_ws = torch.empty(2 * keys, dtype=i32, device=d)
        # The finished batch is laid out BY FRAME (hrf_pack_runs_sorted): the rays of a batch are i.i.d. draws
        # (data_loader.py:540-546) and render / the loss means / the gradient sums do not depend on their order, but in
        # frame order a workgroup of the encode kernels reads one temporal segment's tables, the tiles of the binned
        # gradient scatter hold one segment each, and every XCD's L2 sees one or two frames (as in the prune march).
        self.sort_batch = bool(sort_batch)
        self.sorted = None                       # per-ray arrays of the current batch in frame order (a _RaySet's compact part)
        self.batch_sorted = False
        self._alloc_march(cap_draw, cap_pre)

    # ------------------------------------------------------------------ buffers
    @property
    def ridx(self):  # pixel ids of the rays of the current batch (tests)
        return self.sorted.ridx if self.batch_sorted else self.sets[self.cur].ridx

    def _alloc_march(self, n_rays: int, n_pre: int):
        d, i32 = self.dev, torch.int32
        if n_rays > getattr(self, "cap_march", 0):
            self.cap_march = n_rays
            self.ray_cnt = torch.empty(n_rays, dtype=i32, device=d)
            self.ray_eval = torch.empty(n_rays, dtype=i32, device=d)
            self.out_off = torch.empty(n_rays + 1, dtype=i32, device=d)
            self.order = torch.empty(n_rays, dtype=i32, device=d)
            self.march_ws = torch.zeros(2 * ((n_rays + 4095) // 4096) + 8, dtype=i32, device=d)
            self.cnt_sorted = torch.empty(n_rays, dtype=i32, device=d)
            self.off_sorted = torch.empty(n_rays + 1, dtype=i32, device=d)
        if n_pre > getattr(self, "cap_stage", 0):
            self.cap_stage = n_pre
            self.t_stage = torch.empty(n_pre, dtype=torch.float32, device=d)

    @property
    def evaluated(self):      # samples encoded by the pruning pass so far (device scalar)
        return self.totals[1]

    @property
    def pre_samples(self):    # samples handed to the pruning pass so far (device scalar)
        return self.totals[0]

    def _next_jitter_seed(self) -> int:
        self._jitter_stream = (self._jitter_stream + 0x9E3779B9) & 0xFFFFFFFF
        return self._jitter_stream or 1

    @staticmethod
    def _scan(x, is_u8, n, out, ws):
        check(_lib.lib().hrf_scan_exclusive(ptr(x), 1 if is_u8 else 0, n, ptr(out), ptr(ws if n > 8192 else None), stream_ptr()))

    # ------------------------------------------------------------------ sampler stages (model independent)
    def _sampler_pass(self, draw: _RaySet, dst: _RaySet, rb: int, r0: int) -> None:
        """Draw r0 pixel ids and run ray generation, compaction (into dst's per-ray arrays at row rb) and sample
        generation (into draw.t0 at draw.offsets) on the current stream. No host synchronisation."""
        L, ld, st = _lib.lib(), self.loader, stream_ptr()
        if r0 > draw.cap_draw:
            draw._alloc_draw(int(r0 * 1.25))
        if rb + r0 > dst.cap_rays:
            dst._alloc_compact(int((rb + r0) * 1.5), rb)
        width, height = ld.resolution
        P = width * height
        idx = ld.draw_ray_indices(r0, out=draw.idx)                     # data_loader.py:540-546
        reader = ld.pool_reader() if hasattr(ld, "pool_reader") else None    # data_lock + stream ordering vs the replacer
        if reader is not None:
            reader.__enter__()
        try:
            self._sampler_launches(draw, dst, rb, r0, idx)
        finally:
            if reader is not None:
                reader.__exit__(None, None, None)

    def _sampler_launches(self, draw: _RaySet, dst: _RaySet, rb: int, r0: int, idx) -> None:
        L, ld, st = _lib.lib(), self.loader, stream_ptr()
        width, height = ld.resolution
        P = width * height
        occ = 1 if ld.occupancy else 0
        G = int(ld.occupancy_grid_resolution)
        land = ld.landscape_mode_cuda.view(torch.uint8)
        tex = ld.grid_texture_objects_cuda if occ else None
        with ops._span("sampler_kernels", r0):
            check(L.hrf_sampler_rays(ptr(ld.inverse_krs_cuda), ptr(ld.camera_origins_cuda), ptr(land), ptr(idx), ptr(tex),
                                     ptr(ld.aabb), None, r0, G, width, height, STEP, occ, ptr(draw.dirs_all),
                                     ptr(draw.mm_all), ptr(draw.mask), ptr(draw.count_all), ptr(draw.pre_ws) if occ else None, st))
            self._scan(draw.mask, True, r0, draw.slot, draw.scan_ws)
            self._scan(draw.count_all, False, r0, draw.cand_all, draw.scan_ws)   # candidates: known before compaction
            check(L.hrf_sampler_compact_rays(ptr(idx), ptr(draw.mask), ptr(draw.slot), ptr(draw.dirs_all), ptr(draw.mm_all),
                                             ptr(draw.count_all), ptr(ld.pixel_colors), ptr(ld.camera_origins_cuda),
                                             ptr(ld.frame_numbers_cuda), ptr(ld.camera_numbers_cuda), r0, P,
                                             ptr(dst.origins[rb:]), ptr(dst.dirs[rb:]), ptr(dst.rgba[rb:]),
                                             ptr(dst.frames[rb:]), ptr(dst.cams[rb:]), ptr(dst.minmax[rb:]),
                                             ptr(dst.count[rb:]), ptr(dst.ridx[rb:]), ptr(draw.cand_all),
                                             ptr(dst.offsets[rb:]), st))
            n_dev = draw.slot[r0:]
            # ONE pass: ray r fills a prefix of its candidates' slot range [offsets[r], offsets[r] + count[r]) and reports
            # how many passed the occupancy predicate (the two-pass form evaluated the predicate twice and needed a scan
            # of the surviving counts in between)
            check(L.hrf_sampler_samples(ptr(dst.ridx[rb:]), ptr(tex), ptr(dst.origins[rb:]), ptr(dst.dirs[rb:]),
                                        ptr(dst.minmax[rb:]), ptr(dst.count[rb:]), ptr(dst.offsets[rb:]), r0, ptr(n_dev), P, G,
                                        STEP, occ, ptr(dst.kept[rb:]), ptr(draw.t0), None, draw.cap_pre, st))

    # ------------------------------------------------------------------ prune march over a range of compacted rays
    def _march_pass(self, rays: _RaySet, base: int, upper: int, n_dev, staged: _RaySet, total_pos: int,
                    samp_base: int) -> Tuple[int, int, int]:
        """March the compacted rays [base, base + *n_dev) of `rays` (at most `upper`), whose staged samples are
        staged.t0[offsets[k] : offsets[k] + kept[k]] (offsets / kept: per compacted ray, in `rays`); one host sync;
        survivors packed at samp_base. total_pos: number of drawn rays of the staged set (its candidate total decides
        whether the staging overflowed).
        -> (rays, candidates of the whole staged set, surviving samples or -1 on staging overflow)."""
        L, m, st = _lib.lib(), self.model, stream_ptr()
        self.march_launches += 1
        self._alloc_march(upper, staged.cap_pre)
        m._refresh_half()
        sw1, sw2 = m._sigma_w()
        frames = rays.frames[base:]
        ray_start, ray_len = rays.offsets[base:], rays.kept[base:]
        order = None
        if m.num_frames > 1:  # schedule only: rays by frame, one eighth per XCD
            order = ops.ray_segment_order(frames[:upper], m, n_dev, out=self.order, workspace=self.order_ws)
        with ops._span("prune_march", 1):
            # jitter (volume_rendering.py:63-64) is drawn inside the kernel from a counter-based stream
            check(L.hrf_prune_march(ptr(rays.origins[base:]), ptr(rays.dirs[base:]), ptr(frames), ptr(ray_start),
                                    ptr(staged.t0), None, STEP, 1e-4, 1e-4, ptr(m.frame_numbers_to_segment_numbers),
                                    ptr(m.frame_numbers_to_normalized_local_frame_numbers), ptr(m._tables_h),
                                    ptr(m.vectors), ptr(m._seg_meta), m.num_segments, m.vec_res, ptr(sw1), ptr(sw2),
                                    float(m.density_scale), upper, ptr(n_dev), staged.cap_pre, ptr(self.t_stage), None,
                                    ptr(self.ray_cnt), None, ptr(order), ptr(ray_len), self._next_jitter_seed(),
                                    ptr(self.totals), ops._mlp_mode(sw1, sw2), st))
        self._scan(self.ray_cnt, False, upper, self.out_off, self.march_ws)
        torch.stack([n_dev[0], staged.cand_all[total_pos], self.out_off[upper]], out=self.sizes)
        R, n0_total, n1 = (int(v) for v in self.sizes.cpu())            # the single host sync of the iteration
        if n0_total > staged.cap_pre:                                    # rare: staging overflowed (kernels guard the bound)
            return R, n0_total, -1
        self._reserve_samples(samp_base + n1, samp_base)
        check(L.hrf_pack_runs(ptr(ray_start), ptr(self.ray_cnt), ptr(self.out_off), ptr(self.t_stage), R, None, base,
                              ptr(self.t[samp_base:]), ptr(self.ray[samp_base:]), st))
        return R, n0_total, n1

    def _reserve_samples(self, need: int, keep: int) -> None:
        """Room for `need` packed samples, the first `keep` of them already written. The buffers start at 2.1 x samples_max (a
        full batch plus one overshooting iteration); an iteration can exceed that when rays_initial x samples per ray is itself
        a multiple of the budget (small budgets, an untrained field that prunes nothing). The reference's loop takes whatever
        such an iteration yields and merge_input_batches cuts the batch at 1.1 x samples_max (input.py:33-47, as collect() does
        below), so the buffers grow instead of refusing."""
        if need <= self.cap_samples:
            return
        cap = max(int(need), int(self.cap_samples * 1.5))
        t, ray = torch.empty(cap, dtype=torch.float32, device=self.dev), torch.empty(cap, dtype=torch.int64, device=self.dev)
        t[:keep].copy_(self.t[:keep])
        ray[:keep].copy_(self.ray[:keep])
        self.t, self.ray, self.cap_samples = t, ray, cap
        self.t_alt = self.ray_alt = None                 # (_resort_packed's second pair: reallocated at the new size when needed)

    def _march_range(self, rs: _RaySet, ray_base: int, r_from: int, d_from: int, d_to: int) -> None:
        """March the compacted rays of the prefetched drawn rays [d_from, d_to) -- compacted indices [r_from, slot[d_to)) --
        writing their visible-sample counts to ray_cnt[r_from - ray_base ...]. No synchronisation."""
        L, m, st = _lib.lib(), self.model, stream_ptr()
        upper = d_to - d_from
        self.march_launches += 1
        m._refresh_half()
        sw1, sw2 = m._sigma_w()
        frames = rs.frames[r_from:]
        if r_from == 0:
            n_dev = rs.slot[d_to:]
        else:
            torch.sub(rs.slot[d_to:d_to + 1], r_from, out=self.n_dev)
            n_dev = self.n_dev
        order = None
        if m.num_frames > 1:  # schedule only: rays by frame, one eighth per XCD
            order = ops.ray_segment_order(frames[:upper], m, n_dev, out=self.order, workspace=self.order_ws)
        with ops._span("prune_march", 1):
            # jitter (volume_rendering.py:63-64) is drawn inside the kernel from a counter-based stream
            check(L.hrf_prune_march(ptr(rs.origins[r_from:]), ptr(rs.dirs[r_from:]), ptr(frames), ptr(rs.offsets[r_from:]),
                                    ptr(rs.t0), None, STEP, 1e-4, 1e-4, ptr(m.frame_numbers_to_segment_numbers),
                                    ptr(m.frame_numbers_to_normalized_local_frame_numbers), ptr(m._tables_h),
                                    ptr(m.vectors), ptr(m._seg_meta), m.num_segments, m.vec_res, ptr(sw1), ptr(sw2),
                                    float(m.density_scale), upper, ptr(n_dev), rs.cap_pre, ptr(self.t_stage), None,
                                    ptr(self.ray_cnt[r_from - ray_base:]), None, ptr(order), ptr(rs.kept[r_from:]),
                                    self._next_jitter_seed(), ptr(self.totals), ops._mlp_mode(sw1, sw2), st))

    def _pack_sorted(self, rs: _RaySet, n_rays: int, ray_start=None, t_src=None) -> None:
        """Pack the visible samples of the compacted rays [0, n_rays) of `rs` into the step buffers in FRAME order, and
        their per-ray records into self.sorted (counting sort of the rays by frame, counts carried along; scan; pack).
        Source of ray r's samples: t_src[ray_start[r] : ray_start[r] + ray_cnt[r]] (default: the march's staging)."""
        ray_start = rs.offsets if ray_start is None else ray_start
        t_src = self.t_stage if t_src is None else t_src
        L, m, st = _lib.lib(), self.model, stream_ptr()
        if self.sorted is None or self.sorted.cap_rays < n_rays:
            cap = max(int(n_rays * 1.5), 1 << 15)
            if self.sorted is None:
                self.sorted = _RaySet.__new__(_RaySet)
                self.sorted.dev = self.dev
            self.sorted._alloc_compact(cap)
        so = self.sorted
        if m.num_frames <= 1024:
            table, keys = m._frame_rank, m.num_frames
        else:
            table, keys = m.frame_numbers_to_segment_numbers, m.num_segments
        check(L.hrf_ray_segment_order_values(ptr(rs.frames), ptr(table), n_rays, None, keys, ptr(self.order_ws), ptr(self.order),
                                             ptr(self.ray_cnt), ptr(self.cnt_sorted), st))
        self._scan(self.cnt_sorted, False, n_rays, self.off_sorted, self.march_ws)
        check(L.hrf_pack_runs_sorted(ptr(self.order), ptr(ray_start), ptr(self.ray_cnt), ptr(self.off_sorted),
                                     ptr(t_src), n_rays, ptr(rs.origins), ptr(rs.dirs), ptr(rs.rgba), ptr(rs.frames),
                                     ptr(rs.cams), ptr(rs.minmax), ptr(rs.ridx), ptr(so.origins), ptr(so.dirs), ptr(so.rgba),
                                     ptr(so.frames), ptr(so.cams), ptr(so.minmax), ptr(so.ridx), ptr(self.t), ptr(self.ray), st))

    def _resort_packed(self, rs: _RaySet, n_rays: int, n_samples: int) -> None:
        """Frame order for a batch that is already packed in draw order in self.t / self.ray (rays [0, n_rays) of `rs`)."""
        L, st = _lib.lib(), stream_ptr()
        self._alloc_march(n_rays, 0)
        if getattr(self, "t_alt", None) is None:
            self.t_alt = torch.empty_like(self.t)
            self.ray_alt = torch.empty_like(self.ray)
        starts = ops.ray_offsets(self.ray[:n_samples], n_rays)            # first packed sample of every ray (+ the total)
        torch.sub(starts[1:], starts[:-1], out=self.ray_cnt[:n_rays])
        src_t, src_ray = self.t, self.ray
        self.t, self.ray = self.t_alt, self.ray_alt                        # _pack_sorted writes self.t / self.ray
        self._pack_sorted(rs, n_rays, ray_start=starts, t_src=src_t)
        self.t_alt, self.ray_alt = src_t, src_ray

    def _plan(self, rs: _RaySet, ray_base: int, used: int, spec_end: int, r0: int, total_rays: int, total_samples: int,
              avail: int):
        """Prefix sums of the marched rays' visible samples + the trainer loop replayed over them on the device
        (hrf_batch_plan) + the ONE host synchronisation. -> the plan as a tuple of ints."""
        L, st = _lib.lib(), stream_ptr()
        self._scan(self.ray_cnt, False, spec_end - used, self.out_off, self.march_ws)
        check(L.hrf_batch_plan(ptr(rs.slot), ptr(self.out_off), ray_base, used, spec_end, r0, total_rays, total_samples,
                               self.samples_max, ptr(rs.cand_all[avail:]), ptr(self.plan), st))
        self.plan_host.copy_(self.plan, non_blocking=True)
        if self._prefetch_due:
            # The sampler stages of the step AFTER this one go to the second stream now: the march of this step is already
            # queued (it has the CUs to itself), and the sampler kernels fill the time the device would otherwise idle while
            # the host waits for the plan and then issues the first launches of the training step one by one.
            self._prefetch_due = False
            self.prefetch()
        torch.cuda.current_stream().synchronize()
        return tuple(int(v) for v in self.plan_host.tolist())

    def _classic_iteration(self, rs: _RaySet, r0: int, ray_base: int, samp_base: int) -> Tuple[int, int]:
        """Sampler stages + march for r0 freshly drawn rays, appended to the batch at ray_base / samp_base."""
        while True:
            self._sampler_pass(self.tmp, rs, ray_base, r0)
            R, n0, n1 = self._march_pass(rs, ray_base, r0, self.tmp.slot[r0:], self.tmp, r0, samp_base)
            if n1 >= 0:
                return R, n1
            self.tmp._alloc_pre(int(n0 * 1.5))  # grow the staging and redo (fresh draw)

    # ------------------------------------------------------------------ prefetch of the next step's sampler stages
    def prefetch(self) -> None:
        """Run the sampler stages of the NEXT step on the side stream. No-op when not pipelined or nothing is due."""
        if not self.pipelined or self._pending is None:
            return
        n, self._pending = self._pending, None
        nxt = self.sets[self.cur ^ 1]
        self.side.wait_stream(torch.cuda.current_stream())  # the other set's last readers were enqueued before this
        with torch.cuda.stream(self.side):
            self._sampler_pass(nxt, nxt, 0, n)
            nxt.n_drawn = n
            nxt.ready = torch.cuda.Event()
            nxt.ready.record()

    def wait_prefetch(self) -> None:
        """Make the current stream wait for an in-flight prefetch (before anything rewrites the image pool)."""
        nxt = self.sets[self.cur ^ 1]
        if nxt.ready is not None:
            torch.cuda.current_stream().wait_event(nxt.ready)

    # ------------------------------------------------------------------ trainer.py:138-172
    def collect(self):
        """-> (InputBatch of views into the step buffers, rays drawn, None). Sample statistics accumulate on the device in
        `totals` (pre-prune samples, samples encoded by the pruning pass)."""
        if self.pipelined:
            self.cur ^= 1
        rs = self.sets[self.cur]
        if rs.ready is not None:
            torch.cuda.current_stream().wait_event(rs.ready)
            rs.ready = None
        # The sampler stages of the step AFTER this one start now, on the second stream: they run under this step's
        # prune march, which holds only 4 wavefronts per SIMD (128 VGPRs) and is bound by the gather path, so the
        # small sampler kernels fit next to it. (Measured alternatives: under the forward kernels 5.8 ms/step, under
        # the gradient scatter 6.0 -- its workgroups fill every slot and starve them.) With auto_prefetch off the
        # caller places it (data parallel: into the gradient exchange, when the CUs have nothing else to do).
        self._prefetch_due = bool(self.auto_prefetch and self.prefetch_after_march)
        if self.auto_prefetch and not self._prefetch_due:
            self.prefetch()
        avail, rs.n_drawn = rs.n_drawn, 0                 # prefetched drawn rays not consumed yet
        used = 0
        cuts = None
        sorted_now = False
        r0 = self.rays_initial
        total_rays = total_samples = 0
        ray_base = samp_base = 0
        while True:
            if avail - used >= r0:
                # The prefetched set covers the next iteration. March, in the same launch, everything the loop is expected
                # to need (what the previous step used + 3 %): per-ray results do not depend on the launch they are
                # computed in, so the iterations of trainer.py:143-163 become prefix lookups replayed on the device --
                # one launch and one host sync per step instead of one per iteration, and no tiny first launch
                # (rays_initial drawn rays are ~1 000 surviving rays: a quarter of a wavefront slot per CU). When the
                # marched rays fall short, only the missing ones are marched (a second, small launch) and the loop is
                # replayed over the longer prefix.
                want = r0
                if self.speculate and used == 0:
                    want = max(r0, int(self._predicted_total * self.spec_margin) + 256)
                c_used, c_r0, c_tr, c_ts = used, r0, total_rays, total_samples      # loop state at the start of the chunk
                marched_to, r_marched = used, ray_base
                target = min(avail, used + want)
                self._alloc_march(avail - used, rs.cap_pre)                          # no reallocation inside the chunk
                overflow = False
                while True:
                    if target > marched_to:
                        self._march_range(rs, ray_base, r_marched, marched_to, target)
                        marched_to = target
                    (done, iters, used_new, r0_next, r_abs, n1, err, tr, cand_total, r_marched, q1, q2, q3, rays_c, _,
                     _) = self._plan(rs, ray_base, c_used, target, c_r0, c_tr, c_ts, avail)
                    if cand_total > rs.cap_pre:           # rare: the staging overflowed (the kernels guard the bound)
                        overflow = True
                        break
                    if err:
                        raise AssertionError("There is probably a problem with the predicted geometry.")   # trainer.py:158
                    if done or used_new + r0_next > avail:
                        break                             # finished, or the set cannot serve the next iteration
                    target = min(avail, used_new + r0_next + 64)
                if overflow:                              # drop the set, go classic with a larger staging buffer
                    rs._alloc_pre(int(cand_total * 1.5))
                    avail = 0
                    continue
                self._reserve_samples(samp_base + n1, samp_base)
                whole = ray_base == 0 and samp_base == 0 and done and n1 <= int(self.samples_max * 1.1)
                if whole and self.sort_batch and r_abs > 0 and self.model.num_frames > 1:
                    self._pack_sorted(rs, r_abs)          # the chunk is the whole batch: lay it out by frame
                    sorted_now = True
                elif r_abs > ray_base:
                    check(_lib.lib().hrf_pack_runs(ptr(rs.offsets[ray_base:]), ptr(self.ray_cnt), ptr(self.out_off),
                                                   ptr(self.t_stage), r_abs - ray_base, None, ray_base,
                                                   ptr(self.t[samp_base:]), ptr(self.ray[samp_base:]), stream_ptr()))
                # ray-aligned cut points of the batch (quarter points of the rays): valid when this chunk is the whole batch
                cuts = ([(rays_c * k // 4, q) for k, q in ((1, q1), (2, q2), (3, q3))]
                        if (ray_base == 0 and samp_base == 0 and done and not sorted_now) else None)
                self.iterations_prefetched += iters
                self.march_launch_rays += marched_to - c_used
                ray_base, samp_base = r_abs, samp_base + n1
                total_rays, total_samples, used, r0 = tr, total_samples + n1, used_new, r0_next
                if done:
                    break
                continue                                  # the set is exhausted: classic iterations from here
            avail = 0                                     # whatever is left of the prefetched set is not used
            cuts = None
            assert not sorted_now
            r_it = r0
            R, n1 = self._classic_iteration(rs, r_it, ray_base, samp_base)
            self.iterations_classic += 1
            ray_base += R
            samp_base += n1
            total_rays += r_it
            total_samples += n1
            if total_samples < 0.9 * self.samples_max:
                avg = total_samples / total_rays
                assert avg > 0, "There is probably a problem with the predicted geometry."
                r0 = int((self.samples_max - total_samples) / avg)
            else:
                break
        self.rays_used += total_rays
        self._recent_totals = (self._recent_totals + [total_rays])[-self.spec_history:]
        self._predicted_total = max(self._recent_totals)
        if self._prefetch_due:            # no speculative chunk ran in this step (first steps, exhausted sets)
            self._prefetch_due = False
            self.prefetch()
        if self.pipelined:  # next step: what this one needed, plus a margin
            self._pending = self.rays_initial + int(self.margin * max(total_rays - self.rays_initial, self.rays_initial)) + 1024
        n_rays, n_samples = ray_base, samp_base
        max_num = int(self.samples_max * 1.1)
        if n_samples > max_num:                                          # humanrf/input.py:33-47
            cutoff = int(self.ray[max_num].item())
            n_samples = int(torch.searchsorted(self.ray[:n_samples], cutoff).item())
            n_rays = cutoff
            cuts = None
        if (not sorted_now and self.sort_batch and n_rays > 0 and n_samples > 0 and self.model.num_frames > 1):
            # the batch was assembled from several chunks (first steps of a run, a prefetched set that fell short): lay the
            # packed arrays out by frame after the fact
            self._resort_packed(rs, n_rays, n_samples)
            sorted_now = True
            cuts = None
        self.batch_sorted = sorted_now
        if sorted_now:
            rs = self.sorted
        ib = InputBatch(ray_origins=rs.origins[:n_rays], ray_directions=rs.dirs[:n_rays], minmaxes=rs.minmax[:n_rays],
                        rgba=rs.rgba[:n_rays], frame_numbers=rs.frames[:n_rays].view(-1, 1),
                        camera_numbers=rs.cams[:n_rays].view(-1, 1), sample_distances=self.t[:n_samples].view(-1, 1),
                        ray_indices=self.ray[:n_samples], width=self.loader.resolution[0], height=self.loader.resolution[1])
        # (ray, sample) positions where the batch may be cut into pieces that end on ray boundaries (TrainEngine pipelines
        # the pieces); None when the batch was assembled from several chunks
        ib._cuts = cuts
        ib._sorted_by_frame = sorted_now
        # first packed sample of every ray (+ the total): the scan the frame-ordered pack was made from, so that the training step
        # need not recompute it from the int64 ray ids (k_ray_offsets)
        ib._ray_start = self.off_sorted[:n_rays + 1] if sorted_now else None
        return ib, total_rays, None
