"""Thin tensor-level wrappers over the C ABI (include/hrf.h). They only check arguments the way the reference
does (contiguity + device -> RuntimeError, actorshq/toolbox/native/utils.cuh:5-19), allocate outputs through
torch (ownership as in ray_sampler.cu:233-235) and pass raw pointers + the current stream down."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

STEP = 4e-4  # render_step_size / raymarching_step_size (data_loader.py:573, volume_rendering.py:47,92)


class KernelTimer:
    """Optional per-kernel timing with events recorded on the launch stream (bench.py's roofline leg).
    `names`: only these spans are timed (None = all; event recording costs host time, so the default bench run times
    the roofline kernel only). Usage: ops.TIMER = KernelTimer({"prune_march"}); ...; ops.TIMER.summary()."""

    def __init__(self, names=None):
        self.names = names
        self.records = {}
        self._pool = []

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def span(self, name: str, units: int):
        if self.names is not None and name not in self.names:
            return _NOSPAN
        return _Span(self, name, units)

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            out[name] = {"launches": len(recs), "ms_total": ms, "units": sum(u for _, _, u in recs)}
        return out


class _Span:
    def __init__(self, timer, name, units):
        self.t, self.name, self.units = timer, name, units

    def __enter__(self):
        self.a = self.t._event()
        self.b = self.t._event()
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.t.records.setdefault(self.name, []).append((self.a, self.b, self.units))


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


TIMER: Optional[KernelTimer] = None
_NOSPAN = _NoSpan()


def _span(name: str, units: int):
    return TIMER.span(name, units) if TIMER is not None else _NOSPAN


class Arena:
    """Named persistent output buffers for a caller that runs the same kernel sequence every step and never holds
    an output across steps (TrainEngine.train_step): sample counts change from step to step, and handing each new
    size to the caching allocator ends in an occasional device malloc in the middle of the step (measured: 18 ms
    stalls, ~1 ms per step on average). Buffers grow geometrically and are handed out as views."""

    def __init__(self):
        self.bufs = {}

    def get(self, tag: str, shape, dtype, device):
        n = 1
        for d in shape:
            n *= int(d)
        buf = self.bufs.get(tag)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(max(int(n * 1.25), 1024), dtype=dtype, device=device)
            self.bufs[tag] = buf
        return buf[:n].view(*shape)


ARENA: Optional[Arena] = None


def _new(tag: str, shape, dtype, device, zero: bool = False) -> torch.Tensor:
    """Output allocation: through torch (ownership as in the reference) unless an Arena is active."""
    if ARENA is None:
        return (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=device)
    t = ARENA.get(tag, shape, dtype, device)
    return t.zero_() if zero else t


def _chk(t: Optional[torch.Tensor], name: str, dtype=None, cuda: bool = True):
    if t is None:
        return
    if not t.is_contiguous():
        raise RuntimeError(f"Tensor not contiguous: {name}")
    if cuda and not t.is_cuda:
        raise RuntimeError(f"Tensor is not on the expected device: {name}")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"Tensor {name} has dtype {t.dtype}, expected {dtype}")


def scan_exclusive(x: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix sum of an int32 / uint8 / bool vector -> int32 (n+1,), last element = total."""
    is_u8 = x.dtype in (torch.uint8, torch.bool)
    if not is_u8:
        _chk(x, "scan input", torch.int32)
    n = x.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=x.device)
    ws = torch.zeros(2 * ((n + 4095) // 4096) + 8, dtype=torch.int32, device=x.device) if n > 8192 else None
    check(_lib.lib().hrf_scan_exclusive(ptr(x), 1 if is_u8 else 0, n, ptr(out), ptr(ws), stream_ptr()))
    return out


def query_prep(ray_origins, ray_dirs, ray_frames, sample_ray, t, jitter, frame_to_segment, frame_to_local,
               step: float = STEP):
    """-> xyzt (n,4) fp32, segment (n,) int32; t is updated in place when jitter is given."""
    n = t.numel()
    for a, nm, dt in ((ray_origins, "ray_origins", torch.float32), (ray_dirs, "ray_directions", torch.float32),
                      (ray_frames, "frame_numbers", torch.int32), (sample_ray, "ray_indices", torch.int64),
                      (t, "sample_distances", torch.float32), (jitter, "jitter", torch.float32),
                      (frame_to_segment, "frame_to_segment", torch.int32),
                      (frame_to_local, "frame_to_local", torch.float32)):
        _chk(a, nm, dt)
    xyzt = _new("xyzt", (n, 4), torch.float32, t.device)
    seg = _new("seg", (n,), torch.int32, t.device)
    check(_lib.lib().hrf_query_prep(ptr(ray_origins), ptr(ray_dirs), ptr(ray_frames), ptr(sample_ray), ptr(t),
                                    ptr(jitter), step, ptr(frame_to_segment), ptr(frame_to_local), n, ptr(xyzt),
                                    ptr(seg), stream_ptr()))
    return xyzt, seg


def encode4d_fwd(xyzt, seg, tables_h, vectors, seg_meta_dev, num_segments: int, save_enc: bool):
    _chk(xyzt, "xyzt", torch.float32); _chk(seg, "segment", torch.int32)
    _chk(tables_h, "tables", torch.float16); _chk(vectors, "vectors", torch.float32)
    n = xyzt.shape[0]
    feats = _new("feats", (n, 32), torch.float16, xyzt.device)
    enc = _new("enc", (n, 4, 32), torch.float16, xyzt.device) if save_enc else None
    with _span("encode4d_fwd_save" if save_enc else "encode4d_fwd", n):
        check(_lib.lib().hrf_encode4d_fwd(ptr(xyzt), ptr(seg), ptr(tables_h), ptr(vectors), ptr(seg_meta_dev),
                                          num_segments, vectors.shape[-2], n, ptr(feats), ptr(enc), stream_ptr()))
    return feats, enc


def encode4d_density_fwd(xyzt, seg, tables_h, vectors, seg_meta_dev, num_segments: int, w1, w2, density_scale: float):
    """encode4d_fwd(save_enc=True) + density_mlp_fwd in one launch (hrf_encode4d_density_fwd): -> (feats, enc, h, sigma), bit-identical
    to the two calls."""
    _chk(xyzt, "xyzt", torch.float32); _chk(seg, "segment", torch.int32)
    _chk(tables_h, "tables", torch.float16); _chk(vectors, "vectors", torch.float32)
    mode = _mlp_mode(w1, w2)
    n, dev = xyzt.shape[0], xyzt.device
    feats = _new("feats", (n, 32), torch.float16, dev)
    enc = _new("enc", (n, 4, 32), torch.float16, dev)
    h = _new("h", (n, 16), torch.float16, dev)
    sigma = _new("sigma", (n,), torch.float32, dev)
    with _span("encode4d_fwd_save", n):
        check(_lib.lib().hrf_encode4d_density_fwd(ptr(xyzt), ptr(seg), ptr(tables_h), ptr(vectors), ptr(seg_meta_dev), num_segments,
                                                  vectors.shape[-2], n, ptr(feats), ptr(enc), ptr(w1), ptr(w2), density_scale,
                                                  ptr(h), ptr(sigma), mode, stream_ptr()))
    return feats, enc, h, sigma


def encode4d_bwd(xyzt, seg, enc, vectors, seg_meta_dev, num_segments: int, d_features, grad_scale: float,
                 d_tables, d_vectors, level_major: bool = False, grad_boundary: float = 0.0, flags=None):
    """flags: int32 (1,) found_inf flag of the step, raised when a table gradient is non-finite after the half gradient boundary
    (the atomic table kernels; the binned scatter has its own check)."""
    _chk(flags, "flags", torch.int32)
    _chk(d_features, "d_features"); _chk(enc, "enc_features", torch.float16)
    if d_features.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("d_features must be fp16 or fp32")
    _chk(d_tables, "d_tables", torch.float32); _chk(d_vectors, "d_vectors", torch.float32)  # either may be None
    part = "" if (d_tables is not None and d_vectors is not None) else ("_tables" if d_tables is not None else "_vectors")
    with _span("encode4d_bwd" + part, xyzt.shape[0]):
        check(_lib.lib().hrf_encode4d_bwd(ptr(xyzt), ptr(seg), ptr(enc), ptr(vectors), ptr(seg_meta_dev), num_segments,
                                          vectors.shape[-2], xyzt.shape[0], ptr(d_features),
                                          (2 if level_major else 1) if d_features.dtype == torch.float32 else 0,
                                          grad_scale, float(grad_boundary), ptr(d_tables),
                                          ptr(d_vectors), ptr(flags), stream_ptr()))


class ScatterWorkspace:
    """Device workspace of hrf_encode4d_bwd_tables_binned (include/hrf.h): record queues for batches of up to
    `samples` samples (about 6.3 MB per 1024 samples); `max_level_entries` is the largest level table of the model (the
    binned scatter serves tables of up to 2^19 entries -- log2_hashmap_size 19 on a 100-frame segment, the reference's
    largest default -- and models of up to 1024 temporal segments)."""
    MAX_LEVEL_ENTRIES = 1 << 19
    MAX_SEGMENTS = 1024

    def __init__(self, samples: int, num_segments: int, max_level_entries: int, device):
        nbytes = int(_lib.lib().hrf_scatter_workspace_bytes(int(samples), int(num_segments)))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.samples, self.num_segments = int(samples), int(num_segments)
        self.max_level_entries = int(max_level_entries)

    @staticmethod
    def supports(max_level_entries: int, num_segments: int = 1) -> bool:
        return 0 < int(max_level_entries) <= ScatterWorkspace.MAX_LEVEL_ENTRIES and num_segments <= ScatterWorkspace.MAX_SEGMENTS


def encode4d_bwd_tables_binned(xyzt, seg, vectors, seg_meta_dev, num_segments: int, d_features_lm, grad_scale: float,
                               d_tables, workspace: ScatterWorkspace, flags=None, grad_boundary: float = 0.0):
    """Table gradients of the level-major backward without memory-side atomics (see hrf_encode4d_bwd_tables_binned).
    flags: int32 (1,) found_inf flag of the step (set when a record is non-finite or out of the fixed-point range)."""
    _chk(xyzt, "xyzt", torch.float32); _chk(seg, "segment", torch.int32); _chk(vectors, "vectors", torch.float32)
    _chk(d_features_lm, "d_features", torch.float32); _chk(d_tables, "d_tables", torch.float32)
    _chk(flags, "flags", torch.int32)
    n = xyzt.shape[0]
    if num_segments != workspace.num_segments:
        raise RuntimeError("scatter workspace was built for another model")
    with _span("encode4d_bwd_tables", n):
        check(_lib.lib().hrf_encode4d_bwd_tables_binned(ptr(xyzt), ptr(seg), ptr(vectors), ptr(seg_meta_dev), num_segments,
                                                        vectors.shape[-2], n, ptr(d_features_lm), grad_scale,
                                                        float(grad_boundary), ptr(d_tables),
                                                        ptr(workspace.buf), workspace.samples, workspace.max_level_entries,
                                                        ptr(flags), stream_ptr()))


def scatter_emit(xyzt, seg, vectors, seg_meta_dev, num_segments: int, d_features_lm, grad_scale: float, d_tables,
                 workspace: ScatterWorkspace, grad_boundary: float = 0.0):
    """First half of encode4d_bwd_tables_binned (tile table + record queues; hrf_scatter_emit)."""
    _chk(xyzt, "xyzt", torch.float32); _chk(seg, "segment", torch.int32); _chk(vectors, "vectors", torch.float32)
    _chk(d_features_lm, "d_features", torch.float32); _chk(d_tables, "d_tables", torch.float32)
    n = xyzt.shape[0]
    if num_segments != workspace.num_segments:
        raise RuntimeError("scatter workspace was built for another model")
    with _span("encode4d_bwd_tables", n):
        check(_lib.lib().hrf_scatter_emit(ptr(xyzt), ptr(seg), ptr(vectors), ptr(seg_meta_dev), num_segments, vectors.shape[-2], n,
                                          ptr(d_features_lm), grad_scale, float(grad_boundary), ptr(d_tables), ptr(workspace.buf),
                                          workspace.samples, workspace.max_level_entries, stream_ptr()))


def scatter_accumulate(seg_meta_dev, num_segments: int, d_tables, workspace: ScatterWorkspace, flags=None, seg_first: int = 0,
                       seg_count: int = 0):
    """Second half (hrf_scatter_accumulate): the temporal segments [seg_first, seg_first + seg_count), or every segment the
    batch touches when seg_count <= 0. The data-parallel step calls it group by group (TrainEngine.train_step)."""
    _chk(d_tables, "d_tables", torch.float32); _chk(flags, "flags", torch.int32)
    with _span("encode4d_bwd_tables_accumulate", 0):
        check(_lib.lib().hrf_scatter_accumulate(ptr(seg_meta_dev), num_segments, ptr(d_tables), ptr(workspace.buf), workspace.samples,
                                                workspace.max_level_entries, ptr(flags), int(seg_first), int(seg_count), stream_ptr()))


def scatter_accumulate_signalled(seg_meta_dev, num_segments: int, d_tables, workspace: ScatterWorkspace, flags, groups, group_done):
    """ONE accumulate launch over every temporal segment by id with completion signals per group (hrf_scatter_accumulate_signalled):
    `groups` = lists of consecutive segment ids, ascending and disjoint (at most 8); `group_done` = int64 device tensor of at least
    len(groups) counters that only ever grow -- by len(group) * scatter_signals_per_segment(workspace) per call. A stream made to wait
    for the new total (stream_wait_value64) may read the group's table gradients while the launch is still at work on later groups."""
    _chk(d_tables, "d_tables", torch.float32); _chk(flags, "flags", torch.int32); _chk(group_done, "group_done", torch.int64)
    if not 1 <= len(groups) <= 8 or group_done.numel() < len(groups):
        raise RuntimeError("scatter_accumulate_signalled: 1..8 groups, one counter each")
    bounds = (ctypes.c_int32 * (2 * len(groups)))(*[v for grp in groups for v in (int(grp[0]), int(grp[-1]))])
    with _span("encode4d_bwd_tables_accumulate", 0):
        check(_lib.lib().hrf_scatter_accumulate_signalled(ptr(seg_meta_dev), num_segments, ptr(d_tables), ptr(workspace.buf),
                                                          workspace.samples, workspace.max_level_entries, ptr(flags),
                                                          ctypes.cast(bounds, ctypes.c_void_p), len(groups), ptr(group_done), stream_ptr()))


def scatter_signals_per_segment(workspace: ScatterWorkspace) -> int:
    n = int(_lib.lib().hrf_scatter_signals_per_segment(int(workspace.max_level_entries)))
    if n <= 0:
        raise RuntimeError("hrf_scatter_signals_per_segment: level tables beyond the binned scatter")
    return n


def can_stream_wait_value() -> bool:
    return bool(_lib.lib().hrf_can_stream_wait_value())


def stream_wait_value64(counter: torch.Tensor, index: int, value: int) -> None:
    """The CURRENT stream waits until counter[index] (int64, device) >= value (hipStreamWaitValue64). Enqueue the wait AFTER the launch
    that advances the counter: streams share hardware queues, and a wait in front of that launch in the same queue is never released."""
    _chk(counter, "counter", torch.int64)
    check(_lib.lib().hrf_stream_wait_value64(stream_ptr(), counter.data_ptr() + 8 * int(index), int(value)))


def _mlp_mode(*weights) -> int:
    """The arithmetic type of the MLP kernels is the dtype of the 16-bit weight copies handed in: torch.float16 ->
    mlp_bf16 = 0 (tcnn's FullyFusedMLP), torch.bfloat16 -> 1 (see include/hrf.h)."""
    dt = weights[0].dtype
    if dt not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"MLP weights must be float16 or bfloat16 copies, got {dt}")
    if any(w.dtype != dt for w in weights):
        raise RuntimeError("MLP weights of one network pair must share one 16-bit type")
    for w in weights:
        _chk(w, "MLP weights", dt)
    return 1 if dt == torch.bfloat16 else 0


def _color_depth(w2, g_w2=None) -> int:
    """n_hidden_layers_color (model_args.py:31) of a colour network from its stacked hidden-to-hidden matrices: tcnn's flat parameter
    vector holds [w1 | n_hidden - 1 matrices of (64, 64) | w3]; the kernels are built for one, two and three hidden layers."""
    numel = 0 if w2 is None else int(w2.numel())
    if numel % 4096 != 0 or numel > 2 * 4096:
        raise RuntimeError(f"colour network: the hidden-to-hidden weights must be 0, 1 or 2 stacked (64, 64) matrices, got {numel} values")
    if g_w2 is not None and int(g_w2.numel()) != numel:
        raise RuntimeError("colour network: the gradient buffer of the hidden-to-hidden weights must have their shape")
    return numel // 4096 + 1


def density_mlp_fwd(features, w1, w2, density_scale: float, want_h: bool = True, want_sigma: bool = True):
    _chk(features, "features", torch.float16)
    mode = _mlp_mode(w1, w2)
    n = features.shape[0]
    h = _new("h", (n, 16), torch.float16, features.device) if want_h else None
    sigma = _new("sigma", (n,), torch.float32, features.device) if want_sigma else None
    with _span("density_mlp_fwd", n):
        check(_lib.lib().hrf_density_mlp_fwd(ptr(features), ptr(w1), ptr(w2), density_scale, n, ptr(h), ptr(sigma),
                                             mode, stream_ptr()))
    return h, sigma


def color_mlp_fwd(ray_dirs, sample_ray, h, cam_emb, ray_cameras, emb_dim: int, use_emb: bool, w1, w2, w3, geo_dim: int = 15):
    """geo_dim: geometry_feature_dim of the model (sigma_net outputs 1 + geo_dim values; colour input columns
    [SH 16 | geo | embedding | ones], w1 is (64, 16 * ceil((16 + geo_dim + emb_dim) / 16)))."""
    _chk(ray_dirs, "ray_directions", torch.float32); _chk(sample_ray, "ray_indices", torch.int64)
    _chk(h, "h", torch.float16); _chk(cam_emb, "camera_embeddings", torch.float32)
    _chk(ray_cameras, "camera_numbers", torch.int32)
    mode = _mlp_mode(w1, w2, w3)
    n = h.shape[0]
    rgb = _new("rgb", (n, 3), torch.float16, h.device)
    with _span("color_mlp_fwd", n):
        check(_lib.lib().hrf_color_mlp_fwd(ptr(ray_dirs), ptr(sample_ray), ptr(h), ptr(cam_emb), ptr(ray_cameras),
                                           emb_dim, 1 if use_emb else 0, ptr(w1), ptr(w2), ptr(w3), n, ptr(rgb),
                                           mode, int(geo_dim), _color_depth(w2), stream_ptr()))
    return rgb


def mlp_bwd(features, ray_dirs, sample_ray, cam_emb, ray_cameras, emb_dim, use_emb, sw1, sw2, cw1, cw2, cw3,
            density_scale, d_rgb, d_sigma, g_sw1, g_sw2, g_cw1, g_cw2, g_cw3, g_emb, flags, fp32_out: bool = True,
            level_major: bool = False, grad_boundary: float = 0.0, geo_dim: int = 15):
    _chk(d_rgb, "d_rgb", torch.float32); _chk(d_sigma, "d_sigma", torch.float32)
    mode = _mlp_mode(sw1, sw2, cw1, cw2, cw3)
    n = features.shape[0]
    if level_major:
        d_features = _new("d_features", (16, n, 2), torch.float32, features.device)
    else:
        d_features = _new("d_features_rm", (n, 32), torch.float32 if fp32_out else torch.float16, features.device)
    with _span("mlp_bwd", n):
        check(_lib.lib().hrf_mlp_bwd(ptr(features), ptr(ray_dirs), ptr(sample_ray), ptr(cam_emb), ptr(ray_cameras),
                                     emb_dim, 1 if use_emb else 0, ptr(sw1), ptr(sw2), ptr(cw1), ptr(cw2), ptr(cw3),
                                     density_scale, ptr(d_rgb), ptr(d_sigma), n, ptr(d_features), 2 if level_major else (1 if fp32_out else 0),
                                     float(grad_boundary), ptr(g_sw1), ptr(g_sw2),
                                     ptr(g_cw1), ptr(g_cw2), ptr(g_cw3), ptr(g_emb), ptr(flags), mode, int(geo_dim),
                                     _color_depth(cw2, g_cw2), stream_ptr()))
    return d_features


def density_mlp_bwd(features, w1, w2, d_h, g_w1, g_w2, flags, fp32_out: bool = True, level_major: bool = False,
                    grad_boundary: float = 0.0):
    """Backward of sigma_net alone (tcnn.Network): d_h (n,16) fp32 -> d_features (n,32), or (16,n,2) fp32 level-major (what
    the table scatter of the fused training path reads); g_w1 / g_w2 accumulated."""
    _chk(features, "features", torch.float16); _chk(d_h, "d_h", torch.float32); _chk(flags, "flags", torch.int32)
    _chk(g_w1, "g_w1", torch.float32); _chk(g_w2, "g_w2", torch.float32)
    mode = _mlp_mode(w1, w2)
    n = features.shape[0]
    if level_major:
        d_features = _new("d_features", (16, n, 2), torch.float32, features.device)
    else:
        d_features = torch.empty(n, 32, dtype=torch.float32 if fp32_out else torch.float16, device=features.device)
    with _span("mlp_bwd_density", n):
        check(_lib.lib().hrf_density_mlp_bwd(ptr(features), ptr(w1), ptr(w2), ptr(d_h), n, ptr(d_features),
                                             2 if level_major else (1 if fp32_out else 0), float(grad_boundary), ptr(g_w1),
                                             ptr(g_w2), ptr(flags), mode, stream_ptr()))
    return d_features


def color_mlp_bwd(ray_dirs, sample_ray, h, cam_emb, ray_cameras, emb_dim: int, use_emb: bool, w1, w2, w3, d_rgb, g_w1, g_w2,
                  g_w3, g_emb, flags, d_sigma=None, density_scale: float = 1.0, arena: bool = False, geo_dim: int = 15):
    """Backward of color_net alone (tcnn.NetworkWithInputEncoding): d_rgb (n,3) fp32 -> d_h (n,16) fp32 (gradient of the
    geometry input h[:, 1:]; column 0 is zero, or the backward of truncated_exp when d_sigma (n,) is given: d_h is then the
    whole upstream gradient of sigma_net); weight / embedding gradients accumulated."""
    _chk(ray_dirs, "ray_directions", torch.float32); _chk(sample_ray, "ray_indices", torch.int64); _chk(h, "h", torch.float16)
    _chk(cam_emb, "camera_embeddings", torch.float32); _chk(ray_cameras, "camera_numbers", torch.int32)
    _chk(d_rgb, "d_rgb", torch.float32); _chk(flags, "flags", torch.int32); _chk(d_sigma, "d_sigma", torch.float32)
    mode = _mlp_mode(w1, w2, w3)
    n = h.shape[0]
    d_h = _new("d_h", (n, 16), torch.float32, h.device) if arena else torch.empty(n, 16, dtype=torch.float32, device=h.device)
    with _span("mlp_bwd_color", n):
        check(_lib.lib().hrf_color_mlp_bwd(ptr(ray_dirs), ptr(sample_ray), ptr(h), ptr(cam_emb), ptr(ray_cameras), emb_dim,
                                           1 if use_emb else 0, ptr(w1), ptr(w2), ptr(w3), ptr(d_rgb), ptr(d_sigma),
                                           float(density_scale), n, ptr(d_h), ptr(g_w1), ptr(g_w2), ptr(g_w3), ptr(g_emb),
                                           ptr(flags), mode, int(geo_dim), _color_depth(w2, g_w2), stream_ptr()))
    return d_h


def ray_offsets(sample_ray: torch.Tensor, num_rays: int) -> torch.Tensor:
    _chk(sample_ray, "ray_indices", torch.int64)
    out = _new("ray_start", (num_rays + 1,), torch.int32, sample_ray.device)
    check(_lib.lib().hrf_ray_offsets(ptr(sample_ray), sample_ray.numel(), num_rays, ptr(out), stream_ptr()))
    return out


def visibility(alphas, sigma, ray_start, num_rays: int, early_stop_eps: float, alpha_thre: float, step: float = STEP,
               want_kept: bool = False):
    src = alphas if alphas is not None else sigma
    _chk(src, "alphas/sigma", torch.float32); _chk(ray_start, "ray_start", torch.int32)
    vis = torch.empty(src.numel(), dtype=torch.uint8, device=src.device)
    kept = torch.empty(num_rays, dtype=torch.int32, device=src.device) if want_kept else None
    check(_lib.lib().hrf_visibility(ptr(alphas), ptr(sigma), ptr(ray_start), num_rays, step, early_stop_eps,
                                    alpha_thre, ptr(vis), ptr(kept), stream_ptr()))
    return vis, kept


def prune_march(ray_origins, ray_dirs, ray_frames, ray_start, t0, jitter, model, early_stop_eps: float = 1e-4,
                alpha_thre: float = 1e-4, step: float = STEP, want_sigma: bool = False, want_evaluated: bool = False,
                num_rays_dev=None, t_stage=None, segment_affinity: bool = True, ray_len=None, jitter_seed: int = 0,
                totals=None):
    """Fused prune pass -> (t_stage (N0,), sigma_stage | None, ray_cnt (R,), ray_evaluated | None).
    segment_affinity: schedule the rays by temporal segment over the XCDs (same results, better L2 hit rate).
    ray_len / jitter_seed / totals: see hrf_prune_march in include/hrf.h."""
    R, n0 = ray_origins.shape[0], t0.numel()
    dev = t0.device
    if t_stage is None:
        t_stage = torch.empty(n0, dtype=torch.float32, device=dev)
    sigma_stage = torch.empty(n0, dtype=torch.float32, device=dev) if want_sigma else None
    ray_cnt = torch.empty(R, dtype=torch.int32, device=dev)
    ray_eval = torch.empty(R, dtype=torch.int32, device=dev) if want_evaluated else None
    model._refresh_half()
    sw1, sw2 = model._sigma_w()
    order = ray_segment_order(ray_frames, model, num_rays_dev) if segment_affinity and model.num_frames > 1 else None
    with _span("prune_march", n0):
        check(_lib.lib().hrf_prune_march(ptr(ray_origins), ptr(ray_dirs), ptr(ray_frames), ptr(ray_start), ptr(t0),
                                         ptr(jitter), step, early_stop_eps, alpha_thre,
                                         ptr(model.frame_numbers_to_segment_numbers),
                                         ptr(model.frame_numbers_to_normalized_local_frame_numbers),
                                         ptr(model._tables_h), ptr(model.vectors), ptr(model._seg_meta),
                                         model.num_segments, model.vec_res, ptr(sw1), ptr(sw2),
                                         float(model.density_scale), R, ptr(num_rays_dev), n0, ptr(t_stage), ptr(sigma_stage), ptr(ray_cnt),
                                         ptr(ray_eval), ptr(order), ptr(ray_len), int(jitter_seed) & 0xFFFFFFFF,
                                         ptr(totals), _mlp_mode(sw1, sw2), stream_ptr()))
    return t_stage, sigma_stage, ray_cnt, ray_eval


def uniform_fill(seed: int, n: int, device="cuda", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Values 0..n-1 of the counter-based uniform [0,1) stream `seed` (what hrf_prune_march draws in-kernel for
    jitter_seed == seed)."""
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=device)
    check(_lib.lib().hrf_uniform_fill(int(seed) & 0xFFFFFFFF, n, ptr(out), stream_ptr()))
    return out


def ray_segment_order(ray_frames, model, num_rays_dev=None, out=None, workspace=None, by_frame: bool = True):
    """Ray ids (int32) sorted by frame (or, by_frame=False / more than 1024 frames, by temporal segment): the
    march's schedule (see hrf_prune_march). workspace: 2 * keys int32."""
    _chk(ray_frames, "frame_numbers", torch.int32)
    R = ray_frames.numel()
    if by_frame and model.num_frames <= 1024:
        table, keys = model._frame_rank, model.num_frames
    else:
        table, keys = model.frame_numbers_to_segment_numbers, model.num_segments
    if out is None:
        out = torch.empty(R, dtype=torch.int32, device=ray_frames.device)
    if workspace is None:
        workspace = torch.empty(2 * keys, dtype=torch.int32, device=ray_frames.device)
    check(_lib.lib().hrf_ray_segment_order(ptr(ray_frames), ptr(table), R, ptr(num_rays_dev), keys, ptr(workspace),
                                           ptr(out), stream_ptr()))
    return out


def pack_runs(ray_start, ray_cnt, out_offset, t_stage, n_out: int, num_rays_dev=None, ray_base: int = 0, out_t=None,
              out_r=None):
    if out_t is None:
        out_t = torch.empty(n_out, dtype=torch.float32, device=t_stage.device)
        out_r = torch.empty(n_out, dtype=torch.int64, device=t_stage.device)
        if n_out == 0:  # nothing survived: nothing to pack (and no storage to hand to the kernel)
            return out_t, out_r
    check(_lib.lib().hrf_pack_runs(ptr(ray_start), ptr(ray_cnt), ptr(out_offset), ptr(t_stage), ray_cnt.numel(),
                                   ptr(num_rays_dev), ray_base, ptr(out_t), ptr(out_r), stream_ptr()))
    return out_t, out_r


def compact_samples(vis, slot, t, sample_ray, n_out: int):
    out_t = torch.empty(n_out, dtype=torch.float32, device=t.device)
    out_r = torch.empty(n_out, dtype=torch.int64, device=t.device)
    if n_out == 0:
        return out_t, out_r
    check(_lib.lib().hrf_compact_samples(ptr(vis), ptr(slot), ptr(t), ptr(sample_ray), t.numel(), ptr(out_t),
                                         ptr(out_r), stream_ptr()))
    return out_t, out_r


def composite_fwd(sigma, rgb_h, t, ray_start, background, num_rays: int, step: float = STEP):
    _chk(sigma, "sigma", torch.float32); _chk(rgb_h, "radiance", torch.float16); _chk(t, "t", torch.float32)
    _chk(background, "background_rgb", torch.float32)
    color = _new("color", (num_rays, 3), torch.float32, t.device)
    acc = _new("acc", (num_rays, 1), torch.float32, t.device)
    check(_lib.lib().hrf_composite_fwd(ptr(sigma), ptr(rgb_h), ptr(t), ptr(ray_start), ptr(background), num_rays, step,
                                       ptr(color), ptr(acc), stream_ptr()))
    return color, acc


def composite_bwd(sigma, rgb_h, t, ray_start, background, d_color, d_acc, num_rays: int, step: float = STEP):
    _chk(d_color, "d_color", torch.float32); _chk(d_acc, "d_acc", torch.float32)
    n = t.numel()
    # k_composite_bwd writes every sample of every ray, and the runs [ray_start[r], ray_start[r+1]) cover all n samples
    d_sigma = _new("d_sigma", (n,), torch.float32, t.device)
    d_rgb = _new("d_rgb", (n, 3), torch.float32, t.device)
    check(_lib.lib().hrf_composite_bwd(ptr(sigma), ptr(rgb_h), ptr(t), ptr(ray_start), ptr(background), ptr(d_color),
                                       ptr(d_acc), num_rays, step, ptr(d_sigma), ptr(d_rgb), stream_ptr()))
    return d_sigma, d_rgb


def grad_scaler(device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                growth_interval: int = 2000) -> torch.Tensor:
    """Device-resident hrf_grad_scaler (include/hrf.h) with torch.amp.GradScaler's constructor arguments and defaults."""
    rec = _lib.GradScaler(float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval), 0)
    return torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).clone().to(device)


def grad_scaler_state(scaler: torch.Tensor) -> dict:
    """Host copy of a device-resident scaler (one synchronisation): scale, growth_tracker, ..."""
    rec = _lib.GradScaler.from_buffer_copy(bytes(scaler.cpu().numpy()))
    return {"scale": rec.scale, "growth_factor": rec.growth_factor, "backoff_factor": rec.backoff_factor,
            "growth_interval": rec.growth_interval, "growth_tracker": rec.growth_tracker}


def loss_fwd_bwd(color, acc, rgba, background, huber_delta: float, bce_weight: float, grad_scale: float, sums,
                 ray_frames=None, frame_to_segment=None, group_touched=None, scaler=None, norm_rays: int = 0):
    """norm_rays: rays the loss means run over when `color` holds only a piece of the batch (0 = this call's rays)."""
    n = color.shape[0]
    _chk(scaler, "grad scaler", torch.uint8)
    d_color = _new("d_color", (n, 3), torch.float32, color.device)
    d_acc = _new("d_acc", (n, 1), torch.float32, color.device)
    _chk(ray_frames, "frame_numbers", torch.int32); _chk(frame_to_segment, "frame_to_segment", torch.int32)
    _chk(group_touched, "group_touched", torch.int32)
    check(_lib.lib().hrf_loss_fwd_bwd(ptr(color), ptr(acc), ptr(rgba), ptr(background), n, int(norm_rays), huber_delta, bce_weight,
                                      grad_scale, ptr(d_color), ptr(d_acc), ptr(sums), ptr(ray_frames),
                                      ptr(frame_to_segment), ptr(group_touched), ptr(scaler), stream_ptr()))
    return d_color, d_acc


def render_loss_fused(sigma, rgb_h, t, ray_start, background, rgba, num_rays: int, huber_delta: float, bce_weight: float,
                      grad_scale: float, sums, ray_frames=None, frame_to_segment=None, group_touched=None, scaler=None,
                      norm_rays: int = 0, step: float = STEP, want_color: bool = False):
    """composite_fwd + loss_fwd_bwd + composite_bwd in one launch (hrf_render_loss_fused): -> (d_sigma (n,), d_rgb (n,3),
    color | None, acc | None). Bit-identical to the three calls."""
    _chk(sigma, "sigma", torch.float32); _chk(rgb_h, "radiance", torch.float16); _chk(t, "t", torch.float32)
    _chk(background, "background_rgb", torch.float32); _chk(rgba, "rgba", torch.float32); _chk(ray_start, "ray_start", torch.int32)
    _chk(scaler, "grad scaler", torch.uint8); _chk(ray_frames, "frame_numbers", torch.int32)
    _chk(frame_to_segment, "frame_to_segment", torch.int32); _chk(group_touched, "group_touched", torch.int32)
    n, dev = t.numel(), t.device
    d_sigma = _new("d_sigma", (n,), torch.float32, dev)
    d_rgb = _new("d_rgb", (n, 3), torch.float32, dev)
    color = _new("color", (num_rays, 3), torch.float32, dev) if want_color else None
    acc = _new("acc", (num_rays, 1), torch.float32, dev) if want_color else None
    with _span("render_loss", num_rays):
        check(_lib.lib().hrf_render_loss_fused(ptr(sigma), ptr(rgb_h), ptr(t), ptr(ray_start), ptr(background), ptr(rgba), num_rays,
                                               int(norm_rays), step, huber_delta, bce_weight, grad_scale, ptr(scaler), ptr(ray_frames),
                                               ptr(frame_to_segment), ptr(group_touched), ptr(color), ptr(acc), ptr(d_sigma),
                                               ptr(d_rgb), ptr(sums), stream_ptr()))
    return d_sigma, d_rgb, color, acc


def adam_step(param, grad, exp_avg, exp_avg_sq, p16, lr, beta1, beta2, eps, step: int, grad_scale: float, flags):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    with _span("adam", param.numel()):
        check(_lib.lib().hrf_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(p16), param.numel(),
                                       lr, beta1, beta2, eps, bc1, bc2, grad_scale, ptr(flags), stream_ptr()))


def adam_descriptors(entries, device) -> torch.Tensor:
    """Device array of hrf_adam_tensor descriptors from (param, grad, exp_avg, exp_avg_sq, p16 | None, group) tuples
    (tensors or views; their storage must stay alive and in place). A bfloat16 p16 is refreshed as bf16."""
    arr = (_lib.AdamTensor * len(entries))()
    for k, (p, g, m, v, h, grp) in enumerate(entries):
        if p is None:   # a gradient range another rank steps (sharded exchange): zeroed by the launch, nothing else
            _chk(g, "adam tensor", torch.float32, cuda=False)
            arr[k] = _lib.AdamTensor(None, ptr(g), None, None, None, g.numel(), int(grp), 0)
            continue
        for t in (p, g, m, v):
            _chk(t, "adam tensor", torch.float32, cuda=False)   # (a CPU-built engine only records addresses)
        if h is not None and h.dtype not in (torch.float16, torch.bfloat16):
            raise RuntimeError(f"16-bit copy has dtype {h.dtype}")
        _chk(h, "16-bit copy", None, cuda=False)
        arr[k] = _lib.AdamTensor(ptr(p), ptr(g), ptr(m), ptr(v), ptr(h), p.numel(), int(grp),
                                 1 if h is not None and h.dtype == torch.bfloat16 else 0)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
    return raw.to(device)


def adam_workspace(device) -> torch.Tensor:
    return torch.empty(int(_lib.lib().hrf_adam_workspace_bytes()), dtype=torch.uint8, device=device)


def adam_multi(descriptors: torch.Tensor, count: int, num_groups: int, max_elements: int, lr, beta1, beta2, eps,
               grad_scale: float, state: torch.Tensor, workspace: torch.Tensor, scaler=None):
    """torch.optim.Adam step of every touched tensor in one launch; see hrf_adam_multi (include/hrf.h) for `state`.
    scaler: device-resident hrf_grad_scaler (ops.grad_scaler) -- unscale + GradScaler.update() on the device."""
    _chk(state, "adam state", torch.int32); _chk(scaler, "grad scaler", torch.uint8)
    with _span("adam", max_elements):
        check(_lib.lib().hrf_adam_multi(ptr(descriptors), count, num_groups, max_elements, lr, beta1, beta2, eps,
                                        grad_scale, ptr(state), ptr(scaler), ptr(workspace), stream_ptr()))


def compose_forward(xyz_f, xyt_f, yzt_f, xzt_f, vectors, xyzt):
    for a, nm in ((xyz_f, "xyz_features"), (xyt_f, "xyt_features"), (yzt_f, "yzt_features"), (xzt_f, "xzt_features")):
        _chk(a, nm, torch.float16)
    _chk(vectors, "xyzt_vectors", torch.float32); _chk(xyzt, "xyzt_coordinates", torch.float32)
    out = torch.empty_like(xyz_f)
    check(_lib.lib().hrf_compose_fwd(ptr(xyz_f), ptr(xyt_f), ptr(yzt_f), ptr(xzt_f), ptr(vectors), ptr(xyzt),
                                     xyz_f.shape[0], xyz_f.shape[1], vectors.shape[1], ptr(out), stream_ptr()))
    return out


def compose_backward(xyz_f, xyt_f, yzt_f, xzt_f, vectors, xyzt, d_out):
    for a, nm in ((xyz_f, "xyz_features"), (xyt_f, "xyt_features"), (yzt_f, "yzt_features"), (xzt_f, "xzt_features"),
                  (d_out, "d_output_features")):
        _chk(a, nm, torch.float16)
    _chk(vectors, "xyzt_vectors", torch.float32); _chk(xyzt, "xyzt_coordinates", torch.float32)
    d = [torch.empty_like(xyz_f) for _ in range(4)]
    d_vec = torch.zeros_like(vectors)
    check(_lib.lib().hrf_compose_bwd(ptr(xyz_f), ptr(xyt_f), ptr(yzt_f), ptr(xzt_f), ptr(vectors), ptr(xyzt),
                                     ptr(d_out), xyz_f.shape[0], xyz_f.shape[1], vectors.shape[1], ptr(d[0]),
                                     ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d_vec), stream_ptr()))
    return d + [d_vec]
