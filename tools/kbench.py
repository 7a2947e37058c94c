"""Micro-benchmark of the encode backward variants on a realistic (partially trained) batch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanrf_amd._lib as _hl
if os.environ.get("KB_LIB"):      # a tuning variant of the library (make -C humanrf_amd/csrc variant TAG=... EXTRA=-D...)
    _hl.LIB_PATH = os.path.join(ROOT, os.environ["KB_LIB"])
from humanrf_amd import ops
from humanrf_amd.dataset.synthetic import SyntheticDataLoader, SyntheticScene
from humanrf_amd.scene_representation import HumanRF
from humanrf_amd.trainer import TrainEngine
from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
dev = "cuda"
warm = int(os.environ.get("KB_WARM", "300"))
reps = int(os.environ.get("KB_REPS", "5"))
torch.manual_seed(123)
frames = tuple(range(15, 65))
# KB_CACHE=/tmp/kb.pt: the first run (the warm-up training below) leaves the trained parameters and the collected batch there,
# later runs on the same box start from them in seconds -- one file serves every rocprofv3 counter pass of tools/measure.sh kpmc.
cache = os.environ.get("KB_CACHE", "")
state = None
if cache and os.path.exists(cache):
    try:
        state = torch.load(cache, map_location=dev)
    except Exception as e:                      # a torn file: train again
        print("KB_CACHE unreadable (%s): training" % e)
loader = scene = None
if state is None:
    scene = SyntheticScene(frames, num_cameras=160, width=752, height=752, grid_resolution=256, device=dev)
    seg_sizes = compute_adaptive_segment_sizes(scene.occupancy_grid, list(frames), 1.25)
    if os.environ.get("KB_SEGMENTS"):          # e.g. KB_SEGMENTS=50: one 50-frame segment (--partitioning none: 2^18-entry tables)
        seg_sizes = [int(v) for v in os.environ["KB_SEGMENTS"].split(",")]
else:
    seg_sizes = state["seg_sizes"]
model = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2, log2_hashmap_size=19, n_levels=16,
                coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1,
                n_hidden_layers_color=2, sh_degree=4, segment_sizes=tuple(seg_sizes), camera_embedding_dim=2, device=dev)
if state is None:
    loader = SyntheticDataLoader(scene, batch_size=8192, max_buffer_size=200, max_num_frames_per_batch=8, seed=123)
    iter(loader)
    eng = TrainEngine(model, loader)
    for _ in range(warm):
        eng.train_iteration()
        for _ in range(8):          # turn the pool over like bench.py does (a static pool over-fits: shorter rays)
            eng.replace_next()
    ib, st = eng.collect_batch()
    # a pre-prune batch of the size a step marches (k_prune_march mode): compacted rays + their staged samples
    loader.batch_size = int(os.environ.get("KB_MARCH_RAYS", "240000"))
    pb = next(loader)
    march_in = {"origins": pb.ray_origins.contiguous(), "dirs": pb.ray_directions.contiguous(),
                "frames": pb.frame_numbers.reshape(-1).contiguous(), "t0": pb.sample_distances.reshape(-1).contiguous(),
                "ray_start": ops.ray_offsets(pb.ray_indices.contiguous(), pb.num_rays).clone()}
    if cache:
        torch.save({"seg_sizes": list(seg_sizes), "warm": warm,
                    "params": [p.detach().clone() for p in (model.table_params, model.vectors, model.sigma_params, model.color_params,
                                                             model.camera_embeddings.weight)],
                    "batch": {k: getattr(ib, k).clone() for k in ("ray_origins", "ray_directions", "frame_numbers", "camera_numbers",
                                                                   "sample_distances", "ray_indices")},
                    "march_in": march_in}, cache)
else:
    from types import SimpleNamespace
    with torch.no_grad():
        for p, v in zip((model.table_params, model.vectors, model.sigma_params, model.color_params, model.camera_embeddings.weight),
                        state["params"]):
            p.copy_(v)
    model._refresh_half()
    eng = TrainEngine(model, loader=None)
    ib = SimpleNamespace(**state["batch"])
    ib.num_rays, ib.num_samples = ib.ray_origins.shape[0], ib.ray_indices.shape[0]
    march_in = state["march_in"]
    print("KB_CACHE: parameters after %d warm-up steps and their batch loaded from %s" % (state["warm"], cache))
print("batch: rays", ib.num_rays, "samples", ib.num_samples)
m = model
t = ib.sample_distances.reshape(-1).contiguous(); ray_idx = ib.ray_indices.contiguous()
xyzt, seg = ops.query_prep(ib.ray_origins.contiguous(), ib.ray_directions.contiguous(), ib.frame_numbers.reshape(-1).contiguous(), ray_idx, t, None,
                           m.frame_numbers_to_segment_numbers, m.frame_numbers_to_normalized_local_frame_numbers)
feats, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
n = xyzt.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
dY32 = torch.randn(n, 32, device=dev, generator=g) * 1e-3
dYlm = dY32.view(n, 16, 2).permute(1, 0, 2).contiguous()
dY16 = dY32.half()
d_tab = torch.zeros(m.table_params.numel(), device=dev); d_vec = torch.zeros_like(m.vectors)

def timeit(fn, name):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    print("%-28s %.3f ms" % (name, a.elapsed_time(b) / reps))

only = os.environ.get("KB_ONLY", "")
# KB_SHIFTS=8,9,10: (libraries built with -DENC_PHASE) repeat the gather kernels for each clock shift of the level-phase order
shifts = [v for v in os.environ.get("KB_SHIFTS", "").split(",") if v] or [None]
if only in ("", "fwd", "gather"):
    for sh in shifts:
        if sh is not None:
            os.environ["HRF_PHASE_SHIFT_FWD"] = sh
        tag = "" if sh is None else " shift %s" % sh
        timeit(lambda: ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, False), "encode4d_fwd" + tag)
        timeit(lambda: ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True), "encode4d_fwd_save" + tag)
if only in ("", "bwd16"):
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dY16, 1.0, d_tab, d_vec), "encode4d_bwd half")
if only in ("", "bwd32"):
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dY32, 1.0, d_tab, d_vec), "encode4d_bwd fp32")
if only in ("", "bwdlm"):
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, d_tab, None, level_major=True), "bwd tables level-major")
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, None, d_vec, level_major=True), "bwd vectors level-major")
if only in ("", "adam"):
    timeit(lambda: d_tab.zero_(), "memset d_tables")
if only in ("march", "gather"):
    mi = march_in
    tot = torch.zeros(2, dtype=torch.int64, device=dev)
    def _march():      # (the rays are scheduled by frame over the XCDs inside, as the engine's collector does)
        return ops.prune_march(mi["origins"], mi["dirs"], mi["frames"], mi["ray_start"], mi["t0"], None, m, jitter_seed=7, totals=tot)
    _march(); torch.cuda.synchronize()
    pre, encd = (int(v) for v in tot.cpu())
    print("march: rays %d staged %d encoded %d" % (mi["origins"].shape[0], pre, encd))
    for sh in shifts:
        if sh is not None:
            os.environ["HRF_PHASE_SHIFT"] = sh
        timeit(_march, "k_prune_march" + ("" if sh is None else " shift %s" % sh))
if only in ("mlpbwd",):
    sw1, sw2 = m._sigma_w(); cw1, cw2, cw3 = m._color_w()
    E = m.camera_embedding_dim
    emb = m.camera_embeddings.weight.detach() if E > 0 else None
    dirs = ib.ray_directions.contiguous(); cams = ib.camera_numbers.reshape(-1).contiguous()
    d_rgb = torch.randn(n, 3, device=dev, generator=g) * 1e-2; d_sig = torch.randn(n, device=dev, generator=g) * 1e-4
    gr = eng._grads; kin = m.color_in_pad
    timeit(lambda: ops.mlp_bwd(feats, dirs, ray_idx, emb, cams, E, E > 0, sw1, sw2, cw1, cw2, cw3, float(m.density_scale), d_rgb,
                               d_sig, gr[2][:2048], gr[2][2048:], *m.split_color(gr[3]),
                               gr[4] if E > 0 else None, eng.flags, level_major=True), "k_mlp_bwd")
    for t in gr[2:]: t.zero_()
if only in ("scatterprof",):
    # the two kernels of the binned scatter alone, for rocprofv3 passes (tools/measure.sh kpmc): records per sample first
    ws = ops.ScatterWorkspace(n + 1024, m.num_segments, m.max_level_entries, dev)
    gb = float(os.environ.get("KB_GB", "128"))      # the engine's default: fp16 gradient boundaries (grad_boundary = 128)
    call = lambda: ops.encode4d_bwd_tables_binned(xyzt, seg, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, d_tab, ws,
                                                  grad_boundary=gb)
    call(); torch.cuda.synchronize()
    tiles = (ws.samples + 1023) // 1024 + m.num_segments
    al = lambda x: (x + 255) // 256 * 256
    hdr = 2 * al((m.num_segments + 1) * 4) + 3 * al(tiles * 4)
    qm = int(os.environ.get("KB_QMAX", "64"))       # SB_QMAX of the library variant (64 with 8192-entry chunks)
    cnt = ws.buf[hdr:hdr + 16 * 4 * qm * tiles * 4].view(torch.int32).view(16, 4, qm, tiles)
    n_tiles = int(ws.buf[:al((m.num_segments + 1) * 4)].view(torch.int32)[m.num_segments])
    cnt = cnt[..., :n_tiles]
    per_level = cnt.sum(dim=(1, 2, 3)).cpu().tolist()
    print("records per sample: %.1f  (per level: %s)" % (sum(per_level) / n, " ".join("%.1f" % (v / n) for v in per_level)))
    timeit(call, "scatter binned")
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, d_tab, None, level_major=True), "scatter atomic")
if only in ("scattervec",):
    # Where to put the vector half of the backward (k_encode4d_bwd_vectors, bound by memory-side atomics) relative to the binned table
    # scatter: behind it, under the whole of it from the start (what the engine does), or under the accumulate kernel alone (needs a
    # library built with -DSB_MID_EVENT, which records an event between the two kernels).
    import ctypes
    from humanrf_amd import _lib
    ws = ops.ScatterWorkspace(n + 1024, m.num_segments, m.max_level_entries, dev)
    gb = float(os.environ.get("KB_GB", "128"))
    tables = lambda: ops.encode4d_bwd_tables_binned(xyzt, seg, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, d_tab, ws, grad_boundary=gb)
    vecs = lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, None, d_vec, level_major=True)
    side = torch.cuda.Stream(device=dev)
    e0, e1, mid = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
    def serial():
        tables(); vecs()
    def under_all():
        e0.record()
        with torch.cuda.stream(side):
            side.wait_event(e0); vecs(); e1.record()
        tables()
        torch.cuda.current_stream().wait_event(e1)
    timeit(tables, "tables alone")
    timeit(vecs, "vectors alone")
    timeit(serial, "tables, then vectors")
    timeit(under_all, "vectors under emit+accumulate")
    L = _lib.lib()
    if hasattr(L, "hrf_scatter_set_mid_event"):
        mid.record(); torch.cuda.synchronize()
        L.hrf_scatter_set_mid_event.restype = None
        L.hrf_scatter_set_mid_event(ctypes.c_void_p(mid.cuda_event))
        def under_acc():
            tables()                                   # records `mid` behind the emit kernel
            with torch.cuda.stream(side):
                side.wait_event(mid); vecs(); e1.record()
            torch.cuda.current_stream().wait_event(e1)
        timeit(under_acc, "vectors under accumulate only")
        L.hrf_scatter_set_mid_event(ctypes.c_void_p(0))
if only in ("", "scatter"):
    # table-gradient scatter: level-major atomics vs radix partition + LDS accumulation (csrc/scatter.hip), on the
    # frame-ordered batch of the collector and on a batch in draw order
    ws = ops.ScatterWorkspace(n + 1024, m.num_segments, m.max_level_entries, dev)
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, d_tab, None, level_major=True), "scatter atomic (sorted batch)")
    timeit(lambda: ops.encode4d_bwd_tables_binned(xyzt, seg, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, d_tab, ws), "scatter binned (sorted batch)")
    timeit(lambda: ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, m.num_segments, dYlm, 1.0, None, d_vec, level_major=True), "vectors (sorted batch)")
    if loader is None:
        raise SystemExit("(draw-order half needs the loader: run without KB_CACHE)")
    eng.collector.sort_batch = False
    ib2, _ = eng.collect_batch()
    t2 = ib2.sample_distances.reshape(-1).contiguous(); r2 = ib2.ray_indices.contiguous()
    x2, s2 = ops.query_prep(ib2.ray_origins.contiguous(), ib2.ray_directions.contiguous(), ib2.frame_numbers.reshape(-1).contiguous(), r2, t2, None,
                            m.frame_numbers_to_segment_numbers, m.frame_numbers_to_normalized_local_frame_numbers)
    n2 = x2.shape[0]
    print("draw-order batch: rays", ib2.num_rays, "samples", n2)
    dY2 = (torch.randn(16, n2, 2, device=dev, generator=g) * 1e-3).contiguous()
    ws2 = ops.ScatterWorkspace(n2 + 1024, m.num_segments, m.max_level_entries, dev)
    f2, e2 = ops.encode4d_fwd(x2, s2, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
    timeit(lambda: ops.encode4d_fwd(x2, s2, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True), "encode4d_fwd_save (draw order)")
    timeit(lambda: ops.encode4d_bwd(x2, s2, e2, m.vectors.detach(), m._seg_meta, m.num_segments, dY2, 1.0, d_tab, None, level_major=True), "scatter atomic (draw order)")
    timeit(lambda: ops.encode4d_bwd_tables_binned(x2, s2, m.vectors.detach(), m._seg_meta, m.num_segments, dY2, 1.0, d_tab, ws2), "scatter binned (draw order)")
    timeit(lambda: ops.encode4d_bwd(x2, s2, e2, m.vectors.detach(), m._seg_meta, m.num_segments, dY2, 1.0, None, d_vec, level_major=True), "vectors (draw order)")
