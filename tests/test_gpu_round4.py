"""Round-4 additions to the training step, each against the form it replaces or the rule it implements:

  * hrf_color_mlp_bwd + hrf_density_mlp_bwd (two wavefronts per SIMD each) == hrf_mlp_bwd (the fused MLP backward);
  * the vector-gradient scatter on a second stream under the table-gradient scatter == one after the other;
  * gradient_boundaries="fp16" (include/hrf.h grad_boundary): every value that crosses a module boundary of the reference
    (decomposition4d.py:8-39; tcnn modules hand half tensors over at the GradScaler's scale) is half-representable at 1/128 of
    the fused scale, contributions below that floor vanish, and the binned and the atomic table scatter agree in that mode;
  * TrainEngine.state_dict() / load_state_dict() resume a run exactly;
  * a batch that is not laid out by frame goes to the level-major scatter, whatever table_scatter says."""
import pytest
import torch

from tests.util import make_model, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _engine(table_scale=0.05, **kw):
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.trainer import TrainEngine
    scene = small_scene(DEV, G=64, W=96, H=80, frames=tuple(range(15, 27)), num_cameras=8)
    torch.manual_seed(5)
    model = make_model(DEV, (6, 6), tuple(scene.frame_numbers), log2_T=19, emb=2, table_scale=table_scale)
    loader = SyntheticDataLoader(scene, batch_size=2048, max_buffer_size=16, max_num_frames_per_batch=4, seed=4)
    iter(loader)
    eng = TrainEngine(model, loader, samples_max_batch_size=60_000, rays_initial_batch_size=2048, **kw)
    return model, loader, eng


def _grads_of(eng, batch, **attrs):
    """All gradients of one train_step over `batch` with the optimizer launch replaced by a no-op (the fused Adam zeroes the
    gradient buffers), engine attributes set as given."""
    from humanrf_amd import ops
    keep = {k: getattr(eng, k) for k in attrs}
    real_adam = ops.adam_multi
    ops.ARENA = eng._arena
    try:
        for k, v in attrs.items():
            setattr(eng, k, v)
        ops.adam_multi = lambda *a, **k: None
        eng.flat_grad.zero_()
        torch.manual_seed(77)                              # the random background of train_step
        eng.train_step(batch)
        torch.cuda.synchronize()
        return [g.clone() for g in eng._grads]
    finally:
        ops.adam_multi = real_adam
        ops.ARENA = None
        for k, v in keep.items():
            setattr(eng, k, v)


def _close(a, b, what, rel_max=2e-3):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    rel = float((a - b).norm() / (b.norm() + 1e-300))
    assert rel <= rel_max, (what, rel)


def test_split_mlp_backward_and_overlapped_vector_scatter_equal_the_serial_fused_step():
    model, loader, eng = _engine()
    for _ in range(3):
        eng.train_iteration()
    assert eng.found_inf() == 0
    batch, _ = eng.collect_batch()
    assert batch._sorted_by_frame and batch.num_samples > 40_000
    base = _grads_of(eng, batch, mlp_backward="fused", overlap_vector_scatter=False)
    split = _grads_of(eng, batch, mlp_backward="split", overlap_vector_scatter=False)
    over = _grads_of(eng, batch, mlp_backward="fused", overlap_vector_scatter=True)
    names = ("tables", "vectors", "sigma_net", "color_net", "embeddings")
    for nm, a, b, c in zip(names, base, split, over):
        assert float(a.abs().sum()) > 0, nm
        _close(b, a, "split " + nm)          # the same MFMA products; atomics in another order
        _close(c, a, "overlapped " + nm, rel_max=1e-4)
    # the binned table scatter is bit-reproducible for one layout and one d_features: the overlapped step reproduces it
    assert torch.equal(over[0], base[0])


def test_fp16_gradient_boundaries_round_through_half_and_drop_what_the_reference_drops():
    from humanrf_amd import ops
    model, loader, eng = _engine(table_scale=0.2)
    for _ in range(2):
        eng.train_iteration()
    batch, _ = eng.collect_batch()
    m = model
    t = batch.sample_distances.reshape(-1).contiguous(); ray_idx = batch.ray_indices.contiguous()
    frames = batch.frame_numbers.reshape(-1).contiguous()
    xyzt, seg = ops.query_prep(batch.ray_origins.contiguous(), batch.ray_directions.contiguous(), frames, ray_idx, t, None,
                               m.frame_numbers_to_segment_numbers, m.frame_numbers_to_normalized_local_frame_numbers)
    n = xyzt.shape[0]
    vectors = m.vectors.detach()
    feats, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, vectors, m._seg_meta, m.num_segments, True)
    sw1, sw2 = m._sigma_w(); cw1, cw2, cw3 = m._color_w()
    E, kin = m.camera_embedding_dim, m.color_in_pad
    emb = m.camera_embeddings.weight.detach()
    dirs = batch.ray_directions.contiguous(); cams = batch.camera_numbers.reshape(-1).contiguous()
    g = torch.Generator(device=DEV).manual_seed(3)
    # upstream gradients spanning many binades around the half floor at the GradScaler's scale (2^-24 there = 2^-17 here)
    mag = torch.exp2(torch.randint(-30, 4, (n, 1), device=DEV, generator=g).float())
    d_rgb = (torch.randn(n, 3, device=DEV, generator=g) * mag).contiguous()
    d_sig = (torch.randn(n, device=DEV, generator=g) * mag[:, 0] * 1e-3).contiguous()

    def run(gb):
        gs = [torch.zeros(2048, device=DEV), torch.zeros(1024, device=DEV), torch.zeros(64 * kin, device=DEV),
              torch.zeros(4096, device=DEV), torch.zeros(1024, device=DEV), torch.zeros_like(emb)]
        flags = torch.zeros(1, dtype=torch.int32, device=DEV)
        d = ops.mlp_bwd(feats, dirs, ray_idx, emb, cams, E, True, sw1, sw2, cw1, cw2, cw3, float(m.density_scale), d_rgb, d_sig,
                        *gs, flags, level_major=True, grad_boundary=gb).clone()
        return d
    d0, d1 = run(0.0), run(128.0)
    assert torch.equal((d1 / 128.0).half().float() * 128.0, d1)            # half-representable at 1/128 of the fused scale
    assert not torch.equal((d0 / 128.0).half().float() * 128.0, d0)
    big = d0.abs() > 1e-2 * d0.abs().max()
    # values well above the floor: the half rounding of dL/d(sigma_net output) carried through the network + their own
    assert float(((d1 - d0).abs()[big] / d0.abs()[big]).max()) <= 1e-2
    assert int(((d1 == 0) & (d0 != 0)).sum()) > 0                          # values below it are gone
    # table scatter in that mode: binned == atomic, and a d_features below the floor leaves no gradient at all
    ws = ops.ScatterWorkspace(n + 1024, m.num_segments, m.max_level_entries, DEV)
    for dy, expect_zero in ((d1, False), (torch.full_like(d1, 2.0 ** -20), True)):
        a = torch.zeros(m.table_params.numel(), device=DEV); b = torch.zeros_like(a); c = torch.zeros_like(a)
        ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, m._seg_meta, m.num_segments, dy, 1.0, a, ws, grad_boundary=128.0)
        ops.encode4d_bwd(xyzt, seg, enc, vectors, m._seg_meta, m.num_segments, dy, 1.0, b, None, level_major=True, grad_boundary=128.0)
        ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, m._seg_meta, m.num_segments, dy, 1.0, c, ws)
        torch.cuda.synchronize()
        if expect_zero:
            # |v| <= ~0.5 (vectors ~ N(0, 0.1^2)), so |v * dY| / 128 < 2^-25: every per-encoding gradient rounds to zero in half
            assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0 and float(c.abs().max()) > 0.0
        else:
            _close(a, b, "binned vs atomic with fp16 boundaries", rel_max=1e-4)
            assert int(((a == 0) & (c != 0)).sum()) > 0                    # entries only sub-floor contributions reach


def test_engine_state_dict_resumes_a_run_exactly():
    from humanrf_amd import ops
    model, loader, eng = _engine()
    for _ in range(3):
        eng.train_iteration()
    batch, _ = eng.collect_batch()
    sd = eng.state_dict()
    assert set(sd["model"]) == set(model.reference_state_dict()) and sd["step"] == 3
    ops.ARENA = eng._arena
    try:
        torch.manual_seed(9); eng.train_step(batch)
        want = {k: v.clone() for k, v in model.reference_state_dict().items()}
        want_m = [t.clone() for t in eng.exp_avg]
        want_v = [t.clone() for t in eng.exp_avg_sq]
        want_opt, want_scaler, want_sched = eng.opt_state.clone(), eng.scaler.clone(), eng.sched_step
        eng.load_state_dict(sd)
        assert eng.step == 3 and eng.optimizer_steps()[0] == 3
        torch.manual_seed(9); eng.train_step(batch)
    finally:
        ops.ARENA = None
    torch.cuda.synchronize()
    got = model.reference_state_dict()
    # The step is replayed from the restored state on the same batch with the same background. Everything on the tables' path is
    # deterministic -- forward, per-sample MLP backward, binned scatter (64-bit integer sums), fused Adam -- so tables, their
    # moments, the per-group step counts and the scaler must come back BIT-IDENTICAL: a resume that restored stale moments, step
    # counts or scaler state for any segment shows up here. The MLP / embedding / vector gradients are summed with
    # fp32 atomics in arrival order: those parameters agree to the noise of one such sum.
    for k in want:
        if "encoding.params" in k:
            assert torch.equal(got[k], want[k]), k
    assert torch.equal(eng.exp_avg[0], want_m[0]) and torch.equal(eng.exp_avg_sq[0], want_v[0])
    assert torch.equal(eng.opt_state, want_opt) and torch.equal(eng.scaler, want_scaler)
    assert eng.step == 4 and eng.sched_step == want_sched
    for k in ("sigma_net.params", "color_net.params", "camera_embeddings.weight"):
        assert torch.allclose(got[k], want[k], atol=2e-3), k
    for a, b in zip(eng.exp_avg[1:], want_m[1:]):
        assert torch.allclose(a, b, rtol=1e-2, atol=1e-5 * float(b.abs().max()) + 1e-12)


def test_a_batch_not_laid_out_by_frame_goes_to_the_level_major_scatter():
    """collector.sort_batch = False (what pipeline_pieces > 1 sets): the engine must not run the binned scatter on tiles that
    mix temporal segments (every minority sample would take its direct path, ~1000 atomics each)."""
    from humanrf_amd import ops
    model, loader, eng = _engine()
    for _ in range(3):
        eng.train_iteration()
    eng.pipeline_pieces = 2
    assert eng.collector.sort_batch is False
    eng.pipeline_pieces = 1
    assert eng.collector.sort_batch is True
    eng.collector.sort_batch = False
    eng.train_iteration()
    batch, _ = eng.collect_batch()
    assert not batch._sorted_by_frame
    called = []
    real = ops.encode4d_bwd_tables_binned
    ops.encode4d_bwd_tables_binned = lambda *a, **k: called.append(1) or real(*a, **k)
    try:
        unsorted = _grads_of(eng, batch)
        assert not called and eng.scatter_ws is not None
    finally:
        ops.encode4d_bwd_tables_binned = real
    assert float(unsorted[0].abs().sum()) > 0
