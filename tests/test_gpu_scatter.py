"""The binned table-gradient scatter (csrc/scatter.hip: radix partition + LDS accumulation, no memory-side atomics)
against the level-major atomic kernel it replaces in the training step: the same sums
(tcnn kernel_grid_backward x4 + compose backward, decomposition4d.py:79-122, tensor_composition.cu:85-117) in another
order, so equal up to fp32 summation order. Cases: ray-shaped sample runs of one segment per tile (the training
layout), samples of mixed segments inside the tiles (minority samples take the direct path), random positions (every
sample opens eight new corners: queues overflow and spill to the direct path), a model with a dense level 0, a ragged
last tile, bit-reproducibility, the overflow flag, and the frame-ordered batch of the collector."""
import pytest
import torch

from tests.util import make_model, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"
FRAMES = tuple(range(15, 65))
SEGMENTS = (6, 6, 6, 12, 6, 6, 12)


def _bench_model():
    return make_model(DEV, SEGMENTS, FRAMES, log2_T=19, emb=0, table_scale=0.2)


def _ray_samples(model, n_rays, per_ray, seed, sort_by_segment=True, frames=FRAMES):
    """Samples that walk along rays in steps of 4e-4 (the march step), `per_ray` consecutive samples per ray."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    o = torch.rand(n_rays, 3, device=DEV, generator=g) * 0.6 + 0.2
    d = torch.randn(n_rays, 3, device=DEV, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    fr = torch.randint(frames[0], frames[-1] + 1, (n_rays,), device=DEV, generator=g)
    if sort_by_segment:
        fr = fr.sort().values
    k = torch.arange(per_ray, device=DEV, dtype=torch.float32) * 4e-4
    pos = (o[:, None, :] + k[None, :, None] * d[:, None, :]).reshape(-1, 3).clamp(0.0, 1.0)
    frs = fr.repeat_interleave(per_ray)
    xyzt = torch.cat([pos, model.frame_numbers_to_normalized_local_frame_numbers[frs][:, None]], dim=1).contiguous()
    seg = model.frame_numbers_to_segment_numbers[frs].contiguous()
    return xyzt, seg


def _both(model, xyzt, seg, dy, ws=None):
    from humanrf_amd import ops
    vectors = model.vectors.detach()
    n = xyzt.shape[0]
    ref = torch.zeros(model.table_params.numel(), device=DEV)
    enc = torch.zeros(n, 4, 32, dtype=torch.float16, device=DEV)   # the table half does not read it
    ops.encode4d_bwd(xyzt, seg, enc, vectors, model._seg_meta, model.num_segments, dy, 1.0, ref, None, level_major=True)
    out = torch.zeros_like(ref)
    if ws is None:
        ws = ops.ScatterWorkspace(n + 1024, model.num_segments, model.max_level_entries, DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, model._seg_meta, model.num_segments, dy, 1.0, out, ws, flags=flags)
    # sum of |addends| per entry (an upper bound: |interpolated vector| <= interpolation of |vectors|): what the noise of a
    # re-ordered fp32 sum, and a residue where the atomic sum cancelled to exactly zero, is measured against -- per entry, not
    # against the largest gradient of the whole buffer
    abs_sum = torch.zeros_like(ref)
    ops.encode4d_bwd(xyzt, seg, enc, vectors.abs(), model._seg_meta, model.num_segments, dy.abs(), 1.0, abs_sum, None, level_major=True)
    torch.cuda.synchronize()
    assert int(flags) == 0
    _both.last = (ref, abs_sum)
    return ref, out, ws


def _where(model, flat_idx):
    """(segment, encoding, level, entry, feature) of positions of the flat table-gradient buffer -- for failure messages."""
    out = []
    for i in flat_idx.tolist()[:8]:
        e2, off = i // 2, 0
        for s, entries in enumerate(model.entries_per_segment):
            if e2 < off + 4 * entries:
                enc, loc = (e2 - off) // entries, (e2 - off) % entries
                lv = model._metas_host[s].levels
                l = max(k for k in range(16) if int(lv[k].offset) <= loc)
                out.append((s, enc, l, loc - int(lv[l].offset), i % 2, int(lv[l].size), bool(lv[l].hashed)))
                break
            off += 4 * entries
    return out


def _assert_same_sums(ref, out, model=None):
    assert torch.isfinite(out).all()
    scale = float(ref.abs().max())
    assert scale > 0
    err = float((ref - out).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    # every entry, relative to its own size where that is above the noise of the reordered fp32 sums
    big = ref.abs() > 1e-3 * scale
    rel = ((ref - out).abs()[big] / ref.abs()[big]).max()
    assert float(rel) < 2e-3, float(rel)
    # the same entries are touched; sums 38 bits below the largest record of their table are below the fixed-point unit
    # (an entry whose fp32 atomics happened to cancel to exactly zero in `ref` may keep a residue below the noise here)
    last = getattr(_both, "last", None)
    abs_sum = last[1] if last is not None and last[0] is ref else None     # (only for the reference tensor _both just produced)
    if abs_sum is not None:
        # every entry against ITS OWN addends (the noise of a re-ordered fp32 sum), plus the fixed point's absolute floor: records are
        # quantised to 2^-38 of a bound of their table's largest record, a few hundred records per entry -> ~1e-9 of the largest gradient
        bad = ((ref - out).abs() > 1e-5 * abs_sum + 1e-9 * scale).nonzero().reshape(-1)
        assert bad.numel() == 0, (bad.numel(), ref[bad][:6].tolist(), out[bad][:6].tolist(), abs_sum[bad][:6].tolist(), scale)
        extra = ((out != 0) & (ref == 0) & (out.abs() > 4e-6 * abs_sum + 1e-9 * scale)).nonzero().reshape(-1)
    else:
        extra = ((out != 0) & (ref == 0) & (out.abs() > 1e-5 * scale)).nonzero().reshape(-1)
    assert extra.numel() == 0, (extra.numel(), out[extra][:8].tolist(), _where(model, extra) if model is not None else extra[:8].tolist(),
                                "scale", scale)
    lost = (ref != 0) & (out == 0)
    assert int(lost.sum()) <= 1e-3 * int((ref != 0).sum())
    if bool(lost.any()):
        assert float(ref[lost].abs().max()) <= 1e-9 * scale


@pytest.mark.parametrize("n_rays,per_ray", [(20_000, 16), (9_001, 7), (3_000, 64)])
def test_binned_scatter_equals_atomic_on_ray_runs(n_rays, per_ray):
    model = _bench_model()
    assert model.max_level_entries <= 65536
    xyzt, seg = _ray_samples(model, n_rays, per_ray, seed=n_rays)
    n = xyzt.shape[0]
    g = torch.Generator(device=DEV).manual_seed(7)
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref, out, _ = _both(model, xyzt, seg, dy)
    _assert_same_sums(ref, out)


def test_binned_scatter_mixed_segments_and_overflow():
    """Unsorted rays put several temporal segments into every tile (the minority goes through the direct path), and
    random positions make every sample retire eight corners, which overfills the queues of the busiest chunks."""
    model = _bench_model()
    xyzt, seg = _ray_samples(model, 6_000, 16, seed=3, sort_by_segment=False)
    g = torch.Generator(device=DEV).manual_seed(11)
    n = xyzt.shape[0]
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref, out, _ = _both(model, xyzt, seg, dy)
    _assert_same_sums(ref, out)
    # random positions of ONE segment
    n = 50_000
    xyzt = torch.rand(n, 4, device=DEV, generator=g)
    fr = torch.full((n,), FRAMES[20], device=DEV, dtype=torch.long)
    xyzt[:, 3] = model.frame_numbers_to_normalized_local_frame_numbers[fr]
    seg = model.frame_numbers_to_segment_numbers[fr].contiguous()
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref, out, _ = _both(model, xyzt.contiguous(), seg, dy)
    _assert_same_sums(ref, out)


def test_binned_scatter_dense_level_and_workspace_reuse():
    """One 12-frame segment at log2_T 19 -> 2^16 tables with a dense level 0 (35 944 entries: five chunks, the last one
    partial); the same workspace serves batches of different sizes one after the other."""
    frames = tuple(range(15, 27))
    model = make_model(DEV, (12,), frames, log2_T=19, table_scale=0.2)
    assert not model._metas_host[0].levels[0].hashed and model.max_level_entries == 65536
    from humanrf_amd import ops
    ws = ops.ScatterWorkspace(40_000, 1, model.max_level_entries, DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    for n_rays, per_ray in ((2_000, 16), (311, 9), (2_400, 16)):
        o = torch.rand(n_rays, 3, device=DEV, generator=g) * 0.6 + 0.2
        d = torch.randn(n_rays, 3, device=DEV, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        k = torch.arange(per_ray, device=DEV, dtype=torch.float32) * 4e-4
        pos = (o[:, None, :] + k[None, :, None] * d[:, None, :]).reshape(-1, 3)
        n = pos.shape[0]
        xyzt = torch.cat([pos, torch.rand(n_rays, device=DEV, generator=g).repeat_interleave(per_ray)[:, None]], dim=1).contiguous()
        seg = torch.zeros(n, dtype=torch.int32, device=DEV)
        dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
        ref, out, _ = _both(model, xyzt, seg, dy, ws=ws)
        _assert_same_sums(ref, out)
    with pytest.raises(RuntimeError):   # a batch larger than the workspace was sized for is refused, not truncated
        n = 50_000
        _both(model, torch.rand(n, 4, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV),
              torch.zeros(16, n, 2, device=DEV), ws=ws)


def test_binned_scatter_is_bit_reproducible_on_a_segment_sorted_batch():
    """Records are accumulated in 64-bit fixed point (integer sums do not depend on the order of the records) and a batch
    sorted by temporal segment issues no other atomic than one add per touched entry: two runs give identical bits. The
    atomic kernel it replaces does not (fp32 atomics in arrival order)."""
    model = _bench_model()
    xyzt, seg = _ray_samples(model, 8_000, 16, seed=21)
    assert bool((seg[1:] >= seg[:-1]).all())
    n = xyzt.shape[0]
    g = torch.Generator(device=DEV).manual_seed(2)
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref, a, ws = _both(model, xyzt, seg, dy)
    _, b, _ = _both(model, xyzt, seg, dy, ws=ws)
    _, c, _ = _both(model, xyzt, seg, dy)
    _assert_same_sums(ref, a)
    assert torch.equal(a, b) and torch.equal(a, c)


def test_binned_scatter_raises_the_overflow_flag_on_non_finite_records():
    from humanrf_amd import ops
    model = _bench_model()
    xyzt, seg = _ray_samples(model, 500, 16, seed=5)
    n = xyzt.shape[0]
    dy = torch.zeros(16, n, 2, device=DEV)
    dy[3, 100, 0] = float("inf")
    ws = ops.ScatterWorkspace(n + 1024, model.num_segments, model.max_level_entries, DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.zeros(model.table_params.numel(), device=DEV)
    ops.encode4d_bwd_tables_binned(xyzt, seg, model.vectors.detach(), model._seg_meta, model.num_segments, dy, 1.0, out, ws, flags=flags)
    torch.cuda.synchronize()
    assert int(flags) == 1


def _oracle_table_grads(model, xyzt, seg, dy_lm):
    """Table gradients of sum(features * dY) through the CPU oracle's Decomposition4D (oracle/hrf_oracle.py: tcnn grid
    backward x4 + compose backward by autograd), laid out like model.table_params."""
    from oracle import hrf_oracle as O
    from tests.util import oracle_model_from
    om = oracle_model_from(model, requires_grad=True)
    x, sg = xyzt.cpu(), seg.cpu().long()
    dy = dy_lm.permute(1, 0, 2).reshape(-1, 32).cpu()
    for s in torch.unique(sg).tolist():
        sel = (sg == s).nonzero().reshape(-1)
        f = O.decomposition4d(x[sel], om.tables[s], om.vectors[s], om.levels[s])
        (f * dy[sel]).sum().backward()
    zero = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)
    return torch.cat([zero(t).reshape(-1) for s in range(model.num_segments) for t in om.tables[s]])


def _assert_close_to_oracle(out, want, what):
    a, b = out.double().cpu(), want.double()
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
    rel = float((a - b).norm() / (b.norm() + 1e-300))
    assert cos >= 0.99999 and rel <= 1e-3, (what, cos, rel)          # same fp32 products, another summation order
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1e-4 * scale, (what, float((a - b).abs().max()), scale)
    assert int(((a != 0) & (b == 0)).sum()) == 0, what               # no entry the oracle leaves untouched


@pytest.mark.parametrize("segment,entries", [(25, 1 << 17), (50, 1 << 18), (100, 1 << 19)])
def test_binned_scatter_large_tables_equal_atomic_and_oracle(segment, entries):
    """Level tables of 2^17 / 2^18 / 2^19 entries (the 25- / 50- / 100-frame segments of adaptive_temporal_partitioning.py:8
    at log2_hashmap_size 19, humanrf.py:106-109; `--model.temporal_partitioning none|fixed`): 16 / 32 / 64 chunks with the
    interleaved chunk map, including the dense levels above 65 536 entries (res 43 at 2^17, 55 at 2^18, 73 at 2^19: tables whose
    size is not a power of two). Against the atomic kernel on ray runs and on random positions (every sample retires eight
    corners per level and encoding: the 128-record queues of a 64-chunk level overflow into the direct path), and against
    the CPU oracle's autograd directly."""
    frames = tuple(range(15, 15 + segment))
    model = make_model(DEV, (segment,), frames, log2_T=19, emb=0, table_scale=0.2)
    assert model.max_level_entries == entries
    lv = model._metas_host[0].levels
    dense_big = [int(lv[l].size) for l in range(16) if not lv[l].hashed and int(lv[l].size) > 65536]
    assert len(dense_big) > 0 and all(v & (v - 1) for v in dense_big), dense_big    # e.g. 79 512 entries (res 43) at 2^17
    xyzt, seg = _ray_samples(model, 16_000, 16, seed=segment, frames=frames)
    n = xyzt.shape[0]
    g = torch.Generator(device=DEV).manual_seed(7)
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref, out, ws = _both(model, xyzt, seg, dy)
    _assert_same_sums(ref, out, model)
    _, again, _ = _both(model, xyzt, seg, dy, ws=ws)
    # Integer sums: bit-reproducible for one layout -- except where a queue overflowed into the direct fp32 atomics. With 64
    # queues per (tile, level, encoding) a queue holds 128 records for a mean of 60-70 (standard deviation ~14): about one queue
    # in 10^5 spills, and an entry that receives spilled addends next to its accumulated sum depends on their order in the last
    # bit (measured: 10 of 10^7 entries on this batch).
    diff = (out != again).nonzero().reshape(-1)
    assert diff.numel() <= 1e-4 * int((out != 0).sum()), (diff.numel(), _where(model, diff))
    if diff.numel():
        assert float((out[diff] - again[diff]).abs().max()) <= 1e-6 * float(out.abs().max()), (out[diff][:8].tolist(), again[diff][:8].tolist())
    # random positions
    n2 = 40_000
    x2 = torch.rand(n2, 4, device=DEV, generator=g)
    x2[:, 3] = model.frame_numbers_to_normalized_local_frame_numbers[torch.full((n2,), frames[3], device=DEV, dtype=torch.long)]
    s2 = torch.zeros(n2, dtype=torch.int32, device=DEV)
    dy2 = (torch.randn(16, n2, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref2, out2, _ = _both(model, x2.contiguous(), s2, dy2)
    _assert_same_sums(ref2, out2)
    # the oracle, directly, on a prefix of whole tiles and a ragged last one
    m = 1_500 * 16 + 5
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    from humanrf_amd import ops
    got = torch.zeros(model.table_params.numel(), device=DEV)
    dy3 = dy[:, :m].contiguous()
    ops.encode4d_bwd_tables_binned(xyzt[:m].contiguous(), seg[:m].contiguous(), model.vectors.detach(), model._seg_meta, 1, dy3, 1.0,
                                   got, ws, flags=flags)
    torch.cuda.synchronize()
    assert int(flags) == 0
    _assert_close_to_oracle(got, _oracle_table_grads(model, xyzt[:m], seg[:m], dy3), f"2^{entries.bit_length() - 1}")


def test_binned_scatter_against_the_oracle_on_the_bench_model_and_on_142_segments():
    """The kernel that carries the headline, pinned on the CPU oracle's autograd (not on the atomic kernel): the bench's
    7-segment log2_T 19 model, and the 1 000-frame model whose 142 segments make the accumulate kernel walk its list of
    present segments eight at a time."""
    from humanrf_amd import ops
    for name, segs, frames, n_rays, per_ray in (("bench", SEGMENTS, FRAMES, 2_500, 16),
                                                ("142 segments", (6,) * 120 + (12,) * 20 + (25,) * 2, tuple(range(15, 1015)), 4_000, 8)):
        model = make_model(DEV, segs, frames, log2_T=19, emb=0, table_scale=0.2)
        xyzt, seg = _ray_samples(model, n_rays, per_ray, seed=len(segs), frames=frames)
        n = xyzt.shape[0]
        if len(segs) > 100:
            assert int(torch.unique(seg).numel()) > 100
        g = torch.Generator(device=DEV).manual_seed(13)
        dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
        ws = ops.ScatterWorkspace(n + 1024, model.num_segments, model.max_level_entries, DEV)
        flags = torch.zeros(1, dtype=torch.int32, device=DEV)
        got = torch.zeros(model.table_params.numel(), device=DEV)
        ops.encode4d_bwd_tables_binned(xyzt, seg, model.vectors.detach(), model._seg_meta, model.num_segments, dy, 1.0, got, ws,
                                       flags=flags)
        torch.cuda.synchronize()
        assert int(flags) == 0
        _assert_close_to_oracle(got, _oracle_table_grads(model, xyzt, seg, dy), name)


def test_tables_above_2_to_19_are_refused_by_the_binned_entry_point_and_served_by_the_engine():
    from humanrf_amd import ops
    frames = tuple(range(15, 27))
    model = make_model(DEV, (100,), frames, log2_T=20)        # 2^20-entry level tables: beyond the reference's defaults
    assert model.max_level_entries == 1 << 20 and not ops.ScatterWorkspace.supports(model.max_level_entries)
    ws = ops.ScatterWorkspace(2048, 1, 65536, DEV)
    ws.max_level_entries = model.max_level_entries
    with pytest.raises(RuntimeError, match="524288"):
        ops.encode4d_bwd_tables_binned(torch.rand(64, 4, device=DEV), torch.zeros(64, dtype=torch.int32, device=DEV),
                                       model.vectors.detach(), model._seg_meta, 1, torch.zeros(16, 64, 2, device=DEV), 1.0,
                                       torch.zeros(model.table_params.numel(), device=DEV), ws)
    from humanrf_amd.trainer import TrainEngine
    eng = TrainEngine(model, loader=None, samples_max_batch_size=10_000, rays_initial_batch_size=64)
    assert eng.scatter_ws is None          # "auto" falls back to the level-major atomic kernel
    big = make_model(DEV, (100,), frames, log2_T=19)          # 2^19: the largest table the reference's defaults build
    eng = TrainEngine(big, loader=None, samples_max_batch_size=10_000, rays_initial_batch_size=64)
    assert eng.scatter_ws is not None and big.max_level_entries == 1 << 19


def test_training_step_binned_equals_atomic_scatter():
    """The engine's backward over ONE collected batch (laid out by frame) with either scatter: the table gradients, read
    before the optimizer consumes them, agree to summation order."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.trainer import TrainEngine
    from humanrf_amd import ops
    scene = small_scene(DEV, G=64, W=96, H=80, frames=tuple(range(15, 27)), num_cameras=8)
    torch.manual_seed(5)
    model = make_model(DEV, (6, 6), tuple(scene.frame_numbers), log2_T=19, emb=2, table_scale=0.05)
    loader = SyntheticDataLoader(scene, batch_size=2048, max_buffer_size=16, max_num_frames_per_batch=4, seed=4)
    iter(loader)
    eng = TrainEngine(model, loader, samples_max_batch_size=60_000, rays_initial_batch_size=2048, table_scatter="binned")
    assert eng.scatter_ws is not None
    with pytest.raises(ValueError):
        TrainEngine(model, loader, table_scatter="something")
    for _ in range(3):                  # the first batch of a run is collected before any sampler set was prefetched: draw order
        eng.train_iteration()
    assert eng.found_inf() == 0
    batch, _ = eng.collect_batch()
    assert batch._sorted_by_frame and batch.num_samples > 40_000
    rank = model._frame_rank[batch.frame_numbers.reshape(-1).long()]
    assert bool((rank[1:] >= rank[:-1]).all()) and int(torch.unique(rank).numel()) > 1
    grads = {}
    real_adam, ws = ops.adam_multi, eng.scatter_ws
    ops.ARENA = eng._arena
    try:
        ops.adam_multi = lambda *a, **k: None              # keep the gradients
        for mode in ("binned", "atomic"):
            eng.scatter_ws = ws if mode == "binned" else None
            eng.flat_grad.zero_()
            torch.manual_seed(77)                          # the random background of train_step
            eng.train_step(batch)
            torch.cuda.synchronize()
            grads[mode] = eng._grads[0].clone()
    finally:
        ops.adam_multi = real_adam
        ops.ARENA = None
    _assert_same_sums(grads["atomic"], grads["binned"])


def test_vector_gradients_on_a_frame_sorted_batch_equal_a_plain_torch_restatement():
    """compose_tensors_backward's vector part (tensor_composition.cu:97-108) on a batch laid out by frame, where whole
    workgroups tap the same two rows of the TIME vector (the workgroup-level reduction of k_encode4d_bwd_vectors), against
    the formula written with torch.index_add_ in float64."""
    from humanrf_amd import ops
    model = _bench_model()
    with torch.no_grad():
        model.vectors.copy_(torch.randn(model.vectors.shape, generator=torch.Generator().manual_seed(3)).to(DEV) * 0.3)
    xyzt, seg = _ray_samples(model, 9_000, 16, seed=77)
    n = xyzt.shape[0]
    model._refresh_half()
    vectors = model.vectors.detach()
    _, enc = ops.encode4d_fwd(xyzt, seg, model._tables_h, vectors, model._seg_meta, model.num_segments, True)
    g = torch.Generator(device=DEV).manual_seed(4)
    dy = torch.randn(n, 32, device=DEV, generator=g) * 1e-2
    dy_lm = dy.view(n, 16, 2).permute(1, 0, 2).contiguous()
    got = torch.zeros_like(vectors)
    ops.encode4d_bwd(xyzt, seg, enc, vectors, model._seg_meta, model.num_segments, dy_lm, 1.0, None, got, level_major=True)
    got32 = torch.zeros_like(vectors)
    ops.encode4d_bwd(xyzt, seg, enc, vectors, model._seg_meta, model.num_segments, dy.contiguous(), 1.0, None, got32)
    Rv = vectors.shape[2]
    want = torch.zeros(vectors.shape, dtype=torch.float64, device=DEV)
    pair = (2, 3, 1, 0)                                    # vector vi pairs with encoding {yzt, xzt, xyt, xyz} (:47-54)
    encd, dyd = enc.double(), dy.double()
    for vi in range(4):
        coord = xyzt[:, vi] * float(Rv) - 0.5              # fp32, as the kernel forms it (:37-45)
        fl = torch.floor(coord)
        fr = (coord - fl).double().unsqueeze(1)
        c0 = torch.clamp(fl, 0.0, float(Rv - 1)).long()
        c1 = torch.clamp(fl + 1.0, 0.0, float(Rv - 1)).long()
        val = encd[:, pair[vi], :] * dyd
        flat = want.view(-1, 32)
        base = (seg.long() * 4 + vi) * Rv
        flat.index_add_(0, base + c0, val * (1.0 - fr))
        flat.index_add_(0, base + c1, val * fr)
    for name, a in (("level-major", got), ("row-major", got32)):
        err = float((a.double() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()), (name, err, float(want.abs().max()))
        assert float((a[:, 3].double() - want[:, 3]).abs().max()) <= 2e-5 * float(want[:, 3].abs().max()), name
    assert int((want[:, 3].abs().sum(dim=-1) > 0).sum()) <= 4 * len(FRAMES)   # the time vector: two rows per frame


def test_binned_scatter_with_more_segments_in_a_batch_than_accumulate_slots():
    """A 250-frame model (33 temporal segments) and a batch that touches all of them: the accumulate kernel's grid covers
    eight segments at a time and walks the list of present segments (a training batch holds at most eight; a caller of the
    entry point may hand over anything)."""
    frames = tuple(range(15, 265))
    segs = (6,) * 24 + (12,) * 9
    model = make_model(DEV, segs, frames, log2_T=19, emb=0, table_scale=0.2)
    assert model.num_segments == 33 and model.max_level_entries <= 65536
    g = torch.Generator(device=DEV).manual_seed(8)
    n_rays, per_ray = 12_000, 12
    o = torch.rand(n_rays, 3, device=DEV, generator=g) * 0.6 + 0.2
    d = torch.randn(n_rays, 3, device=DEV, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    fr = torch.randint(frames[0], frames[-1] + 1, (n_rays,), device=DEV, generator=g).sort().values
    k = torch.arange(per_ray, device=DEV, dtype=torch.float32) * 4e-4
    pos = (o[:, None, :] + k[None, :, None] * d[:, None, :]).reshape(-1, 3).clamp(0.0, 1.0)
    frs = fr.repeat_interleave(per_ray)
    xyzt = torch.cat([pos, model.frame_numbers_to_normalized_local_frame_numbers[frs][:, None]], dim=1).contiguous()
    seg = model.frame_numbers_to_segment_numbers[frs].contiguous()
    assert int(torch.unique(seg).numel()) == 33
    n = xyzt.shape[0]
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    ref, out, _ = _both(model, xyzt, seg, dy)
    _assert_same_sums(ref, out)


def test_emit_then_accumulate_by_segment_groups_equals_the_combined_call_bit_for_bit():
    """hrf_scatter_emit + hrf_scatter_accumulate over groups of temporal segments (ABI 8: what the data-parallel step issues, one
    group's reduce-scatter under the next group's accumulation) give the table gradients of hrf_encode4d_bwd_tables_binned to the
    bit, however the segments are grouped, including groups that hold no sample of the batch; each call touches its own segments
    only."""
    from humanrf_amd import ops
    model = _bench_model()
    xyzt, seg = _ray_samples(model, 12_000, 16, seed=33)
    keep = (seg != 2) & (seg != 5)                 # two segments without samples in the batch
    xyzt, seg = xyzt[keep].contiguous(), seg[keep].contiguous()
    n = xyzt.shape[0]
    g = torch.Generator(device=DEV).manual_seed(4)
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    vectors = model.vectors.detach()
    ws = ops.ScatterWorkspace(n + 1024, model.num_segments, model.max_level_entries, DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    whole = torch.zeros(model.table_params.numel(), device=DEV)
    ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, model._seg_meta, model.num_segments, dy, 1.0, whole, ws, flags=flags, grad_boundary=128.0)
    S = model.num_segments
    bounds, o = [], 0
    for e in model.entries_per_segment:
        bounds.append((o * 2, (o + 4 * e) * 2))
        o += 4 * e
    for groups in ([[s] for s in range(S)], [[0, 1, 2], [3, 4], [5, 6]], [[0, 1, 2, 3, 4, 5, 6]], [[6], [0, 1, 2, 3, 4, 5]]):
        out = torch.zeros_like(whole)
        ops.scatter_emit(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, out, ws, grad_boundary=128.0)
        assert float(out.abs().sum()) == 0.0                   # (a segment-sorted batch: the emit half writes queues only)
        for grp in groups:
            before = out.clone()
            ops.scatter_accumulate(model._seg_meta, S, out, ws, flags=flags, seg_first=grp[0], seg_count=len(grp))
            a, b = bounds[grp[0]][0], bounds[grp[-1]][1]
            assert torch.equal(out[:a], before[:a]) and torch.equal(out[b:], before[b:])
            assert torch.equal(out[a:b], whole[a:b])
        assert torch.equal(out, whole)
    assert int(flags) == 0 and float(whole.abs().sum()) > 0


def test_one_signalled_accumulate_launch_equals_the_combined_call_and_releases_its_waiters_group_by_group():
    """hrf_scatter_accumulate_signalled (ABI 10): ONE launch over every segment by id gives the sums of the combined call to the bit;
    group_done[g] grows by (segments of g) x hrf_scatter_signals_per_segment per call -- segments without a sample of the batch
    included -- and keeps growing over the calls (the counters are never reset); a stream that waits for a group's total
    (hrf_stream_wait_value64) reads that group's COMPLETE gradients, whichever position the group has in the launch."""
    from humanrf_amd import ops
    if not ops.can_stream_wait_value():
        pytest.skip("hipDeviceAttributeCanUseStreamWaitValue is 0 on this device")
    model = _bench_model()
    xyzt, seg = _ray_samples(model, 12_000, 16, seed=35)
    keep = seg != 4                                 # a segment without samples in the batch still reports
    xyzt, seg = xyzt[keep].contiguous(), seg[keep].contiguous()
    n = xyzt.shape[0]
    g = torch.Generator(device=DEV).manual_seed(6)
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    vectors = model.vectors.detach()
    S = model.num_segments
    ws = ops.ScatterWorkspace(n + 1024, S, model.max_level_entries, DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    whole = torch.zeros(model.table_params.numel(), device=DEV)
    ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, whole, ws, flags=flags, grad_boundary=128.0)
    bounds, o = [], 0
    for e in model.entries_per_segment:
        bounds.append((o * 2, (o + 4 * e) * 2))
        o += 4 * e
    per_seg = ops.scatter_signals_per_segment(ws)
    assert per_seg == 16 * 4 * max(1, -(-model.max_level_entries // 8192))
    done = torch.zeros(8, dtype=torch.int64, device=DEV)
    goal = [0] * 8
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for rnd, groups in enumerate(([[0, 1, 2], [3], [4, 5], [6]], [[0], [1, 2, 3, 4, 5, 6]], [[s] for s in range(S)], [[1, 2], [5, 6]])):
        out = torch.zeros_like(whole)
        snaps = [torch.zeros_like(whole) for _ in groups]
        ops.scatter_emit(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, out, ws, grad_boundary=128.0)
        side.wait_stream(main)
        for gi, grp in enumerate(groups):
            goal[gi] += per_seg * len(grp)
        # The launch first, then the waiters, as the training step does it: HIP multiplexes its streams onto a few hardware queues, and a
        # wait packet that shares a queue with a launch enqueued BEHIND it would never be released (the first form of this test did that
        # and hung at the end of the full suite, where dozens of streams exist). Each waiter copies its group's range the moment the
        # group's count is reached.
        ops.scatter_accumulate_signalled(model._seg_meta, S, out, ws, flags, groups, done)
        with torch.cuda.stream(side):
            for gi, grp in enumerate(groups):
                ops.stream_wait_value64(done, gi, goal[gi])
                a, b = bounds[grp[0]][0], bounds[grp[-1]][1]
                snaps[gi][a:b].copy_(out[a:b])
        torch.cuda.synchronize()
        covered = torch.zeros_like(whole, dtype=torch.bool)
        for gi, grp in enumerate(groups):
            a, b = bounds[grp[0]][0], bounds[grp[-1]][1]
            assert torch.equal(snaps[gi][a:b], whole[a:b]), (rnd, gi)         # complete when the waiter was released
            covered[a:b] = True
        # one launch over every segment by id: segments outside the signalled groups are accumulated as well
        assert torch.equal(out, whole), rnd
        assert done[:len(groups)].tolist() == goal[:len(groups)], (rnd, done.tolist(), goal)
    assert int(flags) == 0
    with pytest.raises(RuntimeError):
        ops.scatter_accumulate_signalled(model._seg_meta, S, out, ws, flags, [[3, 4], [1, 2]], done)      # not ascending


def test_signalled_accumulate_on_a_model_with_more_segments_than_grid_slots():
    """Twelve temporal segments against the accumulate grid's eight segment slots: a slot's workgroups take segment s and s + 8, every
    (workgroup, segment) pair reports exactly once -- also for segments that own no tile of the batch and for ids outside the signalled
    groups (accumulated, not counted) -- and the sums equal the combined call's to the bit."""
    from humanrf_amd import ops
    if not ops.can_stream_wait_value():
        pytest.skip("hipDeviceAttributeCanUseStreamWaitValue is 0 on this device")
    frames = tuple(range(15, 51))
    model = make_model(DEV, (3,) * 12, frames, log2_T=15, emb=0, table_scale=0.2)
    xyzt, seg = _ray_samples(model, 6_000, 16, seed=41, frames=frames)
    keep = (seg != 1) & (seg != 9)
    xyzt, seg = xyzt[keep].contiguous(), seg[keep].contiguous()
    n, S = xyzt.shape[0], model.num_segments
    assert S == 12 and len(torch.unique(seg)) == 10
    g = torch.Generator(device=DEV).manual_seed(8)
    dy = (torch.randn(16, n, 2, device=DEV, generator=g) * 1e-2).contiguous()
    vectors = model.vectors.detach()
    ws = ops.ScatterWorkspace(n + 1024, S, model.max_level_entries, DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    whole = torch.zeros(model.table_params.numel(), device=DEV)
    ops.encode4d_bwd_tables_binned(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, whole, ws, flags=flags, grad_boundary=128.0)
    per_seg = ops.scatter_signals_per_segment(ws)
    done = torch.zeros(8, dtype=torch.int64, device=DEV)
    goal = [0] * 8
    for groups in ([[0, 1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11]], [[1, 2], [8, 9, 10]], [[s] for s in range(4, 12)]):
        out = torch.zeros_like(whole)
        ops.scatter_emit(xyzt, seg, vectors, model._seg_meta, S, dy, 1.0, out, ws, grad_boundary=128.0)
        for gi, grp in enumerate(groups):
            goal[gi] += per_seg * len(grp)
        ops.scatter_accumulate_signalled(model._seg_meta, S, out, ws, flags, groups, done)
        torch.cuda.synchronize()
        assert torch.equal(out, whole)
        assert done[:len(groups)].tolist() == goal[:len(groups)], (done.tolist(), goal)
    assert int(flags) == 0 and float(whole.abs().sum()) > 0
