"""Size-independent properties at BASELINE.json configs[1] sizes (50 frames -> 7 temporal segments, log2_T 19,
grid 256^3, samples_max_batch_size 640 000), where the CPU oracle cannot follow: exact partition of unity of the hash
encoding, conservation and linearity of the gradient scatter, the fused prune march against the unfused kernel
sequence on millions of samples, sampler idempotence / ordering, compositing bounds."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
FRAMES = tuple(range(15, 65))
SEGMENTS = (6, 6, 6, 12, 6, 6, 12)


@pytest.fixture(scope="module")
def full():
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader, SyntheticScene
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine
    torch.manual_seed(123)
    scene = SyntheticScene(FRAMES, num_cameras=24, width=752, height=752, grid_resolution=256, device=DEV)
    model = HumanRF(density_scale=100, sorted_frame_numbers=FRAMES, n_features_per_level=2, log2_hashmap_size=19, n_levels=16,
                    coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
                    n_hidden_layers_density=1, n_hidden_layers_color=2, sh_degree=4, segment_sizes=SEGMENTS,
                    camera_embedding_dim=2, device=DEV)
    loader = SyntheticDataLoader(scene, batch_size=8192, max_buffer_size=48, max_num_frames_per_batch=8, seed=123)
    iter(loader)
    eng = TrainEngine(model, loader, samples_max_batch_size=640_000, rays_initial_batch_size=8192)
    for _ in range(250):            # sharpen the density so that rays saturate (realistic pruning regime)
        eng.train_iteration()
    torch.cuda.synchronize()
    return scene, model, loader, eng


def _samples(model, n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    xyzt = torch.rand(n, 4, device=DEV, generator=g)
    fr = torch.randint(FRAMES[0], FRAMES[-1] + 1, (n,), device=DEV, generator=g)
    seg = model.frame_numbers_to_segment_numbers[fr].contiguous()
    xyzt[:, 3] = model.frame_numbers_to_normalized_local_frame_numbers[fr]
    return xyzt.contiguous(), seg


def test_partition_of_unity_is_exact_at_full_size(full):
    """Constant tables (0.25) and constant vectors (0.5): every corner weight set sums to one, so all 32 features of
    all 700 k samples, in every segment, level and encoding, must be exactly 4 * 0.25 * 0.5."""
    from humanrf_amd import ops
    _, model, _, _ = full
    tables = torch.full_like(model._tables_h, 0.25)
    vectors = torch.full_like(model.vectors, 0.5)
    xyzt, seg = _samples(model, 700_001, 1)
    feats, enc = ops.encode4d_fwd(xyzt, seg, tables, vectors, model._seg_meta, model.num_segments, True)
    assert torch.all(enc == 0.25) and torch.all(feats == 0.5)
    assert sorted(torch.unique(seg).tolist()) == list(range(len(SEGMENTS)))


def test_gradient_scatter_conserves_and_is_linear(full):
    """sum over the entries of one (segment, encoding, level, feature) of d_tables == sum over the samples of that
    segment of dY * (vector factor of that encoding), because the eight corner weights sum to one; and the scatter is
    linear in dY."""
    from humanrf_amd import ops
    from oracle import hrf_oracle as O
    _, model, _, _ = full
    n = 300_000
    xyzt, seg = _samples(model, n, 2)
    vectors = model.vectors.detach()
    feats, enc = ops.encode4d_fwd(xyzt, seg, model._tables_h, vectors, model._seg_meta, model.num_segments, True)
    g = torch.Generator(device=DEV).manual_seed(5)
    dy1 = torch.randn(16, n, 2, device=DEV, generator=g) * 1e-3
    dy2 = torch.randn(16, n, 2, device=DEV, generator=g) * 1e-3

    def scatter(dy):
        d_tab = torch.zeros(model.table_params.numel(), device=DEV)
        d_vec = torch.zeros_like(vectors)
        ops.encode4d_bwd(xyzt, seg, enc, vectors, model._seg_meta, model.num_segments, dy.contiguous(), 1.0, d_tab, d_vec,
                         level_major=True)
        return d_tab, d_vec
    t1, v1 = scatter(dy1)
    t2, v2 = scatter(dy2)
    t3, v3 = scatter(2.0 * dy1 - 0.5 * dy2)
    for a, b in ((t3, 2.0 * t1 - 0.5 * t2), (v3, 2.0 * v1 - 0.5 * v2)):
        assert float((a - b).norm() / b.norm()) < 1e-5
    # conservation per (segment, encoding, level, feature)
    enc_of_vec = {0: 2, 1: 3, 2: 1, 3: 0}      # vector i multiplies encoding {yzt, xzt, xyt, xyz} (tensor_composition.cu:47-54)
    vec_of_enc = {e: i for i, e in enc_of_vec.items()}
    worst = 0.0
    for s in range(len(SEGMENTS)):
        sel = seg == s
        meta = model._metas_host[s]
        sv = O.vectors_sample(vectors[s].cpu(), xyzt[sel].cpu())          # (4, n_s, 32) interpolated vector rows
        for e in range(4):
            fac = sv[vec_of_enc[e]].to(DEV)                                # (n_s, 32)
            want = (dy1[:, sel, :].permute(1, 0, 2).reshape(-1, 32).double() * fac.double()).sum(0)   # (32,) by feature
            for l in (0, 5, 10, 15):
                lv = meta.levels[l]
                base = (int(meta.table_offset) + e * int(meta.entries) + int(lv.offset)) * 2
                got = t1[base: base + int(lv.size) * 2].view(-1, 2).double().sum(0)
                for f in range(2):
                    w = float(want[2 * l + f])
                    worst = max(worst, abs(float(got[f]) - w) / (abs(w) + 1e-6))
    assert worst < 2e-3, worst


def test_fused_march_equals_unfused_sequence_on_a_full_batch(full):
    """~2.5 M occupancy-surviving samples of one sampler call: the early-terminating march must keep exactly the
    samples the encode -> sigma_net -> visibility sequence keeps."""
    import copy
    import humanrf_amd.volume_rendering as vr
    _, model, loader, _ = full
    loader.batch_size = 80_000
    base = next(loader)
    assert base.num_samples > 1_000_000
    outs = []
    for fused in (True, False):
        vr.FUSED_PRUNE = fused
        ib = copy.copy(base)
        ib.sample_distances = base.sample_distances.clone(); ib.ray_indices = base.ray_indices.clone()
        torch.manual_seed(77)
        vr.prune_samples(ib, model, True)
        outs.append((ib.sample_distances, ib.ray_indices))
    vr.FUSED_PRUNE = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    kept = outs[0][0].numel()
    assert 10_000 < kept < 0.5 * base.num_samples


def test_sampler_is_idempotent_sorted_and_consistent(full):
    scene, _, loader, _ = full
    idx = loader.draw_ray_indices(300_000)
    a, b = loader.sample(idx), loader.sample(idx.clone())
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    origins, dirs, rgba, frames, cams, minmax, mask, t, ray = a
    R = int(mask.sum())
    assert origins.shape[0] == R and 0 < R < idx.numel()
    ray = ray.long()
    assert bool((ray[1:] >= ray[:-1]).all()) and int(ray.max()) < R
    same = ray[1:] == ray[:-1]
    assert bool((t[1:][same] > t[:-1][same]).all())
    assert bool((t >= minmax[ray, 0]).all()) and bool((t < minmax[ray, 1]).all())
    assert bool((minmax[:, 0] < minmax[:, 1]).all())
    assert torch.allclose(dirs.norm(dim=1), torch.ones(R, device=DEV), atol=1e-5)
    # per-ray sample counts never exceed the candidate count (tmax - tmin) / step
    counts = torch.bincount(ray, minlength=R)
    assert bool((counts <= ((minmax[:, 1] - minmax[:, 0]) / 4e-4).int() + 1).all())


def test_compositing_bounds_on_a_training_batch(full):
    from humanrf_amd.volume_rendering import render
    _, model, _, eng = full
    ib, _ = eng.collect_batch()
    assert 0.9 * 640_000 <= ib.num_samples <= 1.1 * 640_000 + 1
    out = render(ib, model, 0.0, False)
    acc = out.weights_sum.detach()
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    col = out.color.detach()
    assert float(col.min()) >= -1e-6 and float(col.max()) <= 1.0 + 1e-5      # convex combination of sigmoid outputs
    white = render(ib, model, 1.0, False).color.detach()
    assert torch.allclose(white - col, (1.0 - acc).expand(-1, 3), atol=2e-6)  # background enters as bg * (1 - acc)


# ------------------------------------------------------------------------------------------------ oracle parity at the benchmark's geometry
@pytest.fixture(scope="module")
def bench_geometry():
    import numpy as np
    from humanrf_amd.dataset.occupancy_grid_native import OccupanyGrid
    from humanrf_amd.dataset.synthetic import SyntheticScene
    scene = SyntheticScene(tuple(range(15, 65)), num_cameras=160, width=752, height=752, grid_resolution=256, device=DEV)
    rng = np.random.RandomState(7)
    cams = rng.choice(160, 6, replace=False)
    frames = rng.choice(scene.frame_numbers, 6, replace=True)
    rgba = torch.stack([scene.render_rgba(int(c), int(f)) for c, f in zip(cams, frames)]).reshape(-1, 4)
    grids = {int(f): scene.occupancy_grid(int(f)) for f in set(frames.tolist())}
    ring = OccupanyGrid(256, len(grids))
    tex = {f: ring.add_grid(g) for f, g in grids.items()}
    grids_np = {f: g.cpu().numpy() for f, g in grids.items()}
    return scene, cams, frames, rgba, grids_np, ring, tex


@pytest.mark.parametrize("mode", ["samples_occupancy", "rays_occupancy", "samples_aabb", "rays_aabb"])
def test_sampler_bit_exact_at_benchmark_geometry(mode, bench_geometry):
    """The four sampler entry points (ray_sampler.cu:80-194) against oracle/sampler_oracle.c at the benchmark's own
    geometry: 256^3 occupancy grids, 752^2 images, the 160-camera rig, 12 000 drawn rays with the light-bloom filter on.
    Bit-exact ray masks, per-ray outputs, sample counts / indices and distances."""
    import numpy as np
    from humanrf_amd.dataset import ray_sampler_native as rs
    from oracle import hrf_oracle as O
    scene, cams, frames, rgba, grids_np, ring, tex = bench_geometry
    B, P = 6, 752 * 752
    idx = torch.randint(0, B * P, (12_000,), generator=torch.Generator().manual_seed(11), dtype=torch.int64)
    light = torch.zeros(B * P, dtype=torch.bool)
    light[idx[::9]] = True
    land = np.array([True, True, False, True, True, True])   # one portrait slot (camera 126 is portrait in the dataset)
    args = (rgba, light.to(DEV), torch.tensor(frames, dtype=torch.int32, device=DEV),
            torch.tensor(cams, dtype=torch.int32, device=DEV),
            torch.tensor([tex[int(f)] for f in frames], dtype=torch.int64, device=DEV), torch.tensor(land, device=DEV),
            idx.to(DEV), scene.all_inverse_krs[cams].contiguous(), scene.all_camera_origins[cams].contiguous(), scene.aabb,
            256, 752, 752, 4e-4, True)
    kind, prune = mode.split("_")
    out = getattr(rs, f"get_{kind}_{prune}_minmax")(*args)
    ref = O.sampler_get_data(rgba.cpu().numpy(), light.numpy(), frames.astype(np.int32), cams.astype(np.int32),
                             [grids_np[int(f)] for f in frames], land, idx.numpy(),
                             scene.all_inverse_krs[cams].cpu().numpy(), scene.all_camera_origins[cams].cpu().numpy(),
                             scene.aabb.cpu().numpy(), 256, 752, 752, 4e-4, True, occupancy=prune == "occupancy",
                             get_samples=kind == "samples")
    assert ref[6].sum() > 500, "degenerate draw"
    for nm, a, b in zip(("origins", "dirs", "rgba", "frames", "cameras", "minmax", "ray_mask", "t", "ray"), out, ref):
        a = a.cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a, b), f"{mode}: {nm} differs from the oracle (bit-exact expected)"
    if kind == "samples":
        assert out[7].numel() > 50_000


def test_single_pass_sample_staging_equals_two_pass(full):
    """hrf_sampler_samples in its single-pass form (slots by candidate count, a prefix of each ray's range filled) holds
    exactly the samples the two-pass form writes densely -- same counts, same distances, ray by ray."""
    from humanrf_amd import _lib, ops
    from humanrf_amd._lib import check, ptr, stream_ptr
    scene, model, loader, eng = full
    L = _lib.lib()
    idx = loader.draw_ray_indices(20_000)
    out = loader.sample(idx)                                     # two-pass, reference-shaped (oracle-checked elsewhere)
    R = out[0].shape[0]
    t_ref, ray_ref = out[7], out[8].long()
    cnt_ref = torch.zeros(R, dtype=torch.int64, device=DEV).index_add_(0, ray_ref, torch.ones_like(ray_ref))
    mm = out[5].contiguous()
    count = ((mm[:, 1] - mm[:, 0]) / 4e-4).to(torch.int32)      # ray_sampler.cu:283-285 (fp32 division, truncation)
    offsets = ops.scan_exclusive(count)
    total = int(offsets[R])
    ridx = idx[out[6]].contiguous()
    kept = torch.empty(R, dtype=torch.int32, device=DEV)
    t0 = torch.full((total,), -1.0, device=DEV)
    W, H = loader.resolution
    check(L.hrf_sampler_samples(ptr(ridx), ptr(loader.grid_texture_objects_cuda), ptr(out[0].contiguous()), ptr(out[1].contiguous()),
                                ptr(mm), ptr(count), ptr(offsets), R, None, W * H, 256, 4e-4, 1, ptr(kept), ptr(t0), None, total,
                                stream_ptr()))
    assert torch.equal(kept.long(), cnt_ref)
    pos = offsets[:R].long()[ray_ref] + (torch.arange(ray_ref.numel(), device=DEV) - ops.scan_exclusive(kept)[:R].long()[ray_ref])
    assert torch.equal(t0[pos], t_ref)
    assert int((t0 >= 0).sum()) == t_ref.numel()                 # nothing else was written


def test_in_kernel_jitter_is_the_exported_stream(full):
    """hrf_prune_march with jitter_seed draws value i of the counter-based stream for staged sample i: bit-identical
    survivors to the same march fed with hrf_uniform_fill's values as an explicit jitter array (which is how the
    oracle comparisons feed the reference's torch.rand_like draw)."""
    from humanrf_amd import ops
    scene, model, loader, eng = full
    ib = next(loader)
    t = ib.sample_distances.reshape(-1).contiguous()
    rs_ = ops.ray_offsets(ib.ray_indices.contiguous(), ib.num_rays)
    args = (ib.ray_origins.contiguous(), ib.ray_directions.contiguous(), ib.frame_numbers.reshape(-1).contiguous(), rs_, t)
    seed = 0xC0FFEE
    u = ops.uniform_fill(seed, t.numel(), DEV)
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 5e-3
    assert float((u * 16777216.0).frac().abs().max()) == 0.0            # 24 random bits, like torch.rand
    assert not torch.equal(u, ops.uniform_fill(seed + 1, t.numel(), DEV))
    a = ops.prune_march(*args, u, model, want_sigma=True)
    b = ops.prune_march(*args, None, model, want_sigma=True, jitter_seed=seed)
    assert torch.equal(a[2], b[2])
    keep = (torch.arange(t.numel(), device=DEV) - rs_[:-1].long()[ib.ray_indices]) < a[2].long()[ib.ray_indices]
    assert torch.equal(a[0][keep], b[0][keep]) and torch.equal(a[1][keep], b[1][keep])
    totals = torch.zeros(2, dtype=torch.int64, device=DEV)
    c = ops.prune_march(*args, None, model, want_evaluated=True, jitter_seed=seed, totals=totals)
    assert int(totals[0]) == t.numel() and int(totals[1]) == int(c[3].sum())


def test_background_replacer_thread_and_resident_capture():
    """The loader's replacer thread (data_loader.py:396-422) refills pool slots from the HBM-resident capture while
    sampler passes run on other streams: after draining, every slot holds exactly the image and tables of the (camera,
    frame) pair the schedule put there, and training iterations run under it."""
    from humanrf_amd.dataset.synthetic import ResidentCapture, SyntheticDataLoader, SyntheticScene
    from humanrf_amd.trainer import TrainEngine
    from tests.util import make_model
    frames = tuple(range(15, 27))
    scene = SyntheticScene(frames, num_cameras=10, width=96, height=80, grid_resolution=64, device=DEV)
    cap = ResidentCapture(scene, list(range(10)), cams_per_call=4)
    assert torch.equal(cap.image(7, 20), scene.render_rgba(7, 20))          # batched rendering == single image
    loader = SyntheticDataLoader(scene, batch_size=1024, max_buffer_size=16, max_num_frames_per_batch=3, seed=5, capture=cap)
    iter(loader)
    model = make_model(DEV, (6, 6), frames, log2_T=15, emb=2)
    eng = TrainEngine(model, loader, samples_max_batch_size=40_000, rays_initial_batch_size=1024)
    loader.start_replacer(replacements_per_tick=3)
    before = loader.replacements
    for _ in range(12):
        st = eng.train_iteration()
    loader.drain_replacer()
    loader.stop_replacer()
    torch.cuda.synchronize()
    assert loader.replacements - before == 36 and not eng.found_inf() and st.num_rays > 0
    fr, cm = loader.frame_numbers_cuda.cpu().tolist(), loader.camera_numbers_cuda.cpu().tolist()
    assert set(fr) == loader.frames_in_pool()
    for slot in range(loader.buffer_size):
        assert torch.equal(loader.pixel_colors[slot], cap.image(cm[slot], fr[slot])), slot
        assert torch.equal(loader.inverse_krs_cuda[slot], scene.all_inverse_krs[cm[slot]])
        assert torch.equal(loader.camera_origins_cuda[slot], scene.all_camera_origins[cm[slot]])
        assert int(loader.grid_texture_objects_cuda[slot]) == loader.frame_to_grid_texture[fr[slot]]
    # frame-synchronous schedule: another rank (own camera order) holds the same frames after the same replacements
    other = SyntheticDataLoader(scene, batch_size=1024, max_buffer_size=16, max_num_frames_per_batch=3, seed=5, camera_seed=99,
                                capture=cap)
    for _ in range(36):
        other.replace_next()
    assert other.frames_in_pool() == loader.frames_in_pool()
    assert other.camera_numbers_cuda.cpu().tolist() != cm


def test_validate_renders_full_images(full):
    """humanrf_amd.inference.validate (Trainer.validate's loop, trainer.py:257-370): ordered pixel ranges, evaluation-mode
    prune + render, ray_masks scatter; the PSNR of the assembled image equals the PSNR over the rendered rays plus the
    background pixels, and a trained model beats an untrained one."""
    from humanrf_amd.inference import validate
    from tests.util import make_model
    scene, model, loader, eng = full
    pf, pc = loader.frame_numbers_cuda.cpu(), loader.camera_numbers_cuda.cpu()
    pair = (int(pc[0]), int(pf[0]))
    res = validate(model, loader, [pair], rays_batch_size=65536, return_images=True)
    img = res["images"][0]
    assert img.shape == (1, 752, 752, 3) and res["psnr"][0] > 12.0
    gt = scene.render_rgba(*pair).float().div(255.0)
    gt_img = (gt[:, :3] * gt[:, 3:4]).view(1, 752, 752, 3)
    sil = gt[:, 3].view(752, 752) > 0
    assert float((img[0][sil] - gt_img[0][sil]).abs().mean()) < 0.2       # the person is there, roughly coloured
    assert float(img[0][~sil].abs().mean()) < 0.05                          # and the background is (nearly) empty
    fresh = make_model(DEV, SEGMENTS, FRAMES, log2_T=19, emb=2)
    assert validate(fresh, loader, [pair], rays_batch_size=65536)["psnr"][0] < res["psnr"][0]


def test_pipelined_pieces_equal_the_single_pass():
    """TrainEngine feeds a batch in four ray-aligned pieces with the gradient scatter of piece k on a second stream while
    piece k+1 is computed (trainer.py's step is one pass): same loss sums, same first moments (= 0.1 x gradient after the
    first step) up to the order of the fp32 atomic sums, same Adam step counts."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.trainer import TrainEngine
    from tests.util import make_model, small_scene
    frames = tuple(range(15, 27))
    scene = small_scene("cuda", G=64, W=128, H=128, frames=frames, num_cameras=12)
    out = {}
    for pieces in (1, 4):
        torch.manual_seed(0)
        m = make_model("cuda", (6, 6), frames, log2_T=15, emb=2, table_scale=0.2)
        loader = SyntheticDataLoader(scene, batch_size=2048, max_buffer_size=12, max_num_frames_per_batch=4, seed=5)
        iter(loader)
        eng = TrainEngine(m, loader, samples_max_batch_size=100_000, rays_initial_batch_size=2048)
        eng.pipeline_pieces, eng.pipeline_min_samples = pieces, 0
        eng.collector.sort_batch = False    # the ray-aligned cut points are those of the batch in draw order
        used = []
        for _ in range(3):       # step 1 runs classic iterations (no prefetched set yet); later steps come with cut points
            torch.manual_seed(100 + len(used))
            batch, _ = eng.collect_batch()
            used.append(len(eng._pieces(batch)))
            eng.loss_sums.zero_()
            eng.train_step(batch)
        torch.cuda.synchronize()
        out[pieces] = (used, eng.loss_sums.cpu(), [t.clone().cpu() for t in eng.exp_avg], eng.optimizer_steps(), eng.found_inf(),
                       batch.num_rays, batch.num_samples)
    assert out[1][0] == [1, 1, 1] and out[4][0][-1] == 4, (out[1][0], out[4][0])
    assert out[1][5:] == out[4][5:] and out[1][3] == out[4][3] and out[1][4] == out[4][4] == 0
    assert torch.allclose(out[1][1], out[4][1], rtol=1e-4)
    for a, b in zip(out[1][2], out[4][2]):
        assert float((a - b).norm() / (a.norm() + 1e-30)) < 2e-3
