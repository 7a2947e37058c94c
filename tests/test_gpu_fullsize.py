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
