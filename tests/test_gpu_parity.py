"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact for the sampler and the visibility mask; stated tolerances for floating-point kernels."""
import numpy as np
import pytest
import torch

from oracle import hrf_oracle as O
from tests.util import make_model, oracle_model_from, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ------------------------------------------------------------------------------------------ sampler
def _sampler_inputs(scene, n_slots=5, seed=0, landscape_flags=None):
    rng = np.random.RandomState(seed)
    cams = rng.choice(len(scene.cameras), n_slots, replace=True)
    frames = rng.choice(scene.frame_numbers, n_slots, replace=True)
    P = scene.width * scene.height
    rgba = torch.stack([scene.render_rgba(int(c), int(f)) for c, f in zip(cams, frames)]).reshape(-1, 4)
    grids = {int(f): scene.occupancy_grid(int(f)) for f in set(frames.tolist())}
    return cams, frames, rgba, grids, P


@pytest.mark.parametrize("mode,G", [("samples_occupancy", 64), ("rays_occupancy", 64), ("samples_aabb", 64), ("rays_aabb", 64),
                                    # march step 0.5/G not a power of two: every fp32 step of the march rounds
                                    ("samples_occupancy", 60), ("samples_occupancy", 50), ("rays_occupancy", 112)])
def test_sampler_bit_exact(mode, G):
    from humanrf_amd.dataset import ray_sampler_native as rs
    from humanrf_amd.dataset.occupancy_grid_native import OccupanyGrid
    scene = small_scene(DEV, G=G)
    cams, frames, rgba, grids, P = _sampler_inputs(scene)
    B = len(cams)
    ring = OccupanyGrid(scene.grid_resolution, len(grids))
    tex = {f: ring.add_grid(g) for f, g in grids.items()}
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, B * P, (3000,), generator=g, dtype=torch.int64)
    light = torch.zeros(B * P, dtype=torch.bool)
    light[idx[::7]] = True
    W, H = scene.width, scene.height
    args = (rgba, light.to(DEV), torch.tensor(frames, dtype=torch.int32, device=DEV),
            torch.tensor(cams, dtype=torch.int32, device=DEV),
            torch.tensor([tex[int(f)] for f in frames], dtype=torch.int64, device=DEV),
            torch.ones(B, dtype=torch.bool, device=DEV), idx.to(DEV), scene.all_inverse_krs[cams].contiguous(),
            scene.all_camera_origins[cams].contiguous(), scene.aabb, scene.grid_resolution, W, H, 4e-4, True)
    out = getattr(rs, f"get_{mode.split('_')[0]}_{mode.split('_')[1]}_minmax")(*args)
    ref = O.sampler_get_data(
        rgba.cpu().numpy(), light.numpy(), frames.astype(np.int32), cams.astype(np.int32),
        [grids[int(f)].cpu().numpy() for f in frames], np.ones(B, bool), idx.numpy(),
        scene.all_inverse_krs[cams].cpu().numpy(), scene.all_camera_origins[cams].cpu().numpy(),
        scene.aabb.cpu().numpy(), scene.grid_resolution, W, H, 4e-4, True,
        occupancy="occupancy" in mode, get_samples="samples" in mode)
    names = ["origins", "dirs", "rgba", "frames", "cameras", "minmax", "ray_mask", "t", "ray"]
    assert ref[6].sum() > 100, "degenerate test scene"
    for nm, a, b in zip(names, out, ref):
        a = a.cpu().numpy()
        assert a.shape == b.shape, (nm, a.shape, b.shape)
        assert np.array_equal(a, b), f"{nm} differs from the oracle (bit-exact expected)"
    if "samples" in mode:
        assert out[7].numel() > 1000 and np.all(np.diff(out[8].cpu().numpy()) >= 0)


def test_sampler_portrait_and_cpu_pool():
    """landscape_modes == False swaps width/height (ray_sampler.cu:105-110); CPU pools follow the reference path."""
    from humanrf_amd.dataset import ray_sampler_native as rs
    from humanrf_amd.dataset.occupancy_grid_native import OccupanyGrid
    scene = small_scene(DEV)
    cams, frames, rgba, grids, P = _sampler_inputs(scene, n_slots=3, seed=3)
    B = len(cams)
    ring = OccupanyGrid(scene.grid_resolution, len(grids))
    tex = {f: ring.add_grid(g) for f, g in grids.items()}
    idx = torch.randint(0, B * P, (1500,), generator=torch.Generator().manual_seed(9), dtype=torch.int64)
    land = np.array([True, False, True])
    W, H = scene.width, scene.height
    out = rs.get_samples_occupancy_minmax(
        rgba.cpu(), torch.zeros(B * P, dtype=torch.bool), torch.tensor(frames, dtype=torch.int32, device=DEV),
        torch.tensor(cams, dtype=torch.int32, device=DEV),
        torch.tensor([tex[int(f)] for f in frames], dtype=torch.int64, device=DEV),
        torch.tensor(land, device=DEV), idx.to(DEV), scene.all_inverse_krs[cams].contiguous(),
        scene.all_camera_origins[cams].contiguous(), scene.aabb, scene.grid_resolution, W, H, 4e-4, False)
    ref = O.sampler_get_data(rgba.cpu().numpy(), None, frames.astype(np.int32), cams.astype(np.int32),
                             [grids[int(f)].cpu().numpy() for f in frames], land, idx.numpy(),
                             scene.all_inverse_krs[cams].cpu().numpy(), scene.all_camera_origins[cams].cpu().numpy(),
                             scene.aabb.cpu().numpy(), scene.grid_resolution, W, H, 4e-4, False, True, True)
    for a, b in zip(out, ref):
        assert np.array_equal(a.cpu().numpy(), b)


def test_sampler_errors_and_empty():
    from humanrf_amd.dataset import ray_sampler_native as rs
    from humanrf_amd.dataset.occupancy_grid_native import OccupanyGrid
    ring = OccupanyGrid(16, 2)
    with pytest.raises(RuntimeError, match="correct resolution"):
        ring.add_grid(torch.zeros(8, 8, 8, dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError, match="expected device"):
        ring.add_grid(torch.zeros(16, 16, 16, dtype=torch.uint8))
    h0 = ring.add_grid(torch.zeros(16, 16, 16, dtype=torch.uint8, device=DEV))
    h1 = ring.add_grid(torch.zeros(16, 16, 16, dtype=torch.uint8, device=DEV))
    assert ring.add_grid(torch.zeros(16, 16, 16, dtype=torch.uint8, device=DEV)) == h0 != h1  # ring reuse
    # empty grid -> no rays, no samples; zero requested rays -> empty outputs
    scene = small_scene(DEV)
    P = scene.width * scene.height
    G = scene.grid_resolution
    empty_ring = OccupanyGrid(G, 1)
    empty_tex = empty_ring.add_grid(torch.zeros(G, G, G, dtype=torch.uint8, device=DEV))
    for n in (64, 0):
        out = rs.get_samples_occupancy_minmax(
            torch.zeros(P, 4, dtype=torch.uint8, device=DEV), torch.zeros(P, dtype=torch.bool, device=DEV),
            torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV),
            torch.tensor([empty_tex], dtype=torch.int64, device=DEV),
            torch.ones(1, dtype=torch.bool, device=DEV), torch.arange(n, dtype=torch.int64, device=DEV),
            scene.all_inverse_krs[:1].contiguous(), scene.all_camera_origins[:1].contiguous(), scene.aabb,
            scene.grid_resolution, scene.width, scene.height, 4e-4, False)
        assert out[0].shape[0] == 0 and out[7].numel() == 0 and out[6].numel() == n and not out[6].any()


# ------------------------------------------------------------------------------------------ encoding
def _random_queries(model, n, seed=0, frames=None):
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 3, generator=g) - 0.5
    fr = torch.tensor(frames if frames is not None else [15], dtype=torch.int32)
    fn = fr[torch.randint(0, fr.numel(), (n,), generator=g)].reshape(-1, 1)
    return pos, fn


@pytest.mark.parametrize("log2_T,sizes,frames", [(15, (6, 12, 6), tuple(range(15, 39))), (19, (100,), tuple(range(15, 115)))])
def test_shared_level_body_equals_the_plain_hashgrid_kernel_bit_for_bit(log2_T, sizes, frames):
    """The level body of the fused gather kernels (enc_level_shared: byte-offset hashing, head-lane cell sharing through
    ds_bpermute, packed corner weights; march + k_encode4d_fwd) against hrf_hashgrid_fwd (one thread per (sample, level), eight
    plain gathers, enc_corners + enc_gather): the four per-encoding outputs must be the same halves, on hashed and dense
    levels (2^19: four dense levels up to res 74), on samples along rays (shared cells) and on scattered ones (none), for
    batches of mixed segments (per-lane segment path of k_encode4d_fwd) and of one segment (scalar path)."""
    from humanrf_amd import _lib, ops
    from humanrf_amd._lib import check, ptr, stream_ptr
    m = make_model(DEV, sizes, frames, log2_T=log2_T, table_scale=0.5)
    g = torch.Generator().manual_seed(5)
    n_rays, per_ray = 96, 48
    o = torch.rand(n_rays, 1, 3, generator=g) - 0.5
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, generator=g), dim=-1)
    t = (torch.arange(per_ray).float() * 4e-4).view(1, -1, 1)
    along = (o + d * t * 3.0).reshape(-1, 3).clamp(-0.5, 0.5)        # consecutive samples of rays: neighbours share cells
    scattered = torch.rand(1500, 3, generator=g) - 0.5
    pos = torch.cat([along, scattered, torch.tensor([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5], [0.5004, 0.2, -0.5003]])])
    fr = torch.tensor(frames, dtype=torch.int32)
    for mixed in (False, True):
        if mixed:   # frames change from sample to sample: wavefronts straddle segments
            fn = fr[torch.randint(0, fr.numel(), (pos.shape[0],), generator=g)].reshape(-1, 1)
        else:       # frame-ordered like a training batch
            fn = fr[(torch.arange(pos.shape[0]) * fr.numel()) // pos.shape[0]].reshape(-1, 1)
        xyzt, seg = m._xyzt_seg(pos.to(DEV), fn.to(DEV))
        _, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
        axes = ((0, 1, 2), (0, 1, 3), (1, 2, 3), (0, 2, 3))   # decomposition4d.py:126-129
        meta_stride = ctypes_sizeof_segment_meta()
        for s_idx in range(m.num_segments):
            rows = (seg == s_idx).nonzero().reshape(-1)
            if rows.numel() == 0:
                continue
            sm = m._metas_host[s_idx]
            for e in range(4):
                x = xyzt[rows][:, list(axes[e])].contiguous()
                table = m._tables_h[2 * (int(sm.table_offset) + e * int(sm.entries)):]
                out = torch.empty(rows.numel(), 32, dtype=torch.float16, device=DEV)
                check(_lib.lib().hrf_hashgrid_fwd(ptr(x), ptr(table), ptr(m._seg_meta[s_idx * meta_stride:]), 16, rows.numel(),
                                                  ptr(out), stream_ptr()))
                got = enc[rows, e].reshape(rows.numel(), 32)
                assert torch.equal(got.view(torch.int16), out.view(torch.int16)), (mixed, s_idx, e)


def ctypes_sizeof_segment_meta():
    import ctypes
    from humanrf_amd._lib import SegmentMeta
    return ctypes.sizeof(SegmentMeta)


@pytest.mark.parametrize("segments", [((12,), tuple(range(15, 27))), ((6, 12, 6), tuple(range(15, 39)))])
def test_encode4d_forward(segments):
    from humanrf_amd import ops
    sizes, frames = segments
    m = make_model(DEV, sizes, frames, log2_T=17, table_scale=0.5)
    om = oracle_model_from(m)
    pos, fn = _random_queries(m, 3001, frames=list(frames))
    # include the corners of the domain and a point slightly outside (samples can overshoot tmax by one step)
    pos[0] = torch.tensor([-0.5, -0.5, -0.5]); pos[1] = torch.tensor([0.5, 0.5, 0.5]); pos[2] = torch.tensor([0.5004, 0.2, -0.5003])
    xyzt, seg = m._xyzt_seg(pos.to(DEV), fn.to(DEV))
    feats, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
    ref = O.model_features(om, pos, fn)
    err = (feats.float().cpu() - ref).abs()
    # one half ulp of slack on top of fp32 reassociation: |x| <= 4 here -> ulp(fp16) <= 2^-8
    assert float(err.max()) <= 2 ** -8 + 1e-3 * float(ref.abs().max()), float(err.max())
    assert float((err > 0).float().mean()) < 0.05  # almost everything is bit-identical after half rounding
    # the per-encoding outputs the backward consumes
    s0 = (seg == 0).nonzero().reshape(-1)[:500]
    x0 = xyzt[s0].cpu()
    ref_xyz = O.hashgrid_encode(x0[:, [0, 1, 2]], om.tables[0][0], om.levels[0])
    ref_xzt = O.hashgrid_encode(x0[:, [0, 2, 3]], om.tables[0][3], om.levels[0])
    assert float((enc[s0, 0].float().cpu() - ref_xyz).abs().max()) <= 2 ** -9
    assert float((enc[s0, 3].float().cpu() - ref_xzt).abs().max()) <= 2 ** -9


def test_compose_op_matches_reference_signature():
    from humanrf_amd.scene_representation import tensor_composition_native as tc
    N, Rv = 777, 128
    g = torch.Generator().manual_seed(1)
    f = [torch.randn(N, 32, generator=g).half() for _ in range(4)]
    vec = torch.randn(4, Rv, 32, generator=g)
    xyzt = torch.rand(N, 4, generator=g)
    dy = torch.randn(N, 32, generator=g).half()
    out = tc.compose_tensors_forward(*[x.to(DEV) for x in f], vec.to(DEV), xyzt.to(DEV))
    ref = O.compose_tensors(*[x.float() for x in f], vec, xyzt)
    assert float((out.float().cpu() - ref).abs().max()) <= 2 ** -7  # |out| <~ 4: one fp16 ulp
    grads = tc.compose_tensors_backward(*[x.to(DEV) for x in f], vec.to(DEV), xyzt.to(DEV), dy.to(DEV))
    fr = [x.float().requires_grad_() for x in f]
    vr = vec.clone().requires_grad_()
    sv = O.vectors_sample(vr, xyzt)
    (fr[0] * sv[3] + fr[1] * sv[2] + fr[2] * sv[0] + fr[3] * sv[1]).backward(dy.float())
    for k in range(4):
        assert float((grads[k].float().cpu() - fr[k].grad).abs().max()) <= 2 ** -6
    assert torch.allclose(grads[4].cpu(), vr.grad, rtol=1e-3, atol=1e-3)
    with pytest.raises(RuntimeError, match="expected device"):
        tc.compose_tensors_forward(*f, vec, xyzt)


# ------------------------------------------------------------------------------------------ MLPs
@pytest.mark.parametrize("emb", [0, 2])
def test_density_and_color_forward(emb):
    from humanrf_amd.scene_representation import QueryInput
    m = make_model(DEV, (12,), tuple(range(15, 27)), log2_T=16, emb=emb, table_scale=0.5)
    om = oracle_model_from(m)
    n = 2049  # not a multiple of the 16-sample tile
    pos, fn = _random_queries(m, n, seed=2, frames=list(range(15, 27)))
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    cams = torch.randint(0, 160, (n, 1), generator=g, dtype=torch.int32)
    for training in (True, False):
        with torch.no_grad():
            q = m(QueryInput(is_training=training, positions=pos.to(DEV), directions=d.to(DEV), frame_numbers=fn.to(DEV),
                             camera_numbers=cams.to(DEV)))
        with torch.no_grad():
            sig_ref, geo_ref, feats_ref = O.model_density(om, pos, fn)
            _, rgb_ref = O.model_forward(om, pos, d, fn, cams, training)
        # h is a half tensor: compare pre-activations within 2 fp16 ulps of their magnitude, sigma relatively
        assert torch.allclose(q.geometry_features.float().cpu(), geo_ref, atol=4e-3, rtol=4e-3)
        assert torch.allclose(q.density.cpu(), sig_ref, rtol=2e-2, atol=1e-3)
        assert float((q.radiance.cpu() - rgb_ref).abs().max()) <= 4e-3  # stated tolerance on RGB
    qd = m.density(QueryInput(is_training=False, positions=pos.to(DEV), frame_numbers=fn.to(DEV)))
    assert torch.allclose(qd.density.cpu(), sig_ref, rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("emb", [0, 2])
def test_field_backward_against_oracle_autograd(emb):
    from humanrf_amd.scene_representation import QueryInput
    m = make_model(DEV, (6, 6), tuple(range(15, 27)), log2_T=15, emb=emb, table_scale=0.5)
    om = oracle_model_from(m, requires_grad=True)
    n = 1500
    pos, fn = _random_queries(m, n, seed=4, frames=list(range(15, 27)))
    g = torch.Generator().manual_seed(5)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    cams = torch.randint(0, 8, (n, 1), generator=g, dtype=torch.int32)
    w_sig = torch.randn(n, generator=g) * 1e-4
    w_rgb = torch.randn(n, 3, generator=g)
    q = m(QueryInput(is_training=True, positions=pos.to(DEV), directions=d.to(DEV), frame_numbers=fn.to(DEV),
                     camera_numbers=cams.to(DEV)))
    ((q.density * w_sig.to(DEV)).sum() + (q.radiance * w_rgb.to(DEV)).sum()).backward()
    sig, rgb = O.model_forward(om, pos, d, fn, cams, True)
    ((sig * w_sig).sum() + (rgb * w_rgb).sum()).backward()

    def close(a, b, name, cos_min=0.999, rel_max=1e-2):   # measured <= 5.0e-4 (profiles/r06_gradient_parity_measured.txt)
        a, b = a.double().reshape(-1).cpu(), b.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        from tests.util import record_parity
        record_parity(f"test_field_backward_against_oracle_autograd[{emb}]", name, rel, cos, rel_max)
        assert cos >= cos_min and rel <= rel_max, (name, cos, rel)

    close(m.sigma_params.grad, torch.cat([w.grad.reshape(-1) for w in om.sigma_w]), "sigma_net")
    close(m.color_params.grad, torch.cat([w.grad.reshape(-1) for w in om.color_w]), "color_net")
    close(m.vectors.grad, torch.stack([v.grad for v in om.vectors]), "vectors")
    close(m.table_params.grad, torch.cat([t.grad.reshape(-1) for seg in om.tables for t in seg]), "tables")
    if emb:
        close(m.camera_embeddings.weight.grad, om.camera_embeddings.grad, "camera_embeddings")


# ------------------------------------------------------------------------------------------ rendering
def _ragged_rays(n_rays, seed, max_len=150):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, max_len, (n_rays,), generator=g)
    lens[0] = 0; lens[1] = 1; lens[2] = 64; lens[3] = 65; lens[4] = 300; lens[-1] = 0
    ray = torch.repeat_interleave(torch.arange(n_rays), lens)
    return ray, lens


def test_visibility_bit_exact_and_compaction():
    from humanrf_amd import ops
    ray, lens = _ragged_rays(257, 0)
    g = torch.Generator().manual_seed(1)
    alphas = torch.rand(ray.numel(), generator=g) * 0.12
    alphas[torch.rand(ray.numel(), generator=g) < 0.2] = 5e-5   # below alpha_thre
    ref = O.render_visibility(alphas, ray, 1e-4, 1e-4)
    rs = ops.ray_offsets(ray.to(DEV), 257)
    assert torch.equal(rs.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)]))
    vis, kept = ops.visibility(alphas.to(DEV), None, rs, 257, 1e-4, 1e-4, want_kept=True)
    assert torch.equal(vis.cpu().bool(), ref)
    assert torch.equal(kept.cpu().long(), torch.zeros(257, dtype=torch.long).index_add(0, ray, ref.long()))
    slot = ops.scan_exclusive(vis)
    t = torch.rand(ray.numel(), generator=g)
    nt, nr = ops.compact_samples(vis, slot, t.to(DEV), ray.to(DEV), int(slot[-1]))
    assert torch.equal(nt.cpu(), t[ref]) and torch.equal(nr.cpu(), ray[ref])


def test_scan_many_chunks_reused_workspace_u8_and_unaligned_views():
    """hrf_scan_exclusive's multi-workgroup path: lengths around the chunk size and up to thousands of chunks, int32 and uint8 inputs,
    the SAME workspace call after call without clearing, garbage in a fresh workspace, and views that are not 16-byte aligned."""
    from humanrf_amd import _lib
    from humanrf_amd._lib import check, ptr, stream_ptr
    g = torch.Generator().manual_seed(11)
    n_max = 6_000_003
    ws = torch.randint(-2 ** 31, 2 ** 31 - 1, (2 * ((n_max + 4095) // 4096) + 8,), dtype=torch.int32, generator=g).to(DEV)   # garbage
    for rep in range(3):
        for n in (8193, 12288, 12289, 4096 * 64, 4096 * 64 + 1, 1_000_003, n_max):
            for u8 in (False, True):
                x = torch.randint(0, 256 if u8 else 700, (n,), dtype=torch.uint8 if u8 else torch.int32, generator=g)
                xd = x.to(DEV)
                out = torch.empty(n + 1, dtype=torch.int32, device=DEV)
                check(_lib.lib().hrf_scan_exclusive(ptr(xd), 1 if u8 else 0, n, ptr(out), ptr(ws), stream_ptr()))
                want = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(x.to(torch.int64), 0)])
                assert torch.equal(out.cpu().to(torch.int64), want), (rep, n, u8)
    # unaligned views of input and output
    n = 50_001
    x = torch.randint(0, 9, (n + 3,), dtype=torch.int32, generator=g).to(DEV)
    out = torch.empty(n + 5, dtype=torch.int32, device=DEV)
    check(_lib.lib().hrf_scan_exclusive(ptr(x[3:]), 0, n, ptr(out[1:]), ptr(ws), stream_ptr()))
    want = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(x[3:].cpu().to(torch.int64), 0)])
    assert torch.equal(out[1:n + 2].cpu().to(torch.int64), want)


def test_scan_exclusive_long():
    from humanrf_amd import ops
    x = torch.randint(0, 9, (100_003,), dtype=torch.int32)
    out = ops.scan_exclusive(x.to(DEV)).cpu()
    assert torch.equal(out[1:].long(), torch.cumsum(x.long(), 0)) and int(out[0]) == 0
    assert torch.equal(ops.scan_exclusive(torch.zeros(0, dtype=torch.int32, device=DEV)).cpu(), torch.zeros(1, dtype=torch.int32))


def test_composite_forward_backward_and_loss():
    from humanrf_amd import ops
    from humanrf_amd.volume_rendering import _CompositeFn
    R = 200
    ray, lens = _ragged_rays(R, 3, max_len=120)
    n = ray.numel()
    g = torch.Generator().manual_seed(2)
    sigma = torch.exp(torch.randn(n, generator=g) * 2 + 3)
    rgb = torch.rand(n, 3, generator=g).half().float()
    t = torch.rand(n, generator=g) + 1.0
    bg = torch.rand(R, 3, generator=g)
    rgba = torch.rand(R, 4, generator=g); rgba[:, 3] = (rgba[:, 3] > 0.5).float()
    s_d = sigma.to(DEV).requires_grad_(); c_d = rgb.to(DEV).requires_grad_()
    rs = ops.ray_offsets(ray.to(DEV), R)
    color, acc = _CompositeFn.apply(s_d, c_d, t.to(DEV), rs, bg.to(DEV), R, 4e-4)
    s_r = sigma.clone().requires_grad_(); c_r = rgb.clone().requires_grad_()
    w = O.render_weight_from_density(t, t + 4e-4, s_r, ray)
    color_r = O.accumulate_along_rays(w, ray, c_r, R) + bg * (1 - O.accumulate_along_rays(w, ray, None, R))
    acc_r = O.accumulate_along_rays(w, ray, None, R)
    assert torch.allclose(color.cpu(), color_r, atol=2e-5, rtol=1e-4)   # fp32 scan order only
    assert torch.allclose(acc.cpu(), acc_r, atol=2e-5, rtol=1e-4)
    # loss + gradient kernel against torch autograd evaluated on the SAME colour / acc values (the BCE term is
    # ill-conditioned at acc -> 0/1, so the two sides must not see differently rounded inputs)
    sums = torch.zeros(3, device=DEV)
    d_color, d_acc = ops.loss_fwd_bwd(color.detach(), acc.detach(), rgba.to(DEV), bg.to(DEV), 0.01, 1e-3, 1.0, sums)
    c_l = color.detach().cpu().requires_grad_(); a_l = acc.detach().cpu().requires_grad_()
    loss_r, _ = O.training_loss(c_l, a_l, rgba, bg)
    loss_r.backward()
    loss_d = sums[0] / (3 * R) + 1e-3 * sums[1] / R
    assert abs(float(loss_d) - float(loss_r.detach())) <= 1e-5 * max(1.0, abs(float(loss_r.detach())))
    assert torch.allclose(d_color.cpu(), c_l.grad, rtol=1e-4, atol=1e-9)
    assert torch.allclose(d_acc.cpu(), a_l.grad, rtol=1e-3, atol=1e-9)
    # composite backward with the same upstream gradients on both sides
    gC = torch.randn(R, 3, generator=g) * 1e-3
    gA = torch.randn(R, 1, generator=g) * 1e-3
    torch.autograd.backward([color, acc], [gC.to(DEV), gA.to(DEV)])
    torch.autograd.backward([color_r, acc_r], [gC, gA])
    for a, b in ((s_d.grad.cpu(), s_r.grad), (c_d.grad.cpu(), c_r.grad)):
        assert float((a - b).norm() / b.norm()) <= 1e-4
        assert torch.allclose(a, b, rtol=2e-2, atol=1e-4 * float(b.abs().max()))


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_fused_encode_density_equals_the_two_calls_bit_for_bit(precision):
    """hrf_encode4d_density_fwd (ABI 8: the render pass of the fused training step) against hrf_encode4d_fwd(save) + hrf_density_mlp_fwd:
    features, per-encoding features, h and sigma bit for bit, on samples of several temporal segments, a ragged last tile, both MLP
    arithmetic types."""
    from humanrf_amd import ops
    m = make_model(DEV, (6, 6), tuple(range(15, 27)), log2_T=15, emb=2, table_scale=0.5, mlp_precision=precision)
    n = 64 * 37 + 5
    pos, fn = _random_queries(m, n, seed=8, frames=list(range(15, 27)))
    fn_s, order = torch.sort(fn.reshape(-1))
    pos = pos[order]
    xyzt = torch.cat([pos + 0.5, m.frame_numbers_to_normalized_local_frame_numbers.cpu()[fn_s.long()][:, None]], dim=1).contiguous().to(DEV)
    seg = m.frame_numbers_to_segment_numbers[fn_s.long().to(DEV)].contiguous()
    m._refresh_half()
    sw1, sw2 = m._sigma_w()
    feats, enc = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, save_enc=True)
    h, sigma = ops.density_mlp_fwd(feats, sw1, sw2, float(m.density_scale))
    f2, e2, h2, s2 = ops.encode4d_density_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, sw1, sw2,
                                              float(m.density_scale))
    torch.cuda.synchronize()
    assert torch.equal(f2, feats) and torch.equal(e2, enc)
    assert torch.equal(h2, h) and torch.equal(s2, sigma)
    assert float(sigma.abs().sum()) > 0 and float(feats.float().abs().sum()) > 0


@pytest.mark.parametrize("with_bg,n_rays", [(True, 3001), (False, 517), (True, 4)])
def test_fused_render_loss_equals_the_three_calls_bit_for_bit(with_bg, n_rays):
    """hrf_render_loss_fused (ABI 8: what the fused training step launches) against hrf_composite_fwd + hrf_loss_fwd_bwd +
    hrf_composite_bwd on ragged rays (empty rays, rays longer than a wavefront, a ray count that is not a multiple of the four rays
    of a workgroup): colour, opacity, d_sigma, d_rgb and the touched-group marks bit for bit; the loss sums to fp32 summation
    order; the device-side GradScaler's scale is applied the same way."""
    from humanrf_amd import ops
    R = n_rays
    ray, lens = _ragged_rays(R, 9, max_len=150) if R > 8 else (torch.tensor([0, 0, 0, 2, 2, 3], dtype=torch.int64), None)
    n = ray.numel()
    g = torch.Generator().manual_seed(21)
    sigma = torch.exp(torch.randn(n, generator=g) * 2 + 3).to(DEV)
    rgb = torch.rand(n, 3, generator=g).half().to(DEV)
    t = (torch.rand(n, generator=g) + 1.0).to(DEV)
    bg = torch.rand(R, 3, generator=g).to(DEV) if with_bg else None
    rgba = torch.rand(R, 4, generator=g); rgba[:, 3] = (rgba[:, 3] > 0.5).float(); rgba = rgba.to(DEV)
    frames = torch.randint(0, 12, (R,), generator=g, dtype=torch.int32).to(DEV)
    f2s = (torch.arange(12, dtype=torch.int32) // 4).to(DEV)
    scaler = ops.grad_scaler(DEV, init_scale=1024.0)
    rs = ops.ray_offsets(ray.to(DEV), R)
    for rep in range(2):
        sums_a, sums_b = torch.zeros(3, device=DEV), torch.zeros(3, device=DEV)
        touched_a, touched_b = torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV)
        color, acc = ops.composite_fwd(sigma, rgb, t, rs, bg, R)
        d_color, d_acc = ops.loss_fwd_bwd(color, acc, rgba, bg, 0.01, 1e-3, 128.0, sums_a, frames, f2s, touched_a, scaler=scaler,
                                          norm_rays=R + 5)
        d_sigma, d_rgb = ops.composite_bwd(sigma, rgb, t, rs, bg, d_color, d_acc, R)
        f_sigma, f_rgb, f_color, f_acc = ops.render_loss_fused(sigma, rgb, t, rs, bg, rgba, R, 0.01, 1e-3, 128.0, sums_b, frames,
                                                               f2s, touched_b, scaler=scaler, norm_rays=R + 5, want_color=True)
        torch.cuda.synchronize()
        assert torch.equal(f_color, color) and torch.equal(f_acc, acc)
        assert torch.equal(f_sigma, d_sigma) and torch.equal(f_rgb, d_rgb)
        assert torch.equal(touched_a, touched_b) and int(touched_a.sum()) >= 1
        assert torch.allclose(sums_a, sums_b, rtol=1e-5, atol=1e-6) and float(sums_a[2]) > 0
    assert float(d_sigma.abs().sum()) > 0


@pytest.mark.parametrize("offset", [0, 1])  # 0: 16-byte aligned buffers (vector kernel + scalar tail), 1: unaligned views
def test_adam_matches_torch(offset):
    from humanrf_amd import ops
    n = 10_007
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    p = torch.zeros(n + 1, device=DEV)[offset:offset + n]; p.copy_(p0)
    m = torch.zeros(n + 1, device=DEV)[offset:offset + n]; v = torch.zeros(n + 1, device=DEV)[offset:offset + n]
    p16 = torch.empty(n + 1, dtype=torch.float16, device=DEV)[offset:offset + n]
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * 1e-3
        ref.grad = gr.clone(); opt.step()
        gd = torch.zeros(n + 1, device=DEV)[offset:offset + n]; gd.copy_(gr * 128.0)
        ops.adam_step(p, gd, m, v, p16, 1e-2, 0.9, 0.99, 1e-15, step, 128.0, flags)
        assert float(gd.abs().max()) == 0.0  # gradient buffer is zeroed for the next step
    assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(p16.cpu(), p.cpu().half())
    flags.fill_(1)
    before = p.clone()
    ops.adam_step(p, torch.ones(n + 1, device=DEV)[offset:offset + n], m, v, p16, 1e-2, 0.9, 0.99, 1e-15, 6, 128.0, flags)
    assert torch.equal(p, before)  # found_inf -> step skipped


def test_device_grad_scaler_follows_torch_amp():
    """hrf_adam_multi with a device-resident scaler = GradScaler.unscale_ + step (skipped on found_inf) + update()
    (trainer.py:250-252). The scale sequence is compared with torch's own update op, the parameters with torch.optim.Adam
    stepping on the unscaled gradients of the clean steps only."""
    from humanrf_amd import ops
    n, G = 4099, 1
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    p = p0.to(DEV); gd = torch.zeros(n, device=DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    desc = ops.adam_descriptors([(p, gd, m, v, p16, 0)], DEV)
    ws = ops.adam_workspace(DEV)
    state = torch.zeros(4 + 2 * G, dtype=torch.int32, device=DEV)
    scaler = ops.grad_scaler(DEV, init_scale=65536.0, growth_interval=3)
    t_scale = torch.full((1,), 65536.0, device=DEV); t_track = torch.zeros(1, dtype=torch.int32, device=DEV)
    pattern = [0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0]        # found_inf per step
    for inf in pattern:
        scale_now = float(t_scale)
        assert ops.grad_scaler_state(scaler)["scale"] == scale_now
        gr = torch.randn(n, generator=g) * 1e-3
        gd.copy_(gr * (128.0 * scale_now))                 # what the scaled backward leaves in the gradient buffer
        state[0] = inf
        ops.adam_multi(desc, 1, G, n, 1e-2, 0.9, 0.99, 1e-15, 128.0, state, ws, scaler=scaler)
        if not inf:
            ref.grad = gr.clone(); opt.step()
        torch._amp_update_scale_(t_scale, t_track, torch.full((1,), float(inf), device=DEV), 2.0, 0.5, 3)
        assert float(gd.abs().max()) == 0.0
    st = ops.grad_scaler_state(scaler)
    assert st["scale"] == float(t_scale) and st["growth_tracker"] == int(t_track)
    assert st["scale"] == 32768.0 and st["growth_tracker"] == 2     # two backoffs net, two clean steps since the last growth
    assert state.cpu().tolist()[:2] == [0, sum(pattern)] and int(state[4]) == len(pattern) - sum(pattern)
    assert torch.allclose(p.cpu(), ref.detach(), rtol=2e-5, atol=1e-6)
    assert torch.equal(p16, p.bfloat16())


# ------------------------------------------------------------------------------------------ end to end
def test_prune_and_render_end_to_end():
    """prune_samples + render through the reference-shaped API against the oracle on a synthetic scene."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.volume_rendering import prune_samples, render
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=600, max_buffer_size=8, max_num_frames_per_batch=3, seed=1)
    m = make_model(DEV, (12,), tuple(scene.frame_numbers), log2_T=15, table_scale=0.3)
    om = oracle_model_from(m)
    torch.manual_seed(11)
    ib = next(iter(loader))
    assert ib.num_rays > 50 and ib.num_samples > 5000
    o, d, fr, cm = ib.ray_origins.cpu(), ib.ray_directions.cpu(), ib.frame_numbers.cpu(), ib.camera_numbers.cpu()
    t0, ri0 = ib.sample_distances.cpu().clone(), ib.ray_indices.cpu().clone()
    torch.manual_seed(21)
    prune_samples(ib, m, True)
    torch.manual_seed(21)
    jitter = torch.rand_like(t0.reshape(-1).to(DEV)).cpu()   # the same stream prune_samples consumed
    t_j, _, vis_ref, sigma_ref = O.prune_samples(om, o, d, fr, t0, ri0, jitter)
    # samples kept by the device path, identified by (ray, t)
    kept_ref = set(zip(ri0[vis_ref].tolist(), t_j.reshape(-1)[vis_ref].tolist()))
    kept_dev = set(zip(ib.ray_indices.cpu().tolist(), ib.sample_distances.reshape(-1).cpu().tolist()))
    # density comes from an fp16 MLP: samples whose alpha or transmittance sits at the threshold may flip
    sym = kept_ref ^ kept_dev
    assert len(sym) <= max(3, 0.005 * len(kept_ref)), (len(sym), len(kept_ref))
    assert ib.num_samples > 100
    bg = torch.rand(ib.num_rays, 3)
    out = render(ib, m, bg.to(DEV), True)
    color_ref, acc_ref = O.render(om, o, d, fr, cm, ib.sample_distances.cpu(), ib.ray_indices.cpu(), bg, True)
    assert float((out.color.detach().cpu() - color_ref).abs().max()) <= 2e-3   # stated tolerance on rendered colour
    assert float((out.weights_sum.detach().cpu() - acc_ref).abs().max()) <= 2e-3


def test_fused_prune_march_equals_unfused_sequence():
    """The march kernel (early termination) must select exactly the samples the unfused
    encode -> sigma_net -> visibility sequence selects, for training (jitter) and inference."""
    import humanrf_amd.volume_rendering as vr
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=900, max_buffer_size=8, max_num_frames_per_batch=3, seed=2)
    m = make_model(DEV, (6, 6), tuple(scene.frame_numbers), log2_T=15, table_scale=0.4)
    torch.manual_seed(5)
    base = next(iter(loader))
    import copy
    for training in (True, False):
        outs = []
        for fused in (True, False):
            vr.FUSED_PRUNE = fused
            ib = copy.copy(base)
            ib.sample_distances = base.sample_distances.clone(); ib.ray_indices = base.ray_indices.clone()
            torch.manual_seed(77)
            vr.prune_samples(ib, m, training)
            outs.append((ib.sample_distances.clone(), ib.ray_indices.clone(), getattr(ib, "_num_evaluated", None)))
        vr.FUSED_PRUNE = True
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert 100 < outs[0][0].numel() < base.num_samples
        evaluated = int(outs[0][2].sum())
        assert outs[0][0].numel() <= evaluated <= base.num_samples


@pytest.mark.parametrize("pipelined", [False, True])
def test_step_collector_matches_reference_shaped_sampler(pipelined):
    """The device-resident collector (one sync per iteration, direct packing into step buffers) must hand the
    training step the same rays the reference-shaped sampler produces, and a consistent, sorted sample set --
    also when the sampler stages of a step were prefetched on the second stream and are consumed as prefixes
    (second and third batch of the pipelined collector)."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.fast_path import StepCollector
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=700, max_buffer_size=8, max_num_frames_per_batch=3, seed=4)
    m = make_model(DEV, (12,), tuple(scene.frame_numbers), log2_T=15, table_scale=0.3)
    col = StepCollector(m, loader, samples_max=20_000, rays_initial=700, cap_rays=1 << 14, cap_pre=1 << 12,
                        pipelined=pipelined)  # cap_pre forces a regrow
    torch.manual_seed(9)
    prefetched = 0
    for it in range(4):
        prefetched += int(col.sets[col.cur ^ 1].n_drawn > 0) if pipelined else 0
        pre_before, enc_before = (int(v) for v in col.totals.cpu())
        ib, drawn, _ = col.collect()
        R, N = ib.num_rays, ib.num_samples
        pre_now, enc_now = (int(v) for v in col.totals.cpu())
        assert R > 50 and N >= 0.9 * 20_000 and N <= 1.1 * 20_000 + 1 and drawn >= 700
        assert pre_now - pre_before >= enc_now - enc_before >= N      # handed to the march >= encoded by it >= visible
        ray = ib.ray_indices.cpu()
        assert int(ray.min()) >= 0 and int(ray.max()) < R and bool((ray[1:] >= ray[:-1]).all())
        # the ray offsets the collector hands to the training step (round 6: instead of k_ray_offsets) are the offsets of the packed batch
        if getattr(ib, "_ray_start", None) is not None:
            from humanrf_amd import ops
            assert ib._ray_start.shape[0] == R + 1 and int(ib._ray_start[R]) == N
            assert torch.equal(ib._ray_start, ops.ray_offsets(ib.ray_indices, R))
        t = ib.sample_distances.reshape(-1).cpu()
        same = ray[1:] == ray[:-1]
        assert bool((t[1:][same] > t[:-1][same]).all())                      # distances increase along each ray
        mm = ib.minmaxes.cpu()
        assert bool((t >= mm[ray, 0] - 1e-6).all()) and bool((t <= mm[ray, 1] + 4e-4 + 1e-6).all())
        # every collected ray, pushed through the reference-shaped sampler, reproduces its per-ray data bit-exactly
        out = loader.sample(col.ridx[:R].clone())
        assert bool(out[6].all())
        for a, b in ((out[0], ib.ray_origins), (out[1], ib.ray_directions), (out[2], ib.rgba), (out[5], ib.minmaxes)):
            assert torch.equal(a, b)
        assert torch.equal(out[3], ib.frame_numbers.reshape(-1)) and torch.equal(out[4], ib.camera_numbers.reshape(-1))
        assert out[7].numel() >= N                                           # pruning only removes samples
        # the survivors of every ray are a subset of its sampler samples shifted by a jitter in [0, step)
        t_ref, r_ref = out[7].cpu().double(), out[8].cpu().long()
        key_ref = r_ref.double() * 16.0 + t_ref                              # distances are < 16 scene units
        pos = torch.searchsorted(key_ref, ray.double() * 16.0 + t.double() + 1e-9, right=True) - 1
        assert bool((pos >= 0).all()) and bool((r_ref[pos] == ray).all())
        shift = t.double() - t_ref[pos]
        assert bool(((shift >= -1e-6) & (shift < 4e-4 + 1e-6)).all())
    if pipelined:
        assert prefetched >= 2, "the prefetched path was not exercised"


def test_batch_plan_replays_the_trainer_loop():
    """hrf_batch_plan == the Python statements of trainer.py:138-163 applied to prefix sums (random masks / counts,
    chunks that finish, chunks that run out of marched rays, continuation from a non-zero loop state)."""
    import numpy as np
    from humanrf_amd import _lib
    from humanrf_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    rng = np.random.RandomState(0)
    for case in range(40):
        n = int(rng.randint(3000, 60000))
        mask = rng.rand(n) < rng.uniform(0.05, 0.6)
        per_ray = rng.randint(0, int(rng.choice([3, 20, 120])) + 1, size=int(mask.sum()))
        slot = np.concatenate([[0], np.cumsum(mask)]).astype(np.int32)
        out_off = np.concatenate([[0], np.cumsum(per_ray)]).astype(np.int32)
        rays_initial, samples_max = int(rng.choice([256, 1024, 8192])), int(rng.choice([5_000, 40_000, 640_000]))
        rays_initial = min(rays_initial, n // 2)
        spec_end = int(rng.randint(rays_initial, n + 1))
        # reference loop over the drawn rays [0, spec_end)
        used, r0, tr, ts, its, done, err = 0, rays_initial, 0, 0, 0, 0, 0
        while used + r0 <= spec_end:
            used += r0; tr += r0; ts = int(out_off[slot[used]]); its += 1
            if ts < 0.9 * samples_max:
                avg = ts / tr
                if not avg > 0:
                    err = 1
                    break
                r0 = int((samples_max - ts) / avg)
            else:
                done = 1
                break
        plan = torch.zeros(16, dtype=torch.int64, device=DEV)
        extra = torch.tensor([12345], dtype=torch.int32, device=DEV)
        slot_d, off_d = torch.from_numpy(slot).to(DEV), torch.from_numpy(out_off).to(DEV)   # (kept alive across the launch)
        check(L.hrf_batch_plan(ptr(slot_d), ptr(off_d), 0, 0, spec_end, rays_initial, 0, 0, samples_max, ptr(extra), ptr(plan),
                               stream_ptr()))
        got = plan.cpu().tolist()
        assert got[:10] == [done, its, used, r0, int(slot[used]), ts, err, tr, 12345, int(slot[spec_end])], (case, got)
        rays_c = int(slot[used])   # ray-aligned cut points of the chunk: sample offsets at its quarter points
        assert got[10:] == [int(out_off[rays_c * k // 4]) for k in (1, 2, 3)] + [rays_c, 0, 0], (case, got)
        if not done and not err and its > 0 and used + r0 <= n:   # continue from this state with a second chunk
            base = int(slot[used])
            rel = (out_off[base:] - out_off[base]).astype(np.int32)
            u2, r2, tr2, ts2, its2, done2 = used, r0, tr, ts, 0, 0
            while u2 + r2 <= n:
                u2 += r2; tr2 += r2; ts2 = ts + int(rel[slot[u2] - base]); its2 += 1
                if ts2 < 0.9 * samples_max:
                    r2 = int((samples_max - ts2) / (ts2 / tr2))
                else:
                    done2 = 1
                    break
            rel_d = torch.from_numpy(rel).to(DEV)
            check(L.hrf_batch_plan(ptr(slot_d), ptr(rel_d), base, used, n, r0, tr, ts, samples_max, None, ptr(plan), stream_ptr()))
            got = plan.cpu().tolist()
            assert got[:8] == [done2, its2, u2, r2, int(slot[u2]), ts2 - ts, 0, tr2], (case, "continuation", got)


def test_segment_schedule_is_a_permutation_and_does_not_change_results():
    """hrf_ray_segment_order: ray ids sorted by temporal segment (a schedule for the march, one eighth per XCD);
    the march's outputs must be bit-identical with and without it, also with a device-side ray count below the
    host's upper bound."""
    from humanrf_amd import ops
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=1500, max_buffer_size=8, max_num_frames_per_batch=4, seed=3)
    m = make_model(DEV, (3, 3, 3, 3), tuple(scene.frame_numbers), log2_T=15, table_scale=0.4)
    ib = next(iter(loader))
    R = ib.num_rays
    frames = ib.frame_numbers.reshape(-1).contiguous()
    seg = m.frame_numbers_to_segment_numbers[frames.long()]
    for live in (R, R - 37):
        n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
        for by_frame, key in ((True, frames), (False, seg)):
            order = ops.ray_segment_order(frames, m, n_dev, by_frame=by_frame)[:live].long()
            assert torch.equal(torch.sort(order).values, torch.arange(live, device=DEV))
            s = key[order]
            assert bool((s[1:] >= s[:-1]).all())
    t = ib.sample_distances.reshape(-1).contiguous()
    ray_start = ops.ray_offsets(ib.ray_indices.contiguous(), R)
    jit = torch.rand_like(t)
    outs = []
    for aff in (True, False):
        ts, sg, cnt, ev = ops.prune_march(ib.ray_origins.contiguous(), ib.ray_directions.contiguous(), frames, ray_start, t,
                                          jit, m, want_sigma=True, want_evaluated=True, segment_affinity=aff)
        off = torch.zeros(R + 1, dtype=torch.int32, device=DEV)
        torch.cumsum(cnt, 0, out=off[1:])
        nt, nr = ops.pack_runs(ray_start, cnt, off, ts, int(off[-1]))
        outs.append((cnt.clone(), ev.clone(), nt, nr))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert 100 < int(outs[0][0].sum()) < t.numel()


def test_empty_and_ragged_inputs_through_the_whole_path():
    """Zero samples, rays without samples, a single sample, and runs that are not multiples of the wavefront: every
    operator must accept them (the reference's ops run on whatever the sampler hands over, including nothing)."""
    from humanrf_amd import ops
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.volume_rendering import prune_samples, render
    m = make_model(DEV, (6, 6), tuple(range(15, 27)), log2_T=15, emb=2, table_scale=0.3)
    om = oracle_model_from(m)
    R = 7
    g = torch.Generator().manual_seed(3)
    o = (torch.rand(R, 3, generator=g) - 0.5) * 0.2
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1)
    fr = torch.randint(15, 27, (R, 1), generator=g, dtype=torch.int32)
    cm = torch.randint(0, 6, (R, 1), generator=g, dtype=torch.int32)
    for counts in ([0, 0, 0, 0, 0, 0, 0], [0, 1, 0, 65, 0, 130, 3], [1, 0, 0, 0, 0, 0, 0]):
        counts_t = torch.tensor(counts)
        ray = torch.repeat_interleave(torch.arange(R), counts_t)
        n = int(counts_t.sum())
        t = (torch.cat([torch.arange(c, dtype=torch.float32) for c in counts]) * 4e-4 + 0.01) if n else torch.zeros(0)

        def batch():
            return InputBatch(ray_origins=o.to(DEV), ray_directions=d.to(DEV), frame_numbers=fr.to(DEV), camera_numbers=cm.to(DEV),
                              rgba=torch.rand(R, 4, device=DEV), minmaxes=torch.zeros(R, 2, device=DEV),
                              sample_distances=t.clone().view(-1, 1).to(DEV), ray_indices=ray.to(DEV),
                              unique_frame_numbers=fr.unique().view(-1, 1).to(DEV), width=4, height=4)
        ib = batch()
        bg = torch.rand(R, 3, device=DEV)
        out = render(ib, m, bg, True)
        assert out.color.shape == (R, 3) and out.weights_sum.shape == (R, 1)
        color_ref, acc_ref = O.render(om, o, d, fr, cm, t.view(-1, 1), ray, bg.cpu(), True)
        assert float((out.color.detach().cpu() - color_ref).abs().max()) <= 2e-3
        assert float((out.weights_sum.detach().cpu() - acc_ref).abs().max()) <= 2e-3
        empty = counts_t == 0
        assert torch.allclose(out.color.detach().cpu()[empty], bg.cpu()[empty])      # rays without samples: background
        if n:
            out.color.sum().backward()                                                # backward through ragged runs
            assert torch.isfinite(m.table_params.grad).all()
            m.zero_grad()
        ib2 = batch()
        prune_samples(ib2, m, False)                                                  # fused march on the same runs
        assert ib2.num_samples <= n and ib2.ray_indices.numel() == ib2.sample_distances.numel()
        if ib2.num_samples:
            r2 = ib2.ray_indices.cpu()
            assert bool((r2[1:] >= r2[:-1]).all()) and bool((counts_t[r2] > 0).all())
    # a batch in which nothing is visible (density ~ 0 everywhere): both prune paths must return an empty sample set
    import humanrf_amd.volume_rendering as vr
    scale = m.density_scale
    m.density_scale = 1e-9
    for fused in (True, False):
        vr.FUSED_PRUNE = fused
        ib3 = batch()
        prune_samples(ib3, m, True)
        assert ib3.num_samples == 0 and ib3.ray_indices.numel() == 0
        out3 = render(ib3, m, bg, False)
        assert torch.allclose(out3.color.detach(), bg)
    vr.FUSED_PRUNE = True
    m.density_scale = scale
    # direct operator calls with n == 0
    z4 = torch.zeros(0, 4, device=DEV)
    zs = torch.zeros(0, dtype=torch.int32, device=DEV)
    feats, enc = ops.encode4d_fwd(z4, zs, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, True)
    assert feats.shape == (0, 32) and enc.shape == (0, 4, 32)
    assert ops.scan_exclusive(torch.zeros(0, dtype=torch.int32, device=DEV)).tolist() == [0]
