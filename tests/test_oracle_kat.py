"""Analytic known-answer tests for the CPU oracle itself: they pin the oracle independently
of the HIP kernels. The reference has no tests or golden vectors for this path (parity unpinned)."""
import math

import numpy as np
import pytest
import torch

from oracle import hrf_oracle as O

PLS = float(np.exp(np.log(2048 / 32) / 15))


def _camera_setup(G=32, W=16, H=12):
    # one camera at z=-2 looking along +z, identity rotation, focal 40 px
    K = np.array([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]])
    inv = np.linalg.inv(K)
    ikr = inv.T.astype(np.float32)[None]  # "column-major" as data_loader.py:194-207 passes it
    org = np.array([[0.0, 0.0, -2.0]], np.float32)
    aabb = np.array([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]], np.float32)
    return ikr, org, aabb, W, H


def test_texture_predicate_quantisation():
    G = 8
    g = np.zeros((G, G, G), np.uint8)
    g[3, 3, 3] = 255
    c = (3 + 0.5) / G
    assert O.tex_gt0(g, c, c, c)                       # texel centre
    assert O.tex_gt0(g, c + 0.9 / G, c, c)             # neighbour cell still blends in texel 3
    assert not O.tex_gt0(g, c + 1.0 / G, c, c)         # exactly the next texel centre: weight 0
    # within 1/512 texel of the next centre the 8-bit weight rounds to 0 (9-bit fixed point, A.5)
    assert not O.tex_gt0(g, c + (1.0 - 1.0 / 1024) / G, c, c)
    assert O.tex_gt0(g, c + (1.0 - 3.0 / 512) / G, c, c)
    # clamp addressing: a fully occupied border texel answers outside queries
    g2 = np.zeros((G, G, G), np.uint8)
    g2[0, 0, 0] = 255
    assert O.tex_gt0(g2, -0.3, -0.2, -0.1)


def test_sampler_full_and_empty_grid():
    G = 32
    ikr, org, aabb, W, H = _camera_setup(G)
    idx = np.arange(W * H, dtype=np.int64)
    rgba = np.full((W * H, 4), 255, np.uint8)
    common = dict(rgba_u8=rgba, light_mask=np.zeros(W * H, bool), frame_numbers=np.array([7], np.int32),
                  camera_numbers=np.array([3], np.int32), landscape_modes=np.array([True]), all_ray_indices=idx,
                  inverse_krs=ikr, camera_origins=org, aabb=aabb, grid_resolution=G, image_width=W, image_height=H,
                  raymarching_step_size=4e-4, filter_light_bloom=False, get_samples=True)
    full = np.full((G, G, G), 255, np.uint8)
    out_occ = O.sampler_get_data(grids=[full], occupancy=True, **common)
    out_aabb = O.sampler_get_data(grids=None, occupancy=False, **common)
    # a fully occupied grid reproduces the slab test to within one march step (0.5/G)
    mm_o, mm_a = out_occ[5], out_aabb[5]
    assert out_occ[6].all() and out_aabb[6].all()
    assert np.all(np.abs(mm_o - mm_a) <= 0.5 / G + 1e-6)
    # counts = int((tmax - tmin)/step); samples are tmin + k*step, sorted by ray
    t, ray = out_aabb[7], out_aabb[8]
    cnt = ((mm_a[:, 1] - mm_a[:, 0]) / np.float32(4e-4)).astype(np.int32)
    assert t.shape[0] == cnt.sum() and np.all(np.diff(ray) >= 0)
    first = np.searchsorted(ray, np.arange(W * H))
    assert np.array_equal(t[first], mm_a[:, 0])
    k = 5
    assert t[first[3] + k] == np.float32(mm_a[3, 0] + np.float32(k) * np.float32(4e-4))
    # central ray: straight through the cube, length 1 -> ~2500 samples
    centre = (H // 2) * W + W // 2
    assert abs(cnt[centre] - 2500) <= 3
    assert np.allclose(out_aabb[2], 1.0) and (out_aabb[3] == 7).all() and (out_aabb[4] == 3).all()
    # empty grid: every ray masked out, nothing sampled
    empty = np.zeros((G, G, G), np.uint8)
    out_e = O.sampler_get_data(grids=[empty], occupancy=True, **common)
    assert not out_e[6].any() and out_e[0].shape[0] == 0 and out_e[7].shape[0] == 0


def test_sampler_occupancy_slab():
    # occupied slab z in [12,20) of a 32^3 grid: tmin/tmax of the central ray bracket it
    G = 32
    ikr, org, aabb, W, H = _camera_setup(G)
    g = np.zeros((G, G, G), np.uint8)
    g[12:20] = 255
    centre = np.array([(H // 2) * W + W // 2], np.int64)
    out = O.sampler_get_data(rgba_u8=np.zeros((W * H, 4), np.uint8), light_mask=None,
                             frame_numbers=np.array([0], np.int32), camera_numbers=np.array([0], np.int32),
                             grids=[g], landscape_modes=np.array([True]), all_ray_indices=centre, inverse_krs=ikr,
                             camera_origins=org, aabb=aabb, grid_resolution=G, image_width=W, image_height=H,
                             raymarching_step_size=4e-4, filter_light_bloom=False, occupancy=True, get_samples=True)
    tmin, tmax = out[5][0]
    # texel z covers [z/G, (z+1)/G]; trilinear support extends half a texel either side
    z_lo = 12.0 / G - 0.5 - 0.5 / G + 2.0   # distance from the origin at z=-2
    z_hi = 20.0 / G - 0.5 + 0.5 / G + 2.0
    assert abs(tmin - z_lo) < 0.5 / G and abs(tmax - z_hi) < 0.5 / G + 1e-3
    z = out[7] - 2.0 + 0.5
    assert z.min() >= 12.0 / G - 0.5 / G - 1e-4 and z.max() <= 20.0 / G + 0.5 / G + 1e-4


def test_hashgrid_partition_of_unity_and_dense_trilinear():
    levels = O.hashgrid_levels(16, 15, 32, PLS)
    n = levels[-1].offset + levels[-1].size
    x = torch.rand(257, 3)
    const = torch.full((n, 2), 0.25)
    out = O.hashgrid_encode(x, const, levels)
    assert torch.allclose(out, torch.full_like(out, 0.25), atol=1e-3)  # weights sum to one
    # dense level 0 == trilinear interpolation of the reshaped table (x*scale + 0.5 convention)
    lv = levels[0]
    assert not lv.hashed and lv.res == 32
    table = torch.zeros(n, 2)
    vol = torch.randn(32, 32, 32).half().float()
    table[:32 ** 3, 0] = vol.reshape(-1)  # index = x + y*res + z*res^2
    feats = O.hashgrid_encode(x * 0.9, table, levels)[:, 0]
    pos = x * 0.9 * lv.scale + 0.5
    i0 = pos.floor().long()
    w = pos - pos.floor()
    ref = torch.zeros(x.shape[0])
    for c in range(8):
        dx, dy, dz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        wt = (w[:, 0] if dx else 1 - w[:, 0]) * (w[:, 1] if dy else 1 - w[:, 1]) * (w[:, 2] if dz else 1 - w[:, 2])
        ref += wt * vol[i0[:, 2] + dz, i0[:, 1] + dy, i0[:, 0] + dx]
    assert torch.allclose(feats, ref, atol=2e-3)


def test_hash_function_known_values():
    lv = O.Level(scale=2047.0, res=2048, size=1 << 19, offset=0, hashed=True)
    x = torch.tensor([[100.25 / 2047.0, 200.5 / 2047.0, 300.75 / 2047.0]]) - 0.5 / 2047.0
    idx, w = O.hashgrid_indices(x, lv)
    gx, gy, gz = 100, 200, 300
    expect0 = ((gx * 1) ^ ((gy * 2654435761) & 0xFFFFFFFF) ^ ((gz * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    expect7 = (((gx + 1) * 1) ^ (((gy + 1) * 2654435761) & 0xFFFFFFFF) ^ (((gz + 1) * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    assert int(idx[0, 0]) == expect0 and int(idx[0, 7]) == expect7
    assert abs(float(w.sum()) - 1.0) < 1e-6


def test_vectors_match_grid_sample_border():
    # `coord*Rv - 0.5` + clamped taps == grid_sample(align_corners=False, padding_mode='border') in 1-D
    Rv, F = 64, 4
    vec = torch.randn(4, Rv, F)
    xyzt = torch.rand(100, 4)
    sv = O.vectors_sample(vec, xyzt)
    for i in range(4):
        inp = vec[i].t().reshape(1, F, 1, Rv)
        grid = torch.stack([xyzt[:, i] * 2 - 1, torch.zeros(100)], 1).reshape(1, 1, 100, 2)
        ref = torch.nn.functional.grid_sample(inp, grid, mode="bilinear", padding_mode="border", align_corners=False)
        assert torch.allclose(sv[i], ref[0, :, 0].t(), atol=1e-5)


def test_sh16_axes():
    d = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    sh = O.sh16((d + 1) * 0.5)
    assert torch.allclose(sh[:, 0], torch.full((3,), 0.28209479))
    assert abs(sh[0, 3] + 0.48860251) < 1e-6 and abs(sh[1, 1] + 0.48860251) < 1e-6 and abs(sh[2, 2] - 0.48860251) < 1e-6
    assert abs(sh[2, 6] - (0.94617470 - 0.31539157)) < 1e-6
    assert abs(sh[0, 8] - 0.54627422) < 1e-6 and abs(sh[0, 15] + 0.59004359) < 1e-6


def test_constant_density_weights_and_visibility():
    sigma, dt, n = 50.0, 4e-4, 600
    t = torch.arange(n, dtype=torch.float32) * dt + 1.0
    ray = torch.zeros(n, dtype=torch.long)
    w = O.render_weight_from_density(t, t + dt, torch.full((n,), sigma), ray)
    dts = (t + dt) - t
    T = torch.exp(-torch.cumsum(sigma * dts.double(), 0) + sigma * dts.double())
    assert torch.allclose(w.double(), T * (1 - torch.exp(-sigma * dts.double())), rtol=1e-4, atol=1e-9)
    assert abs(float(w.sum()) - (1 - math.exp(-float((sigma * dts.double()).sum())))) < 1e-4
    alphas = torch.full((n,), 1 - math.exp(-sigma * 4e-4))
    vis = O.render_visibility(alphas, ray, 1e-4, 1e-4)
    k = math.ceil(math.log(1e-4) / math.log(1 - float(alphas[0])))  # first i with T_i < eps
    assert int(vis.sum()) in (k - 1, k, k + 1) and bool(vis[0]) and not bool(vis[-1])
    assert torch.equal(vis, torch.cumsum((~vis).long(), 0) == 0)  # prefix property
    # alpha threshold drops transparent samples but they still attenuate nothing measurable
    a2 = alphas.clone(); a2[::2] = 5e-5
    v2 = O.render_visibility(a2, ray, 1e-4, 1e-4)
    assert not v2[::2].any()
    # two rays are independent
    ray2 = torch.cat([ray, ray + 1])
    v3 = O.render_visibility(torch.cat([alphas, alphas]), ray2, 1e-4, 1e-4)
    assert torch.equal(v3[:n], vis) and torch.equal(v3[n:], vis)


def test_truncated_exp_gradient_clamp():
    x = torch.tensor([-20.0, 0.0, 20.0], requires_grad=True)
    y = O.truncated_exp(x)
    y.sum().backward()
    assert torch.allclose(x.grad, torch.exp(torch.tensor([-15.0, 0.0, 15.0])))


def test_compose_backward_matches_reference_formulas():
    # autograd of the oracle's compose == the explicit formulas of tensor_composition.cu:85-117
    N, Rv = 50, 32
    f = [torch.randn(N, 32).half().float().requires_grad_() for _ in range(4)]
    vec = torch.randn(4, Rv, 32, requires_grad=True)
    xyzt = torch.rand(N, 4)
    dy = torch.randn(N, 32)
    out = f[0] * O.vectors_sample(vec, xyzt)[3] + f[1] * O.vectors_sample(vec, xyzt)[2] + \
        f[2] * O.vectors_sample(vec, xyzt)[0] + f[3] * O.vectors_sample(vec, xyzt)[1]
    out.backward(dy)
    sv = O.vectors_sample(vec.detach(), xyzt)
    assert torch.allclose(f[0].grad, sv[3] * dy, atol=1e-6) and torch.allclose(f[2].grad, sv[0] * dy, atol=1e-6)
    feats = [f[2], f[3], f[1], f[0]]  # {yzt, xzt, xyt, xyz} pair with vectors {0,1,2,3}
    dvec = torch.zeros_like(vec)
    for i in range(4):
        coord = xyzt[:, i] * Rv - 0.5
        fl = coord.floor(); fr = (coord - fl).unsqueeze(1)
        c0 = fl.clamp(0, Rv - 1).long(); c1 = (fl + 1).clamp(0, Rv - 1).long()
        dval = feats[i].detach() * dy
        dvec[i].index_add_(0, c0, dval * (1 - fr))
        dvec[i].index_add_(0, c1, dval * fr)
    assert torch.allclose(vec.grad, dvec, atol=1e-5)


# ----------------------------------------------------------------------------------------------------------------
# Occupancy grids from masks (oracle/occgen_oracle.c): second, vectorised restatement + analytic cases
# ----------------------------------------------------------------------------------------------------------------
def _np_grid_from_masks(masks, proj_t, land, thr, G, width, height):
    """Independent NumPy float32 restatement (same operand order, no FMA in NumPy) of
    occupancy_grid_generation.cu:16-80, vectorised over voxels, sequential over cameras."""
    f = np.float32
    ax = (np.arange(G, dtype=f) / f(G - 1) - f(0.5)).astype(f)
    vz, vy, vx = np.meshgrid(ax, ax, ax, indexing="ij")
    covered = np.zeros((G, G, G), np.int32)
    in_hull = np.zeros((G, G, G), bool)
    done = np.zeros((G, G, G), bool)
    C = proj_t.shape[0]
    for c in range(C):
        m = proj_t[c].reshape(16).astype(f)  # column-major flat
        cw, ch = (width, height) if land[c] else (height, width)
        with np.errstate(all="ignore"):
            px = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * f(1))
            py = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * f(1))
            pz = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * f(1))
            qx, qy = (px / pz).astype(f), (py / pz).astype(f)
        def rz(q):
            q = np.where(np.isnan(q), f(0), q)
            return np.trunc(np.clip(q.astype(np.float64), -2147483648.0, 2147483647.0)).astype(np.int64)
        x, y = rz(qx), rz(qy)
        inside = (x >= 0) & (x < cw) & (y >= 0) & (y < ch) & ~done
        xs, ys = np.clip(x, 0, cw - 1), np.clip(y, 0, ch - 1)
        x1, y1 = np.minimum(xs + 1, cw - 1), np.minimum(ys + 1, ch - 1)
        mk = masks[c]
        empty = (mk[xs + ys * cw] == 0) & (mk[x1 + ys * cw] == 0) & (mk[xs + y1 * cw] == 0) & (mk[x1 + y1 * cw] == 0)
        rest = C - c - 1
        stop_empty = inside & empty & (covered + rest < thr)
        hit = inside & ~empty
        covered = covered + hit.astype(np.int32)
        newly = hit & (covered >= thr)
        in_hull |= newly
        done |= stop_empty | newly
    return np.where(in_hull, 255, 0).astype(np.uint8)


def _toy_cameras(n, W, H):
    from humanrf_amd.dataset.synthetic import make_cameras
    cams = make_cameras(n, W, H, radius=2.2)
    return cams


def _proj_t(cams):
    p = np.stack([c.projection_matrix_world2pixel() for c in cams], 0).astype(np.float32)
    return np.ascontiguousarray(np.transpose(p, (0, 2, 1)))


def test_grid_from_masks_against_vectorised_restatement_and_analytic_cases():
    W, H, G = 40, 32, 24
    cams = _toy_cameras(7, W, H)
    proj = _proj_t(cams)
    land = np.array([True, True, False, True, False, True, True])
    rng = np.random.RandomState(0)
    masks = (rng.rand(7, W * H) < 0.35).astype(np.uint8) * 255
    for thr in (1, 3, 6, 7, 8):
        a = O.grid_from_masks(masks, proj, land, thr, G, W, H)
        b = _np_grid_from_masks(masks, proj, land, thr, G, W, H)
        assert a.shape == (G, G, G) and np.array_equal(a, b), thr
    assert O.grid_from_masks(masks, proj, land, 8, G, W, H).sum() == 0           # more cameras than exist
    assert O.grid_from_masks(np.zeros_like(masks), proj, land, 1, G, W, H).sum() == 0
    full = O.grid_from_masks(np.full_like(masks, 255), proj, np.ones(7, bool), 1, G, W, H)
    # all-foreground masks, threshold 1: occupied exactly where SOME camera sees the voxel inside its image
    ax = np.arange(G, dtype=np.float32) / np.float32(G - 1) - np.float32(0.5)
    c = G // 2  # a voxel near the scene centre is seen by every camera (they look at the origin)
    assert full[c, c, c] == 255
    with pytest.raises(RuntimeError):
        O.grid_from_masks(masks[:, :-1], proj, land, 1, G, W, H)


def test_mask_dilate_equals_grey_dilation():
    from scipy import ndimage
    rng = np.random.RandomState(1)
    m = (rng.rand(3, 21, 17) < 0.05).astype(np.uint8) * rng.randint(1, 256, (3, 21, 17)).astype(np.uint8)
    for k in (1, 2, 3, 5, 6):
        ref = np.stack([ndimage.maximum_filter(x, size=k, mode="constant", cval=0) for x in m])
        assert np.array_equal(O.mask_dilate(m, k), ref), k


def test_sh16_is_the_orthonormal_real_basis_of_scipy():
    """Independent pin of the SH restatement (tcnn SphericalHarmonics degree 4): on the unit sphere the 16 functions are
    orthonormal (Gauss-Legendre x uniform-azimuth quadrature, exact for these polynomial degrees) and, degree by degree,
    span the same space as scipy's spherical harmonics (each function's projection onto scipy's degree-l real basis has
    unit norm, and none leaks into another degree)."""
    from scipy import special
    nt, nphi = 16, 32
    ct, wt = np.polynomial.legendre.leggauss(nt)
    phi = (np.arange(nphi) + 0.5) * (2 * np.pi / nphi)
    CT, PH = np.meshgrid(ct, phi, indexing="ij")
    W = np.repeat(wt[:, None], nphi, 1) * (2 * np.pi / nphi)
    st = np.sqrt(1 - CT ** 2)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3)
    Y = O.sh16(torch.from_numpy((d + 1) * 0.5).float()).double().numpy()          # (P, 16)
    G = (Y * W.reshape(-1, 1)).T @ Y
    assert np.abs(G - np.eye(16)).max() < 2e-6                                      # orthonormal
    theta = np.arccos(CT).reshape(-1)                                               # polar
    ph = PH.reshape(-1)
    start = 0
    for l in range(4):
        basis = []
        for m in range(-l, l + 1):
            if hasattr(special, "sph_harm_y"):
                c = special.sph_harm_y(l, abs(m), theta, ph)
            else:
                c = special.sph_harm(abs(m), l, ph, theta)
            basis.append(c.real * (np.sqrt(2) if m > 0 else 1.0) if m >= 0 else c.imag * np.sqrt(2))
        B = np.stack(basis, 1)                                                       # (P, 2l+1) real, orthonormal
        ours = Y[:, start:start + 2 * l + 1]
        C = (B * W.reshape(-1, 1)).T @ ours                                          # coefficients of ours in scipy's basis
        assert np.abs(C.T @ C - np.eye(2 * l + 1)).max() < 5e-6                      # same (2l+1)-dimensional space, unit norms
        start += 2 * l + 1


def test_mlp_known_answers_and_half_rounding_between_layers():
    """Bias-free MLP restatement (tcnn FullyFusedMLP, A.2): hand-computable cases, including one where the half rounding
    of the hidden activations is visible in the output, ReLU clipping, and the sigmoid output activation."""
    x = torch.tensor([[1.0, 2.0 ** -12]])                       # fp32 accumulation: 1 + 2^-12 before rounding
    w1 = torch.tensor([[1.0, 1.0], [-1.0, 0.0]])                # hidden = relu([1 + 2^-12, -1]) -> half -> [1.0, 0]
    w2 = torch.tensor([[4096.0, 5.0]])
    out = O.mlp(x, [w1, w2], "None")
    assert float(out[0, 0]) == 4096.0                           # (1 + 2^-12) * 4096 = 4097 without the rounding
    # identity chain keeps half-representable inputs exact; negative pre-activations are clipped
    v = torch.tensor([[0.5, -0.25, 3.0]])
    eye = torch.eye(3)
    assert torch.equal(O.mlp(v, [eye, eye, eye], "None"), torch.tensor([[0.5, 0.0, 3.0]]))
    # the last layer has no ReLU; Sigmoid output is rounded to half
    s = O.mlp(v, [eye, -eye], "Sigmoid")
    want = torch.sigmoid(torch.tensor([[-0.5, 0.0, -3.0]])).half().float()
    assert torch.equal(s, want)
    with pytest.raises(ValueError):
        O.mlp(v, [eye], "Tanh")


def test_color_net_input_layout():
    """Composite[SphericalHarmonics(3 -> 16), Identity(rest)] padded with ones to a multiple of 16 (A.3, humanrf.py:135-156):
    layout, padding value and half rounding, without (15 geometry features -> 32) and with a 2-D camera embedding (-> 48)."""
    d = torch.nn.functional.normalize(torch.tensor([[0.3, -0.5, 0.8], [0.0, 0.0, 1.0]]), dim=1)
    geo = torch.tensor([[0.1 * i for i in range(15)], [-(0.05 * i) for i in range(15)]])
    x = O.color_net_input(d, geo, None)
    assert x.shape == (2, 32)
    assert torch.equal(x[:, :16], O.sh16((d + 1.0) * 0.5).half().float())
    assert torch.equal(x[:, 16:31], geo.half().float())
    assert torch.equal(x[:, 31], torch.ones(2))
    emb = torch.tensor([[0.25, -0.75], [1.5, 2.0]])
    y = O.color_net_input(d, geo, emb)
    assert y.shape == (2, 48)
    assert torch.equal(y[:, :31], x[:, :31]) and torch.equal(y[:, 31:33], emb) and torch.equal(y[:, 33:], torch.ones(2, 15))


def test_rendering_functions_against_per_ray_python_loops():
    """Second, loop-level restatement of nerfacc 0.3.1's three functions (A.4) on small ragged inputs: per ray, in fp64,
    T_i = prod_{j<i}(1 - alpha_j) for visibility and exp(-sum_{j<i} sigma_j dt_j) for the weights."""
    g = torch.Generator().manual_seed(11)
    counts = [0, 1, 7, 0, 64, 65, 3, 130]
    ray = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts))
    n = int(sum(counts))
    sigma = torch.exp(torch.randn(n, generator=g) * 2.5 + 4.0)
    t0 = torch.rand(n, generator=g)
    dt = 4e-4 * (0.5 + torch.rand(n, generator=g))
    vals = torch.rand(n, 3, generator=g)
    t1 = t0 + dt
    dt = t1 - t0                     # what the functions see: the fp32 difference of the interval ends
    alphas = 1.0 - torch.exp(-sigma * dt)
    vis = O.render_visibility(alphas, ray, 1e-4, 1e-4)
    w = O.render_weight_from_density(t0, t1, sigma, ray)
    acc = O.accumulate_along_rays(w, ray, vals, len(counts))
    asum = O.accumulate_along_rays(w, ray, None, len(counts))
    i = 0
    for r, c in enumerate(counts):
        T32 = np.float32(1.0)          # visibility: sequential fp32 product, as the build fixes it
        opt = 0.0                      # weights: optical depth in fp64
        col = np.zeros(3)
        tot = 0.0
        for k in range(c):
            a = np.float32(alphas[i])
            assert bool(vis[i]) == bool(T32 >= np.float32(1e-4) and a >= np.float32(1e-4)), (r, k)
            T32 = np.float32(T32 * np.float32(np.float32(1.0) - a))
            wk = math.exp(-opt) * (1.0 - math.exp(-float(sigma[i]) * float(dt[i])))
            assert abs(float(w[i]) - wk) <= 2e-6 + 1e-5 * wk
            opt += float(sigma[i]) * float(dt[i])
            col += wk * vals[i].double().numpy()
            tot += wk
            i += 1
        assert np.allclose(acc[r].double().numpy(), col, atol=1e-5) and abs(float(asum[r, 0]) - tot) <= 1e-5
    assert i == n and 0 < int(vis.sum()) < n


def test_hashgrid_against_scalar_restatement():
    """Sample-by-sample, level-by-level restatement of tcnn's grid encoding (A.1) with Python integers (explicit 32-bit
    wrap) and fractions computed in fp64 from the same fp32 position: indices must agree exactly with the vectorised
    oracle on every level (dense and hashed), features to 1e-6 before the half rounding."""
    levels = O.hashgrid_levels(16, 15, 32, PLS)           # T = 2^15: level 0 dense (32^3 entries), the rest hashed
    n_entries = levels[-1].offset + levels[-1].size
    g = torch.Generator().manual_seed(3)
    table = ((torch.rand(n_entries, 2, generator=g) * 2 - 1) * 0.5).half().float()
    x = torch.rand(23, 3, generator=g)
    x[0] = torch.tensor([0.0, 0.0, 0.0]); x[1] = torch.tensor([1.0, 1.0, 1.0])   # box corners: far corner wraps the dense level
    got = O.hashgrid_encode(x, table, levels)
    P = (1, 2654435761, 805459861)
    kinds = set()
    for li, lv in enumerate(levels):
        idx_v, w_v = O.hashgrid_indices(x, lv)
        kinds.add(bool(lv.hashed))
        for s in range(x.shape[0]):
            pos = [np.float32(float(x[s, d]) * float(np.float32(lv.scale)) + 0.5) for d in range(3)]   # one rounding (fma)
            gi = [int(math.floor(p)) for p in pos]
            fr = [float(np.float32(p - np.float32(math.floor(p)))) for p in pos]
            f = [0.0, 0.0]
            for c in range(8):
                cc = [gi[d] + ((c >> d) & 1) for d in range(3)]
                wgt = 1.0
                for d in range(3):
                    wgt *= fr[d] if (c >> d) & 1 else 1.0 - fr[d]
                if lv.hashed:
                    h = 0
                    for d in range(3):
                        h ^= (cc[d] * P[d]) & 0xFFFFFFFF
                    index = h % lv.size
                else:
                    index = ((cc[0] + cc[1] * lv.res + cc[2] * lv.res * lv.res) & 0xFFFFFFFF) % lv.size
                assert index == int(idx_v[s, c]), (li, s, c)
                assert abs(wgt - float(w_v[s, c])) < 1e-6
                f[0] += wgt * float(table[lv.offset + index, 0]); f[1] += wgt * float(table[lv.offset + index, 1])
            for k in range(2):
                assert abs(f[k] - float(got[s, 2 * li + k])) <= 2e-3 * max(1.0, abs(f[k])) , (li, s, k)   # half output
    assert kinds == {False, True}


def test_half_accumulate_hashgrid_bounds_the_fp32_accumulate_deviation():
    """tcnn's kernel_grid interpolates with __half weights into a __half accumulator (eight roundings); the oracle's
    definition, which the gfx950 kernels reproduce to the bit, keeps fp32 weights and an fp32 sum and rounds once (a stated
    deviation). hashgrid_encode(..., accumulate="fp16" | "fp16_legacy") restates the upstream forms
    [UPSTREAM-KNOWLEDGE]; this pins how far the definitions are apart on the model's 16-level grid, from the initial table
    magnitude (1e-4) to trained ones (0.05 - 2): at most 2^-9 of the largest table value (four half ulps at that magnitude;
    four half subnormal steps for the initial tables) -- the interpolated features are sums of eight products of that size."""
    g = torch.Generator().manual_seed(0)
    levels = O.hashgrid_levels(16, 15, 32, float(np.exp(np.log(2048 / 32) / 15)))
    entries = levels[-1].offset + levels[-1].size
    x = torch.rand(6000, 3, generator=g)
    for scale in (1e-4, 0.05, 0.5, 2.0):
        table = O.round_half((torch.rand(entries, 2, generator=g) * 2 - 1) * scale)
        a = O.hashgrid_encode(x, table, levels)
        bound = max(2.0 ** -9 * float(table.abs().max()), 4 * 2.0 ** -24)
        for mode in ("fp16", "fp16_legacy"):
            b = O.hashgrid_encode(x, table, levels, accumulate=mode)
            assert float((a - b).abs().max()) <= bound, (scale, mode)
            assert float((a != b).float().mean()) > 0.2     # really different roundings, not the same code path
    with pytest.raises(ValueError):
        O.hashgrid_encode(x, table, levels, accumulate="bf16")


def test_half_accumulate_mode_bounds_the_fp32_accumulate_deviation():
    """tcnn's FullyFusedMLP accumulates in __half fragments, the gfx950 kernels in fp32 on the matrix cores (a stated
    deviation). The oracle restates both (mlp(..., accumulate=...)); this pins how far apart they can
    be on the two networks of the model (humanrf.py:123-156) at the magnitudes the encoding produces: two half ulps on the
    sigma_net output, 4e-3 relative on sigma = exp(h0) * scale, one half ulp on RGB."""
    g = torch.Generator().manual_seed(0)

    def xavier(o, i):
        return O.round_half((torch.rand(o, i, generator=g) * 2 - 1) * (6.0 / (i + o)) ** 0.5)
    sw, cw = [xavier(64, 32), xavier(16, 64)], [xavier(64, 48), xavier(64, 64), xavier(16, 64)]
    differing = 0.0
    for scale in (0.1, 0.5, 2.0):
        x = O.round_half((torch.rand(20_000, 32, generator=g) * 2 - 1) * scale)
        a, b = O.mlp(x, sw, "None"), O.mlp(x, sw, "None", accumulate="fp16")
        ulp = 2.0 ** (np.floor(np.log2(float(a.abs().max()))) - 10)
        assert float((a - b).abs().max()) <= 2 * ulp
        rel = ((torch.exp(a[:, 0]) - torch.exp(b[:, 0])).abs() / torch.exp(a[:, 0])).max()
        assert float(rel) <= 4e-3
        differing = max(differing, float((a != b).float().mean()))
        xc = O.round_half(torch.cat([torch.rand(20_000, 33, generator=g) * 2 - 1, torch.ones(20_000, 15)], 1))
        ca, cb = O.mlp(xc, cw, "Sigmoid"), O.mlp(xc, cw, "Sigmoid", accumulate="fp16")
        assert float((ca - cb).abs().max()) <= 2.0 ** -11 + 1e-9
    assert differing > 0.2          # the two modes really are different roundings, not the same code path
    with pytest.raises(ValueError):
        O.mlp(x, sw, "None", precision="bf16", accumulate="fp16")


def test_openmp_restatement_of_the_decomposition4d_forward_equals_the_oracle_bit_for_bit():
    """oracle/encode_oracle.c (the OpenMP gather bench.py's cpu_baseline uses for its pruning pass) against
    oracle.hrf_oracle.decomposition4d, which the reference-executed fixtures pin: hashed and dense levels, coordinates on the box faces,
    values that round to half subnormals and across binades, one thread and several."""
    import torch
    from oracle import hrf_oracle as O
    g = torch.Generator().manual_seed(5)
    for log2_T, n_levels in ((12, 16), (19, 16), (15, 5)):
        levels = O.hashgrid_levels(n_levels, log2_T, 32, float(np.exp(np.log(2048 / 32) / 15)))
        ent = levels[-1].offset + levels[-1].size
        tables = [((torch.rand(ent, 2, generator=g) * 2 - 1) * (3e-5 if k == 0 else 0.4)).half().float() for k in range(4)]
        vectors = torch.randn(4, 2048, 2 * n_levels, generator=g) * 0.3
        xyzt = torch.rand(3001, 4, generator=g)
        xyzt[:40] = torch.tensor([0.0, 1.0, 0.5, 1.0])
        xyzt[40:80, 0] = 1.0
        xyzt[80:120, 3] = 0.0
        with torch.no_grad():
            want = O.decomposition4d(xyzt, tables, vectors, levels)
            for threads in (1, 4):
                got = O.decomposition4d_c(xyzt, tables, vectors, levels, threads)
                assert torch.equal(got, want), (log2_T, n_levels, threads, float((got - want).abs().max()))
        assert float(want.abs().sum()) > 0
    # the switch: grad mode off -> the C path, grad mode on -> torch (autograd)
    O.C_ENCODE_THREADS = 2
    try:
        with torch.no_grad():
            assert torch.equal(O.decomposition4d(xyzt, tables, vectors, levels), want)
        t0 = tables[0].clone().requires_grad_()
        out = O.decomposition4d(xyzt, [t0] + tables[1:], vectors, levels)
        assert out.requires_grad and torch.equal(out.detach(), want)
    finally:
        O.C_ENCODE_THREADS = 0
