"""The oracle's restatement of the reference's in-repo CUDA against THE REFERENCE'S OWN KERNEL SOURCE compiled for the host
(oracle/build_ref_cuda.py: the device functions and kernel bodies of ray_sampler.cu:9-194, tensor_composition.cu:9-118 and
occupancy_grid_generation.cu:10-80, cut out of /root/reference at build time and compiled by g++ over oracle/ref_cuda/shim.h ->
oracle/_ref/libhrf_refcuda.so).

What this pins: the control flow, expression structure, operand order, index arithmetic and half roundings of
oracle/sampler_oracle.c, oracle/occgen_oracle.c and the oracle's compose_tensors forward / backward are those of the reference's source
text, under
the floating-point semantics the build fixes (IEEE fp32, no contraction; GLM restated in shim.h; ONE texture definition,
orc_tex_gt0, shared by both sides -- the texture unit itself stays a definition)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref_cuda  # noqa: E402
from oracle import hrf_oracle as O  # noqa: E402

if not build_ref_cuda.available() and not os.path.exists(build_ref_cuda.LIB):
    pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/libhrf_refcuda.so is present", allow_module_level=True)


@pytest.fixture(scope="module")
def ref():
    lib = ctypes.CDLL(build_ref_cuda.build())
    for name in ("ref_compute_minmax", "ref_sample_distances", "ref_compose_forward", "ref_compose_backward"):
        getattr(lib, name).restype = None
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _scene(seed, G=32, W=24, H=20, cams=5):
    """Look-at cameras around the unit cube, one of them in portrait mode, a blob + a thin sheet in the occupancy volumes."""
    rng = np.random.default_rng(seed)
    ikr, org, land = [], [], []
    for c in range(cams):
        portrait = c == 1
        w, h = (H, W) if portrait else (W, H)
        eye = rng.normal(size=3)
        eye = eye / np.linalg.norm(eye) * rng.uniform(1.6, 2.4)
        fwd = -eye / np.linalg.norm(eye)
        up = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        dn = np.cross(fwd, right)
        R = np.stack([right, dn, fwd])                      # world -> camera
        f = rng.uniform(0.9, 1.4) * w
        K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1.0]])
        ikr.append((R.T @ np.linalg.inv(K)).T.astype(np.float32))     # memory is read column-major (data_loader.py:194-207)
        org.append(eye.astype(np.float32))
        land.append(not portrait)
    grids = np.zeros((cams, G, G, G), np.uint8)
    zz, yy, xx = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
    for c in range(cams):
        ctr = rng.uniform(G * 0.35, G * 0.65, size=3)
        grids[c][(xx - ctr[0]) ** 2 + (yy - ctr[1]) ** 2 + (zz - ctr[2]) ** 2 < (G * 0.22) ** 2] = 255
        grids[c][:, int(G * 0.8), :] = 1                    # a one-texel sheet: exercises the refinement steps
    aabb = np.array([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]], np.float32)
    return (np.ascontiguousarray(np.stack(ikr)), np.ascontiguousarray(np.stack(org)), np.array(land, np.uint8), grids, aabb, G, W, H)


def _grid_ptrs(grids):
    arr = (ctypes.c_void_p * grids.shape[0])()
    for i in range(grids.shape[0]):
        arr[i] = grids[i].ctypes.data
    return arr


@pytest.mark.parametrize("occupancy,seed,grid", [(True, 0, 32), (True, 1, 32), (False, 0, 32), (False, 1, 32),
                                                 (True, 2, 50), (True, 3, 112)])   # 50 / 112: the grid sizes of the reference's tests
def test_minmax_and_sample_kernels_equal_the_c_oracle_bit_for_bit(ref, occupancy, seed, grid):
    ikr, org, land, grids, aabb, G, W, H = _scene(seed, G=grid)
    B, P = ikr.shape[0], W * H
    rng = np.random.default_rng(100 + seed)
    idx = np.sort(rng.choice(B * P, size=min(2000, B * P), replace=False)).astype(np.int64)
    R = idx.shape[0]
    gp = _grid_ptrs(grids)
    olib = O._lib()
    outs = []
    for which in ("reference", "oracle"):
        dirs, mm, mask = np.zeros((R, 3), np.float32), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8)
        if which == "reference":
            ref.ref_compute_minmax(ctypes.c_int(int(occupancy)), _p(ikr), _p(org), _p(land), _p(idx), gp, _p(aabb),
                                   ctypes.c_int64(R), ctypes.c_int(B), ctypes.c_int(G), ctypes.c_int(W), ctypes.c_int(H),
                                   _p(dirs), _p(mm), _p(mask))
        else:
            olib.orc_minmax(_p(ikr), _p(org), _p(land), _p(idx), gp, _p(aabb), ctypes.c_int64(R), ctypes.c_int(G),
                            ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(int(occupancy)), _p(dirs), _p(mm), _p(mask))
        outs.append((dirs, mm, mask))
    (d_r, mm_r, mk_r), (d_o, mm_o, mk_o) = outs
    assert np.array_equal(d_r.view(np.uint32), d_o.view(np.uint32))
    assert np.array_equal(mm_r.view(np.uint32), mm_o.view(np.uint32))
    assert np.array_equal(mk_r != 0, mk_o != 0)
    assert 0.05 < (mk_r != 0).mean() < 0.98                     # rays both hit and miss

    # ---- compute_sample_distances_kernel over the compacted rays (host glue of ray_sampler.cu:283-323 done in numpy)
    keep = mk_r != 0
    s_idx, s_mm, s_dirs = idx[keep], np.ascontiguousarray(mm_r[keep]), np.ascontiguousarray(d_r[keep])
    image = (s_idx // P).astype(np.int64)
    s_org = np.ascontiguousarray(org[image])
    Rc = s_idx.shape[0]
    step = np.float32(4e-3)                                     # (coarser than the training step: keeps the test small)
    counts = np.zeros(Rc, np.int32)
    olib.orc_counts(_p(s_mm), ctypes.c_int64(Rc), ctypes.c_float(step), _p(counts))
    assert np.array_equal(counts, ((s_mm[:, 1] - s_mm[:, 0]) / step).astype(np.int32))   # ((max - min) / step).to(int), :283-285
    end = np.cumsum(counts).astype(np.int32)
    ros = np.repeat(np.arange(Rc, dtype=np.int32), counts)
    N = int(end[-1])
    t_all, keep_all = np.zeros(N, np.float32), np.zeros(N, np.uint8)
    ref.ref_sample_distances(ctypes.c_int(int(occupancy)), _p(s_idx), gp, ctypes.c_int(B), ctypes.c_int(G), _p(s_mm), _p(s_org),
                             _p(s_dirs), _p(end), _p(ros), ctypes.c_int64(Rc), ctypes.c_int64(N), ctypes.c_int(P),
                             ctypes.c_float(step), _p(t_all), _p(keep_all))
    rgrids = None
    if occupancy:
        rgrids = (ctypes.c_void_p * Rc)()
        for r in range(Rc):
            rgrids[r] = grids[image[r]].ctypes.data
    olib.orc_samples.restype = ctypes.c_int64
    n = olib.orc_samples(_p(s_org), _p(s_dirs), _p(s_mm), _p(counts), rgrids, ctypes.c_int64(Rc), ctypes.c_int(G),
                         ctypes.c_float(step), None, None)
    t_o, ray_o = np.zeros(n, np.float32), np.zeros(n, np.int32)
    olib.orc_samples(_p(s_org), _p(s_dirs), _p(s_mm), _p(counts), rgrids, ctypes.c_int64(Rc), ctypes.c_int(G),
                     ctypes.c_float(step), _p(t_o), _p(ray_o))
    sel = keep_all != 0                                          # the boolean-mask compaction of ray_sampler.cu:322-323
    assert int(sel.sum()) == n and n > 1000
    assert np.array_equal(t_all[sel].view(np.uint32), t_o.view(np.uint32))
    assert np.array_equal(ros[sel], ray_o)
    if not occupancy:
        assert sel.all()


def test_golden_sampler_inputs_through_the_reference_kernels(ref):
    """The committed golden vector's sampler block (tests/golden/hotpath_seed123.npz: produced by the oracle, reproduced bit for
    bit by the HIP sampler on the GPU) equals what the reference's own kernels give on its inputs."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "hotpath_seed123.npz"))
    ikr, org = np.ascontiguousarray(g["in_inverse_krs"]), np.ascontiguousarray(g["in_camera_origins"])
    grids, aabb, idx = np.ascontiguousarray(g["in_grids"]), np.ascontiguousarray(g["in_aabb"]), np.ascontiguousarray(g["in_idx"])
    W, H, G = int(g["in_W"]), int(g["in_H"]), int(g["in_G"])
    B, R = ikr.shape[0], idx.shape[0]
    land = np.ones(B, np.uint8)
    dirs, mm, mask = np.zeros((R, 3), np.float32), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8)
    ref.ref_compute_minmax(ctypes.c_int(1), _p(ikr), _p(org), _p(land), _p(idx), _grid_ptrs(grids), _p(aabb), ctypes.c_int64(R),
                           ctypes.c_int(B), ctypes.c_int(G), ctypes.c_int(W), ctypes.c_int(H), _p(dirs), _p(mm), _p(mask))
    keep = mask != 0
    assert np.array_equal(keep, g["smp_ray_mask"])
    assert np.array_equal(dirs[keep].view(np.uint32), g["smp_dirs"].view(np.uint32))
    assert np.array_equal(mm[keep].view(np.uint32), g["smp_minmax"].view(np.uint32))


def _halves(rng, shape, scale):
    return (rng.uniform(-1, 1, size=shape) * scale).astype(np.float16)


def test_compose_kernels_equal_the_oracle_restatement(ref):
    """compose_tensors_forward_kernel / _backward_kernel against oracle.hrf_oracle.compose_tensors and the backward restated in
    oracle/ref_stubs.py (what the reference-executed fixtures were generated with): outputs and the four per-encoding gradients
    (half tensors in the reference) bit for bit; the vector gradients, which the kernel sums with atomics in thread order and the
    restatement with index_add, to fp32 summation-order accuracy. Coordinates inside [0, 1] (outside, the reference reads out of
    bounds, tensor_composition.cu:41-42; the build clamps both taps)."""
    from oracle import ref_stubs
    rng = np.random.default_rng(3)
    N, F, V = 700, 32, 64
    feats = [_halves(rng, (N, F), 0.7) for _ in range(4)]
    vec = rng.normal(size=(4, V, F)).astype(np.float32)
    xyzt = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
    xyzt[0] = 0.0; xyzt[1] = 1.0; xyzt[2] = 0.5 / V; xyzt[3] = 1.0 - 0.5 / V     # the clamped end taps
    d_out = _halves(rng, (N, F), 3.0)
    out = np.zeros((N, F), np.uint16)
    u16 = [f.view(np.uint16) for f in feats]
    ref.ref_compose_forward(_p(u16[0]), _p(u16[1]), _p(u16[2]), _p(u16[3]), _p(vec), _p(xyzt), ctypes.c_int64(N), ctypes.c_int(F),
                            ctypes.c_int(V), _p(out))
    tf = [torch.from_numpy(f.astype(np.float32)) for f in feats]
    want = O.compose_tensors(tf[0], tf[1], tf[2], tf[3], torch.from_numpy(vec), torch.from_numpy(xyzt))
    assert np.array_equal(out.view(np.float16), want.numpy().astype(np.float16))
    d = [np.zeros((N, F), np.uint16) for _ in range(4)]
    d_vec = np.zeros_like(vec)
    ref.ref_compose_backward(_p(u16[0]), _p(u16[1]), _p(u16[2]), _p(u16[3]), _p(vec), _p(xyzt), _p(d_out.view(np.uint16)),
                             ctypes.c_int64(N), ctypes.c_int(F), ctypes.c_int(V), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d_vec))
    assert ref_stubs.HALF_GRAD_OUTPUTS
    w = ref_stubs.compose_tensors_backward(tf[0], tf[1], tf[2], tf[3], torch.from_numpy(vec), torch.from_numpy(xyzt),
                                           torch.from_numpy(d_out.astype(np.float32)))
    for k in range(4):
        assert np.array_equal(d[k].view(np.float16), w[k].numpy().astype(np.float16)), k
    scale = float(np.abs(w[4].numpy()).max())
    assert float(np.abs(d_vec - w[4].numpy()).max()) <= 2e-6 * scale


def test_grid_carving_kernel_equals_the_c_oracle(ref):
    """generate_from_masks_kernel (occupancy_grid_generation.cu:16-80) compiled from the reference's source against
    oracle/occgen_oracle.c (which the HIP kernel hrf_occgrid_from_masks is held to bit for bit on the GPU): visual-hull carving of
    random and of blob-shaped masks, portrait and landscape cameras, every threshold incl. one beyond the camera count. Cameras sit
    outside the scene volume (projected depth > 0 for every voxel: the conversion of a non-finite pixel coordinate to int is what
    the build fixes separately, occgen_oracle.c's header)."""
    from humanrf_amd.dataset.synthetic import make_cameras
    W, H, G, C = 40, 32, 24, 7
    cams = make_cameras(C, W, H, radius=2.2)
    proj = np.ascontiguousarray(np.transpose(np.stack([c.projection_matrix_world2pixel() for c in cams], 0).astype(np.float32), (0, 2, 1)))
    land = np.array([1, 1, 0, 1, 0, 1, 1], np.uint8)
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:H, 0:W]
    blob = (((xx - W / 2) ** 2 + (yy - H / 2) ** 2) < (0.3 * H) ** 2).astype(np.uint8).reshape(-1) * 255
    for masks in ((rng.rand(C, W * H) < 0.35).astype(np.uint8) * 255, np.ascontiguousarray(np.tile(blob, (C, 1)))):
        for thr in (1, 3, 6, 7, 8):
            want = O.grid_from_masks(masks, proj, land.astype(bool), thr, G, W, H)
            got = np.zeros((G, G, G), np.uint8)
            ref.ref_grid_from_masks.restype = ctypes.c_int
            rc = ref.ref_grid_from_masks(_p(masks), _p(proj), _p(land), ctypes.c_int(thr), ctypes.c_int(C), ctypes.c_int(G),
                                         ctypes.c_int(W), ctypes.c_int(H), _p(got))
            assert rc == 0 and np.array_equal(got, want), thr
