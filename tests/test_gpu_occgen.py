"""GPU parity of the occupancy-grid generation step against oracle/occgen_oracle.c:
byte work, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import hrf_oracle as O
from tests.util import small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _proj_t(cams):
    p = np.stack([c.projection_matrix_world2pixel() for c in cams], 0).astype(np.float32)
    return np.ascontiguousarray(np.transpose(p, (0, 2, 1)))


@pytest.mark.parametrize("G,thr", [(32, 1), (32, 4), (48, 6), (33, 7)])
def test_generate_from_masks_bit_exact_random_masks_mixed_orientation(G, thr):
    from humanrf_amd.toolbox import occupancy_grid_generation_native as native
    scene = small_scene(DEV)
    cams = scene.cameras
    W, H = scene.width, scene.height
    C = len(cams)
    rng = np.random.RandomState(G + thr)
    masks = (rng.rand(C, W * H) < 0.3).astype(np.uint8) * 255
    land = rng.rand(C) < 0.6
    proj = _proj_t(cams)
    got = native.generate_from_masks(torch.from_numpy(masks).to(DEV), torch.from_numpy(proj).to(DEV),
                                     torch.from_numpy(land).to(DEV), thr, G, W, H)
    ref = O.grid_from_masks(masks, proj, land, thr, G, W, H)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (G, G, G)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert 0 < int((ref == 255).sum()) < G ** 3 or thr > C


def test_visual_hull_of_the_synthetic_person_contains_it():
    """End to end on the synthetic capture: silhouettes -> dilation -> carving; the result must equal the oracle's and
    contain every voxel centre that lies inside the person (the hull of the silhouettes contains the object)."""
    from humanrf_amd.toolbox.generate_occupancy_grids_from_masks import generate_occupancy_grid_from_masks, projection_matrices_for
    scene = small_scene(DEV, W=96, H=80, num_cameras=10)
    frame = scene.frame_numbers[3]
    C, W, H, G = len(scene.cameras), scene.width, scene.height, 40
    masks = torch.stack([scene.render_rgba(c, frame)[:, 3].reshape(H, W) for c in range(C)]).contiguous()
    assert masks.dtype == torch.uint8 and 0 < int((masks > 0).sum()) < masks.numel()
    grid = generate_occupancy_grid_from_masks(masks, scene.cameras, G, camera_coverage_threshold=C - 3, dilate=True)
    k = max(W, H) // 128
    dil = O.mask_dilate(masks.cpu().numpy(), k) if k > 0 else masks.cpu().numpy()
    ref = O.grid_from_masks(dil.reshape(C, -1), projection_matrices_for(scene.cameras, "cpu").numpy(),
                            np.array([c.width > c.height for c in scene.cameras]), C - 3, G, W, H)
    assert np.array_equal(grid.cpu().numpy(), ref)
    # analytic inside test at the voxel centres
    c, r, _ = scene._ellipsoids_normalised(frame)
    lin = torch.arange(G, device=DEV, dtype=torch.float32) / (G - 1) - 0.5
    z, y, x = torch.meshgrid(lin, lin, lin, indexing="ij")
    p = torch.stack([x, y, z], -1).reshape(-1, 1, 3)
    inside = ((((p - c.unsqueeze(0)) / r.unsqueeze(0)) ** 2).sum(-1) < 0.5).any(1).reshape(G, G, G)
    assert int(inside.sum()) > 20
    assert int((inside & (grid == 0)).sum()) == 0
    assert int((grid == 255).sum()) < 0.5 * G ** 3


@pytest.mark.parametrize("k", [1, 2, 5, 6])
def test_mask_dilate_bit_exact(k):
    from humanrf_amd.toolbox import occupancy_grid_generation_native as native
    rng = np.random.RandomState(k)
    m = ((rng.rand(4, 37, 53) < 0.04) * rng.randint(1, 256, (4, 37, 53))).astype(np.uint8)
    got = native.dilate_masks(torch.from_numpy(m).to(DEV), k)
    assert np.array_equal(got.cpu().numpy(), O.mask_dilate(m, k))


def test_generate_from_masks_errors():
    from humanrf_amd.toolbox import occupancy_grid_generation_native as native
    masks = torch.zeros(2, 100, dtype=torch.uint8, device=DEV)
    proj = torch.zeros(2, 4, 4, device=DEV)
    land = torch.ones(2, dtype=torch.bool, device=DEV)
    with pytest.raises(RuntimeError):
        native.generate_from_masks(masks, proj, land, 1, 8, 11, 10)            # width*height mismatch
    with pytest.raises(RuntimeError):
        native.generate_from_masks(masks.cpu(), proj, land, 1, 8, 10, 10)      # wrong device
