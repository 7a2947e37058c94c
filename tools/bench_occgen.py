"""Timing of the grid-generation step at the dataset's size: 160 cameras, 4x masks (1028 x 752 -> 752^2 crop), G = 256."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humanrf_amd.dataset.synthetic import SyntheticScene
from humanrf_amd.toolbox.generate_occupancy_grids_from_masks import generate_occupancy_grid_from_masks
dev = "cuda"
G = int(os.environ.get("G", "256"))
scene = SyntheticScene(tuple(range(15, 20)), num_cameras=160, width=752, height=752, grid_resolution=G, device=dev)
frame = 17
C, W, H = len(scene.cameras), scene.width, scene.height
masks = torch.stack([scene.render_rgba(c, frame)[:, 3].reshape(H, W) for c in range(C)]).contiguous()
for thr in (C - 8, C // 2):
    generate_occupancy_grid_from_masks(masks, scene.cameras, G, thr); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        grid = generate_occupancy_grid_from_masks(masks, scene.cameras, G, thr)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    ref = scene.occupancy_grid(frame)
    print("G %d, %d cameras, threshold %d: %.2f ms per grid (dilate k=%d + carve), %.1f G voxel-camera tests/s upper bound, occupied %.2f %% (analytic dilated grid %.2f %%), analytic-inside voxels missed %d"
          % (G, C, thr, ms, max(W, H) // 128, G ** 3 * C / ms / 1e6, 100.0 * float((grid == 255).float().mean()), 100.0 * float((ref == 255).float().mean()),
             int(((ref == 255) & (grid == 0)).sum())))
