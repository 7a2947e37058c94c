"""Workload for PMC profiling of the gather kernels: trains a little, then runs the prune pass a few times."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humanrf_amd import ops
from humanrf_amd.dataset.synthetic import SyntheticDataLoader, SyntheticScene
from humanrf_amd.scene_representation import HumanRF
from humanrf_amd.trainer import TrainEngine
dev = "cuda"
torch.manual_seed(123)
frames = tuple(range(15, 65))
scene = SyntheticScene(frames, num_cameras=160, width=752, height=752, grid_resolution=256, device=dev)
model = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2, log2_hashmap_size=19, n_levels=16,
                coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1,
                n_hidden_layers_color=2, sh_degree=4, segment_sizes=(6, 6, 6, 12, 6, 6, 12), camera_embedding_dim=2, device=dev)
loader = SyntheticDataLoader(scene, batch_size=8192, max_buffer_size=200, max_num_frames_per_batch=8, seed=123)
iter(loader)
eng = TrainEngine(model, loader)
for _ in range(int(os.environ.get("PM_WARM", "300"))):
    eng.train_iteration()
torch.cuda.synchronize()
eng.collector.evaluated.zero_()
n = 0
col = eng.collector
orig_iter = col._iteration
def traced(r0, ray_base, samp_base):
    before = int(col.evaluated.item())
    out = orig_iter(r0, ray_base, samp_base)
    print("march launch: r0 %d rays %d n0 %d n1 %d evaluated %d" % (r0, out[0], out[1], out[2], int(col.evaluated.item()) - before), flush=True)
    return out
col._iteration = traced
for _ in range(4):
    ib, drawn, pre = col.collect()
    n += 1
torch.cuda.synchronize()
print("collects", n, "evaluated samples total", int(eng.collector.evaluated.item()), "visible last", ib.num_samples, "rays last", ib.num_rays, "pre last", pre)
# stand-alone encode on the last batch for comparison
m = model
t = ib.sample_distances.reshape(-1).contiguous(); ray_idx = ib.ray_indices.contiguous()
xyzt, seg = ops.query_prep(ib.ray_origins.contiguous(), ib.ray_directions.contiguous(), ib.frame_numbers.reshape(-1).contiguous(), ray_idx, t, None,
                           m.frame_numbers_to_segment_numbers, m.frame_numbers_to_normalized_local_frame_numbers)
for _ in range(3):
    ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, False)
torch.cuda.synchronize()
print("standalone encode samples", xyzt.shape[0])
