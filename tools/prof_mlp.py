"""Workload for the MFMA counter passes: a few hundred training steps (every kernel of the step runs each step)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
n = 0
for _ in range(int(os.environ.get("PM_WARM", "400"))):
    st = eng.train_iteration(); n = st.num_samples
torch.cuda.synchronize()
print("samples in the last step", n)
