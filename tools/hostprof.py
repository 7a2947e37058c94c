import os, sys, time, torch, cProfile, pstats
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
for _ in range(300):
    eng.train_iteration()
torch.cuda.synchronize()
tc = ts = tt = 0.0
N = 40
for _ in range(N):
    t0 = time.perf_counter()
    batch, st = eng.collect_batch()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.loss_sums.zero_()
    eng.train_step(batch)
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    tc += t1 - t0; ts += t2 - t1; tt += t3 - t1
print("collect (synced) %.3f ms | train_step enqueue (cpu only) %.3f ms | train_step total %.3f ms" % (1e3 * tc / N, 1e3 * ts / N, 1e3 * tt / N))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    eng.train_iteration()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
