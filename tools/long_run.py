"""Regime curve of a longer training run at configs[1] sizes: visible samples per ray, throughput, PSNR every 2 500 steps."""
import gc, os, sys, time, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
from humanrf_amd.dataset.synthetic import SyntheticDataLoader, SyntheticScene
from humanrf_amd.scene_representation import HumanRF
from humanrf_amd.trainer import TrainEngine
dev = "cuda"
torch.manual_seed(123)
frames = tuple(range(15, 65))
scene = SyntheticScene(frames, num_cameras=160, width=752, height=752, grid_resolution=256, device=dev)
seg = compute_adaptive_segment_sizes(scene.occupancy_grid, list(frames), 1.25)
model = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2, log2_hashmap_size=19, n_levels=16,
                coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1,
                n_hidden_layers_color=2, sh_degree=4, segment_sizes=tuple(seg), camera_embedding_dim=2, device=dev, seed=1337)
loader = SyntheticDataLoader(scene, batch_size=8192, max_buffer_size=200, max_num_frames_per_batch=8, seed=123)
iter(loader)
eng = TrainEngine(model, loader)
gc.collect(); gc.freeze()
total = int(os.environ.get("STEPS", "20000"))
every = 2500
step = 0
while step < total:
    for i in range(every - 100):
        eng.train_iteration(); step += 1
        if step % 16 == 15:
            eng.replace_next()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rays = samples = 0; sums = torch.zeros(3, device=dev)
    for i in range(100):
        st = eng.train_iteration(); step += 1
        rays += st.num_rays; samples += st.num_samples; sums += st.sums
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pf, pc = loader.frame_numbers_cuda.cpu(), loader.camera_numbers_cuda.cpu()
    vframe = int(torch.mode(pf[pf >= 0]).values)
    seen = set(pc[pf == vframe].tolist())
    vcam = next(c for c in range(160) if c not in seen)
    val = bench.validation_psnr(model, scene, vcam, vframe)
    print("step %6d: %.2f Mray/s, %.2f ms/step, %.1f visible samples/ray, train PSNR %.2f dB, novel-view PSNR %.2f dB (cam %d frame %d), skipped %d"
          % (step, rays / dt / 1e6, 1e3 * dt / 100, samples / rays, TrainEngine.psnr_from_sums(sums, rays), val, vcam, vframe, eng.found_inf()), flush=True)
