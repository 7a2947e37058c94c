"""Regime curve of a longer training run at configs[1] sizes: throughput, visible samples per ray, training PSNR and
novel-view PSNR (held-out validation cameras, humanrf_amd.inference.validate) every EVERY steps.
Environment: STEPS (default 20000), EVERY (2500), REPLACE (pool slots refilled per step by the replacer thread, default 8;
0 = the round-1 cadence of one synchronous replacement every 16 steps), EMB (camera_embedding_dim, default 2), EXTRA (further
bench.py arguments, e.g. EXTRA="--frames 250": the configs[3] shape on one GPU), VIEWS (held-out views per point, default 8)."""
import gc, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from humanrf_amd.inference import validate
from humanrf_amd.trainer import TrainEngine

sys.argv = [sys.argv[0], "--emb", os.environ.get("EMB", "2")] + os.environ.get("EXTRA", "").split()
args = bench.parse()
replace = int(os.environ.get("REPLACE", "8"))
dev = "cuda"
torch.manual_seed(123)
scene, loader, seg, val_cams, capture, frames = bench.build_scene(args, dev, 0, 1)
model, eng = bench.build_engine(args, dev, 0, 1, loader, frames, seg, 1337)
gc.collect(); gc.freeze()
if replace > 0:
    loader.start_replacer(replace)
total, every = int(os.environ.get("STEPS", "20000")), int(os.environ.get("EVERY", "2500"))
print("segments", list(seg), "training cameras", len(loader.camera_numbers), "validation cameras", val_cams, "replacements/step",
      replace if replace > 0 else "1/16 (synchronous)", "emb", args.emb, flush=True)
step = 0
while step < total:
    for i in range(every - 100):
        eng.train_iteration(); step += 1
        if replace == 0 and step % 16 == 15:
            eng.replace_next()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rays = samples = 0; sums = torch.zeros(3, device=dev)
    for i in range(100):
        st = eng.train_iteration(); step += 1
        rays += st.num_rays; samples += st.num_samples; sums += st.sums
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loader.pause_replacing()
    pf = loader.frame_numbers_cuda.cpu()
    vframe = int(torch.mode(pf[pf >= 0]).values)
    nv = int(os.environ.get("VIEWS", "8"))
    pairs = [(val_cams[i % len(val_cams)], vframe if i % 2 == 0 else scene.frame_numbers[(step // every * 7 + i * 17) % len(scene.frame_numbers)])
             for i in range(nv)]
    res = validate(model, loader, pairs, 65536)
    loader.continue_replacing()
    print("step %6d: %.2f Mray/s, %.2f ms/step, %.1f visible samples/ray, train PSNR %.2f dB, novel-view PSNR %s (mean %.2f dB), "
          "pairs loaded %d, skipped %d"
          % (step, rays / dt / 1e6, 1e3 * dt / 100, samples / rays, TrainEngine.psnr_from_sums(sums, rays),
             ["%.2f" % p for p in res["psnr"]], res["psnr_mean"], loader.replacements, eng.found_inf()), flush=True)
loader.stop_replacer()
