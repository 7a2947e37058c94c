"""Run-to-run spread of the novel-view PSNR at the headline regime (configs[1] sizes, 2 080 steps), by variant of the step.
One scene and one resident capture are built once; every run then starts from the same seeds (model 1337, loader 123, torch
123), so whatever differs between two runs of one variant comes from the step itself (order of the fp32 atomics, timing of the
pool replacer). Each run reports

  novel   PSNR of the held-out validation views (evaluation mode: zero camera embedding, humanrf.py:196-204)
  t_eval  PSNR of a TRAINING camera rendered the same way (zero embedding)
  t_emb   PSNR of the same training camera rendered with its own embedding
  e_rms   rms of the training cameras' embeddings

usage: python tools/psnr_variance.py VARIANT:RUNS [VARIANT:RUNS ...]
variants: default | atomic (table_scatter=atomic, frame-ordered batch) | r2path (atomic scatter, batch not frame-ordered) |
          emb0 (camera_embedding_dim 0) | static (no pool replacement) | fp16b (gradient_boundaries="fp16": the reference's half
          gradient tensors between modules, include/hrf.h grad_boundary) | emb0fp16b (both) | fp16bmlp / fp16btab (only the MLP
          backward's boundaries / only the table scatter's boundary rounded through half)"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    plan = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]] or [("default", 3)]
    steps = int(os.environ.get("STEPS", "2080"))
    sys.argv = sys.argv[:1]
    args = bench.parse()
    from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
    from humanrf_amd.dataset.synthetic import ResidentCapture, SyntheticDataLoader, SyntheticScene
    from humanrf_amd.inference import psnr_of_rendered_rays, render_image, validate
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine
    from humanrf_amd.volume_rendering import RenderOutput, prune_samples, render
    from humanrf_amd.dataset.input_batch import InputBatch
    dev = "cuda"
    frames = tuple(range(15, 15 + args.frames))
    scene = SyntheticScene(frames, num_cameras=args.cameras, width=args.image, height=args.image, grid_resolution=args.grid,
                           device=dev)
    segs = tuple(compute_adaptive_segment_sizes(scene.occupancy_grid, list(frames), 1.25))
    val_cams = [c for c in bench.VALIDATION_CAMERAS if c < args.cameras]
    train_cams = [c for c in range(args.cameras) if c not in val_cams]
    capture = ResidentCapture(scene, list(range(args.cameras)))
    print("segments", list(segs), "validation cameras", val_cams[:4], flush=True)

    @torch.no_grad()
    def with_embedding(model, loader, cam, frame):
        model.eval()
        parts_b, parts_o = [], []
        for b in loader.validation_batches(cam, frame, 65536):
            parts_b.append(InputBatch(ray_masks=b.ray_masks, rgba=b.rgba, width=b.width, height=b.height))
            if b.num_rays == 0:
                parts_o.append(RenderOutput(color=torch.zeros(0, 3, device=dev), weights_sum=torch.zeros(0, 1, device=dev)))
                continue
            prune_samples(b, model, False)
            parts_o.append(render(b, model, 0.0, True))      # is_training=True: the camera's own embedding, no jitter here
        model.train()
        full = InputBatch(ray_masks=torch.cat([b.ray_masks for b in parts_b], 0), rgba=torch.cat([b.rgba for b in parts_b], 0),
                          width=parts_b[0].width, height=parts_b[0].height)
        return psnr_of_rendered_rays(RenderOutput.merge_render_outputs(parts_o), full.rgba, 0.0)

    for variant, runs in plan:
        for r in range(runs):
            torch.manual_seed(123)
            emb = 0 if variant.startswith("emb0") else args.emb
            model = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2,
                            log2_hashmap_size=args.log2_hashmap_size, n_levels=16, coarsest_resolution=32,
                            finest_resolution=2048, geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1,
                            n_hidden_layers_color=2, sh_degree=4, segment_sizes=segs, camera_embedding_dim=emb, device=dev,
                            seed=1337)
            loader = SyntheticDataLoader(scene, batch_size=args.rays_initial, camera_numbers=train_cams, max_buffer_size=200,
                                         max_num_frames_per_batch=8, seed=123, camera_seed=123, capture=capture,
                                         frame_synchronous=True)
            iter(loader)
            eng = TrainEngine(model, loader, samples_max_batch_size=args.samples_max, rays_initial_batch_size=args.rays_initial,
                              table_scatter="atomic" if variant in ("atomic", "r2path") else "auto",
                              gradient_boundaries="fp16" if "fp16b" in variant else "fp32")   # (explicit: the engine's default is fp16)
            if variant.endswith("fp16bmlp"):      # only the MLP backward's two boundaries (dL/d(sigma_net output), dL/d(features))
                eng._gb_tables = 0.0
            elif variant.endswith("fp16btab"):    # only the compose op's per-encoding gradients (the table scatter's boundary)
                eng._gb_mlp = 0.0
            if variant == "r2path":
                eng.collector.sort_batch = False
            if variant != "static":
                loader.start_replacer(args.replacements_per_step)
            t0 = time.perf_counter()
            rays = samples = 0
            sums = torch.zeros(3, device=dev)
            for i in range(steps):
                st = eng.train_iteration()
                if i >= steps - 60:
                    rays += st.num_rays; samples += st.num_samples; sums += st.sums
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            loader.pause_replacing()
            pf = loader.frame_numbers_cuda.cpu()
            vframe = int(torch.mode(pf[pf >= 0]).values)
            pairs = [(val_cams[i % len(val_cams)], vframe if i % 2 == 0 else scene.frame_numbers[(i * 17) % len(scene.frame_numbers)])
                     for i in range(int(os.environ.get("VIEWS", "8")))]
            res = validate(model, loader, pairs, rays_batch_size=65536)
            tcam = loader.camera_numbers[0]
            t_eval = validate(model, loader, [(tcam, vframe)], 65536)["psnr_mean"]
            t_emb = with_embedding(model, loader, tcam, vframe) if emb > 0 else float("nan")
            e_rms = float(model.camera_embeddings.weight.detach()[torch.tensor(train_cams, device=dev)].pow(2).mean().sqrt()) \
                if emb > 0 else 0.0
            print("%-9s run %2d: novel %.2f dB %s, t_eval %.2f, t_emb %.2f, e_rms %.3f, train PSNR %.2f, %.1f samples/ray, "
                  "%.1f s, skipped %d" % (variant, r, res["psnr_mean"], ["%.1f" % p for p in res["psnr"]], t_eval, t_emb, e_rms,
                                          TrainEngine.psnr_from_sums(sums, rays), samples / max(rays, 1), dt, eng.found_inf()),
                  flush=True)
            loader.stop_replacer()
            del eng, loader, model
            gc.collect()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
