#!/usr/bin/env python3
"""bench.py -- training throughput of the HumanRF hot path on MI355X.

One "step" = one iteration of Trainer.train's loop body (humanrf/trainer.py:135-187): batch growing
(sampler + prune passes until >= 0.9 * samples_max_batch_size), merge, render, loss, backward, Adam.
Workload (BASELINE.json configs[1]): Actor01/Sequence1-shaped synthetic capture, 4x scale (752^2 centre crop),
50 frames (15..64), all 160 cameras, 256^3 occupancy grids, example_humanrf.py model
(log2_hashmap_size 19, adaptive temporal partitioning, camera_embedding_dim 2, samples_max_batch_size 640000,
rays_initial_batch_size 8192). Data: synthetic, weights: random init (no dataset / checkpoints offline).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run. W untimed steps, then exactly K timed steps bracketed by barrier + synchronize, MAX
over ranks, rank 0 prints ONE JSON line. value = rays entering train_step summed over ranks / time.

Before the W warm-up steps the model is trained for --pretrain untimed steps (default 3000): a step always renders
~640 k samples, so rays per step = 640 k / visible samples per ray, which falls from ~190 at random initialisation to
~10 after 3000 steps and 6-8 later; `regime_at_random_init` reports the same loop from step 3.
Extra objects on the line: `roofline` (fused prune march: algorithmic 2128 B per encoded sample / its launch time,
events on the launch stream; `traffic` from the PMC passes under profiles/), `cpu_baseline` (the oracle port on the
host cores, bounded sample, rank 0 at N = 1 only), `validation_psnr_db` (a novel view of a frame in training, rendered
through the inference path), `collector_iterations` (batch-growing iterations served from prefetched sampler stages)."""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ENC_BYTES_PER_SAMPLE = 2128    # 2048 B table reads + 16 B xyzt + 64 B features out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pretrain", type=int, default=3000,
                    help="untimed training steps before warmup: throughput depends on how sharp the density already is "
                         "(~230 visible samples/ray at init, ~10 once trained); the reference trains 50 001 steps, so the "
                         "trained regime is where a run spends its time. 0 = measure from random init.")
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--cameras", type=int, default=160)
    ap.add_argument("--image", type=int, default=752)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--log2-hashmap-size", type=int, default=19)
    ap.add_argument("--samples-max", type=int, default=640_000)
    ap.add_argument("--rays-initial", type=int, default=8192)
    ap.add_argument("--emb", type=int, default=2)
    ap.add_argument("--partitioning", default="adaptive", choices=["adaptive", "none"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-validation", action="store_true")
    ap.add_argument("--kernel-breakdown", action="store_true", help="time every kernel span (adds host overhead)")
    ap.add_argument("--cpu-rays", type=int, default=98304, help="rays drawn for the CPU baseline sample (~10 % survive the occupancy mask)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0 (with --backend gloo)")
    return ap.parse_args()


def cpu_baseline(model, loader, n_rays: int):
    """The CPU oracle (a port: the reference has no CPU path at all) timed on the host
    cores on a bounded sample of the same workload: one prune + render + loss + backward over `n_rays` drawn
    rays, every core torch can see."""
    from oracle import hrf_oracle as O
    from tests.util import oracle_model_from
    # torch's intra-op pool degrades badly past a few dozen threads on these small index/gather ops (measured:
    # 256 threads were >100x slower than 8); `cores` reports the threads actually used.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    loader.batch_size = n_rays
    ib = next(loader)
    om = oracle_model_from(model, requires_grad=True)
    o, d = ib.ray_origins.cpu(), ib.ray_directions.cpu()
    fr, cm, rgba = ib.frame_numbers.cpu(), ib.camera_numbers.cpu(), ib.rgba.cpu()
    t0s, ri = ib.sample_distances.cpu(), ib.ray_indices.cpu()
    g = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    jitter = torch.rand(t0s.shape[0], generator=g)
    t_j, _, vis, _ = O.prune_samples(om, o, d, fr, t0s, ri, jitter)
    t1, r1 = t_j[vis], ri[vis]
    bg = torch.rand(o.shape[0], 3, generator=g)
    color, acc = O.render(om, o, d, fr, cm, t1, r1, bg, True)
    loss, _ = O.training_loss(color, acc, rgba, bg)
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": o.shape[0] / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{o.shape[0]} rays ({t0s.shape[0]} pre-prune, {int(vis.sum())} post-prune samples): "
                      f"prune + render + loss + backward, no optimizer step, {dt:.1f} s"}


@torch.no_grad()
def validation_psnr(model, scene, camera: int, frame: int, batch: int = 16384):
    """Inference form of the path (trainer.py:283-308): full image in batches, background 0, PSNR vs ground truth."""
    from humanrf_amd.dataset import ray_sampler_native as rs
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.dataset.occupancy_grid_native import OccupanyGrid
    from humanrf_amd.volume_rendering import prune_samples, render
    dev = scene.device
    P = scene.width * scene.height
    rgba = scene.render_rgba(camera, frame)
    ring = OccupanyGrid(scene.grid_resolution, 1)
    tex = torch.tensor([ring.add_grid(scene.occupancy_grid(frame))], dtype=torch.int64, device=dev)
    se, n = 0.0, 0
    for s in range(0, P, batch):
        idx = torch.arange(s, min(s + batch, P), dtype=torch.int64, device=dev)
        out = rs.get_samples_occupancy_minmax(
            rgba, torch.zeros(P, dtype=torch.bool, device=dev), torch.tensor([frame], dtype=torch.int32, device=dev),
            torch.tensor([camera], dtype=torch.int32, device=dev), tex, torch.ones(1, dtype=torch.bool, device=dev), idx,
            scene.all_inverse_krs[camera:camera + 1].contiguous(), scene.all_camera_origins[camera:camera + 1].contiguous(),
            scene.aabb, scene.grid_resolution, scene.width, scene.height, 4e-4, False)
        ib = InputBatch(ray_origins=out[0], ray_directions=out[1], rgba=out[2], frame_numbers=out[3].view(-1, 1),
                        camera_numbers=out[4].view(-1, 1), minmaxes=out[5], ray_masks=out[6].view(-1, 1),
                        sample_distances=out[7].view(-1, 1), ray_indices=out[8].long(),
                        unique_frame_numbers=out[3][:1].view(-1, 1))
        if ib.num_rays == 0:
            continue
        prune_samples(ib, model, False)
        ro = render(ib, model, 0.0, False)
        gt = ib.rgba[:, :3] * ib.rgba[:, 3:4]            # evaluate_one_image: gt blended onto background 0 (trainer.py:383-385)
        se += float(torch.square(ro.color - gt).sum())   # psnr over the rendered (ray-masked) rays, trainer.py:218-223,389
        n += 3 * ib.num_rays
    import math
    return -10.0 * math.log10(max(se / max(n, 1), 1e-20))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from humanrf_amd import _lib, ops
    _lib.lib()
    from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader, SyntheticScene
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine

    torch.manual_seed(123 + rank)  # run_args.py:125; per-rank stream for ray sharding
    frames = tuple(range(15, 15 + args.frames))  # presets.py:41
    scene = SyntheticScene(frames, num_cameras=args.cameras, width=args.image, height=args.image,
                           grid_resolution=args.grid, device=dev)
    if args.partitioning == "adaptive":
        segment_sizes = compute_adaptive_segment_sizes(scene.occupancy_grid, list(frames), 1.25)
    else:
        segment_sizes = [len(frames)]
    model = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2,
                    log2_hashmap_size=args.log2_hashmap_size, n_levels=16, coarsest_resolution=32,
                    finest_resolution=2048, geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1,
                    n_hidden_layers_color=2, sh_degree=4, segment_sizes=tuple(segment_sizes),
                    camera_embedding_dim=args.emb, device=dev, seed=1337)  # identical replicas on every rank
    loader = SyntheticDataLoader(scene, batch_size=args.rays_initial, max_buffer_size=200, max_num_frames_per_batch=8,
                                 seed=123 + rank)
    iter(loader)
    eng = TrainEngine(model, loader, samples_max_batch_size=args.samples_max, rays_initial_batch_size=args.rays_initial,
                      world_size=world)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The scene / model / library objects built above are long-lived: park them in the permanent generation so the
    # cyclic collector's periodic full passes do not stall the launch thread for ~10 ms in the middle of a step
    # (measured: 3 such stalls per 60 steps, always at the same launch).
    gc.collect()
    gc.freeze()
    # the same loop measured from random initialisation too (sigma ~ 100 everywhere: a ray keeps ~230
    # samples, so a 640 k-sample step holds ~3 k rays). Reported next to the headline regime, not instead of it.
    init_regime = None
    if args.pretrain >= 16:
        for _ in range(3):
            eng.train_iteration()
        sync()
        t_i = time.perf_counter()
        r_i = s_i = 0
        for _ in range(8):
            st = eng.train_iteration()
            r_i += st.num_rays; s_i += st.num_samples
        sync()
        dt_i = time.perf_counter() - t_i
        init_regime = {"rays_per_s_this_rank": round(r_i / dt_i, 1), "ms_per_step": round(1e3 * dt_i / 8, 3),
                       "samples_per_ray_post": round(s_i / max(r_i, 1), 1), "steps_trained_before": 3}
    for i in range(max(args.pretrain - 11, 0) + args.warmup if init_regime else args.pretrain + args.warmup):
        eng.train_iteration()
        if i % 16 == 15:
            eng.replace_next()  # pool replacement (the reference's replacer thread), outside the timed region
    sync()
    ops.TIMER = ops.KernelTimer(None if args.kernel_breakdown else {"prune_march", "encode4d_fwd"})
    eng.evaluated.zero_()
    if eng.collector is not None:
        eng.collector.evaluated.zero_()
        eng.collector.iterations_prefetched = eng.collector.iterations_classic = 0
    rays = rays_drawn = n0 = n1 = 0
    sums = torch.zeros(3, device=dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = eng.train_iteration()
        rays += st.num_rays; rays_drawn += st.num_rays_drawn; n0 += st.num_samples_pre; n1 += st.num_samples
        sums += st.sums
    sync()
    dt = time.perf_counter() - t0
    timer = ops.TIMER.summary()
    ops.TIMER = None
    skipped = eng.found_inf()

    n0 = int(n0.item()) if torch.is_tensor(n0) else n0
    n_eval = int(eng.evaluated.item()) + (int(eng.collector.evaluated.item()) if eng.collector is not None else 0)
    stat = torch.tensor([dt, rays, rays_drawn, n0, n1], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        mx = stat.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
        stat[0] = mx[0]
    dt_max, rays_all, drawn_all, n0_all, n1_all = [float(x) for x in stat.tolist()]

    if rank == 0:
        # dominant gather kernel: the fused prune march (its encode stage); algorithmic bytes = samples it actually
        # encoded x 2 128 B. Falls back to the stand-alone encode kernel when fusion is off.
        roofline = None
        enc = timer.get("prune_march")
        kname = "k_prune_march (hash gather + sigma_net + visibility, prune pass)"
        if enc is not None:
            enc = dict(enc, units=n_eval)
        else:
            enc = timer.get("encode4d_fwd", {"ms_total": 0.0, "units": 0, "launches": 0})
            kname = "k_encode4d_fwd (prune pass)"
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if enc["ms_total"] > 0 and kname.startswith("k_prune_march") and os.path.exists(tj):
            # HBM-side bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE per
            # encoded sample, collected in separate rocprofv3 --pmc runs) x the samples one launch encoded here
            t = json.load(open(tj))["k_prune_march"]
            traffic = round((t["fetch_bytes_per_encoded_sample"] + t["write_bytes_per_encoded_sample"]) * enc["units"] /
                            max(enc["launches"], 1))
        if enc["ms_total"] > 0:
            achieved = enc["units"] * ENC_BYTES_PER_SAMPLE / (enc["ms_total"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE)",
                        "algorithmic_bytes_per_launch": round(enc["units"] * ENC_BYTES_PER_SAMPLE / max(enc["launches"], 1)),
                        "launches": enc["launches"],
                        "avg_launch_ms": round(enc["ms_total"] / max(enc["launches"], 1), 4),
                        "algorithmic_bytes_per_sample": ENC_BYTES_PER_SAMPLE}
        breakdown = {k: round(v["ms_total"] / args.steps, 3) for k, v in sorted(timer.items())}
        out = {
            "metric": "training rays/sec", "value": round(rays_all / dt_max, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "pretrain_steps": args.pretrain,
            "ms_per_step": round(1e3 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 tables/MLP operands, f32 accumulate + master weights",
            "data": f"synthetic ActorsHQ-shaped scene, random-init weights trained for {args.pretrain + args.warmup} steps "
                    "before the timed region",
            "config": {"workload": f"Actor01/Sequence1-shaped {({752: '4x', 3008: '1x'}).get(args.image, 'custom scale')}, {args.frames} frames, {args.cameras} cams, "
                                   f"{args.image}^2 px, grid {args.grid}^3, segments {list(segment_sizes)}, "
                                   f"log2_T {args.log2_hashmap_size}, emb {args.emb}",
                       "samples_max_batch_size": args.samples_max, "rays_initial_batch_size": args.rays_initial,
                       "parallelism": f"ray-sharded dp{world}"},
            "rays_drawn_per_s": round(drawn_all / dt_max, 1),
            "samples_pre_prune_per_s": round(n0_all / dt_max, 1), "samples_post_prune_per_s": round(n1_all / dt_max, 1),
            "samples_encoded_by_prune_per_s": round(n_eval / dt_max, 1),
            "samples_per_ray_pre": round(n0_all / max(drawn_all, 1), 2), "samples_per_ray_post": round(n1_all / max(rays_all, 1), 2),
            "train_psnr_db": round(TrainEngine.psnr_from_sums(sums, max(rays, 1)), 3),
            "skipped_step_flag": bool(skipped),
            "kernel_ms_per_step": breakdown,
            "roofline": roofline,
        }
        if init_regime is not None:
            out["regime_at_random_init"] = init_regime
        if eng.collector is not None:  # batch-growing iterations served from the prefetched sampler stages vs classic ones
            out["collector_iterations"] = {"prefetched": eng.collector.iterations_prefetched,
                                           "classic": eng.collector.iterations_classic}
        if not args.no_validation:
            # a view the training never saw, of a frame it is currently training on (novel-view validation): the
            # frame most present in the pool, the first camera that is not in the pool for that frame
            pf, pc = loader.frame_numbers_cuda.cpu(), loader.camera_numbers_cuda.cpu()
            vframe = int(torch.mode(pf[pf >= 0]).values)
            seen = set(pc[pf == vframe].tolist())
            vcam = next(c for c in range(args.cameras) if c not in seen)
            out["validation_psnr_db"] = round(validation_psnr(model, scene, vcam, vframe), 3)
            out["validation_view"] = {"camera": vcam, "frame": vframe, "in_training_pool": False}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, loader, args.cpu_rays)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
