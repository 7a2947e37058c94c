#!/usr/bin/env python3
"""bench.py -- training throughput of the HumanRF hot path on MI355X.

One "step" = one iteration of Trainer.train's loop body (humanrf/trainer.py:135-187): batch growing
(sampler + prune passes until >= 0.9 * samples_max_batch_size), merge, render, loss, backward, Adam -- with the loader's
replacer thread refilling pool slots while the steps run (data_loader.py:396-422).
Workload (BASELINE.json configs[1]): Actor01/Sequence1-shaped synthetic capture, 4x scale (752^2 centre crop),
50 frames (15..64), the 160-camera rig, 256^3 occupancy grids, example_humanrf.py model (log2_hashmap_size 19, adaptive
temporal partitioning, camera_embedding_dim 2, samples_max_batch_size 640000, rays_initial_batch_size 8192). The
reference's ten validation cameras (presets.py "siggraph_train_validation") are held out of training, as its example
configuration does, so that validation views are novel. Data: synthetic, weights: random init (no dataset / checkpoints
offline). Other configurations: --partitioning none (one 2^18 segment), --segment-size 100 (2^19 tables),
--frames 250 (configs[3] shape on one GPU), --image 3008 (configs[2]).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run -- and when it does not (the plain command with N > 1 and no WORLD_SIZE around it) bench.py starts the N
ranks itself, through the same launcher (self_launch). W untimed steps, then exactly K timed steps bracketed by barrier + synchronize, MAX over ranks,
rank 0 prints ONE JSON line. value = rays entering train_step summed over ranks / time.

Regime. A step always renders ~640 k samples, so rays per step = 640 k / (visible samples per ray), which falls from
~190 at random initialisation to ~15 after 2 000 steps and 6-8 from 5 000 steps on. The headline `value`
is measured after --pretrain untimed training steps, default 2 000 = the point SURVEY.md 8(d) names; `regime_curve`
carries the same measurement from random initialisation, at the headline point and (when --curve allows) further on.
Extra objects on the line: `roofline` (whichever of the three gather / scatter kernels takes the most time per step:
algorithmic bytes / its launch time, events on the launch stream), `roofline_kernels` (all three), `step_algorithmic`
(the whole step's algorithmic bytes against the HBM peak), `cpu_baseline` (the oracle port on the
host cores, bounded sample, rank 0 at N = 1 only), `validation` (novel views rendered through
humanrf_amd.inference.validate), `collector_iterations`, `replacer`."""
import argparse
import gc
import json
import os
import sys
import time

# RCCL / tensor sharing across the ranks' processes needs dmabuf IPC on this pool's hosts (the image exports this already; a launcher
# that scrubs the environment must not take it away). Set before the HIP runtime comes up.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA (no sparsity)
SIGMA_FLOPS = 2 * (32 * 64 + 64 * 16)          # sigma_net forward, 6 144 flop / sample
ENC_BYTES_PER_SAMPLE = 2128    # 2048 B table reads + 16 B xyzt + 64 B features out
BWD_BYTES_PER_SAMPLE = 4176    # 64 B dY + 16 B + 2 x 2048 B read-modify-write
VALIDATION_CAMERAS = (10, 19, 33, 44, 50, 73, 83, 90, 104, 117)   # presets.py "siggraph_train_validation"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pretrain", type=int, default=2000,
                    help="untimed training steps before the warm-up and the timed steps (SURVEY.md 8(d): 2 000). "
                         "0 = measure from random initialisation.")
    ap.add_argument("--trials", type=int, default=3,
                    help="independent training trajectories (model seed, ray / background seed) the headline is taken over: "
                         "rays/s = 640 k / (visible samples per ray) / step time, and the samples per ray a model has reached after "
                         "--pretrain steps vary from run to run by more than the kernels do; `value` is the MEDIAN trial's")
    ap.add_argument("--curve", default="5000", help="comma-separated later points of the regime curve (total steps trained); "
                                                    "'' = none")
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--cameras", type=int, default=160)
    ap.add_argument("--image", type=int, default=752)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--log2-hashmap-size", type=int, default=19)
    ap.add_argument("--samples-max", type=int, default=640_000)
    ap.add_argument("--rays-initial", type=int, default=8192)
    ap.add_argument("--emb", type=int, default=2)
    ap.add_argument("--partitioning", default="adaptive", choices=["adaptive", "none", "fixed"])
    ap.add_argument("--segment-size", type=int, default=100, help="segment size of --partitioning fixed")
    ap.add_argument("--train-all-cameras", action="store_true", help="do not hold the validation cameras out of training")
    ap.add_argument("--replacements-per-step", type=int, default=8,
                    help="pool slots the replacer thread refills per training step (200-slot pool: one full turnover "
                         "every 25 steps; the reference's thread is paced by JPEG decoding)")
    ap.add_argument("--capture-budget-gb", type=float, default=128.0,
                    help="keep the whole capture resident in HBM when it fits this budget (4x, 50 frames: 18 GB)")
    ap.add_argument("--host-capture-gb", type=float, default=48.0,
                    help="captures beyond --capture-budget-gb: pinned host memory for the images the replacer streams from "
                         "(as many training cameras as fit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the two short secondary legs (bf16 MLP on the headline workload; the configs[2] shape, 1x scale 3008^2 px "
                         "from a pinned host capture) that ride on the default N = 1 line as `other_configs`")
    ap.add_argument("--other-pretrain", type=int, default=600, help="training steps before the 20 timed steps of a secondary leg")
    ap.add_argument("--no-validation", action="store_true")
    ap.add_argument("--validation-views", type=int, default=8, help="held-out (camera, frame) pairs rendered for the PSNR half of the metric")
    ap.add_argument("--kernel-breakdown", action="store_true", help="time every kernel span (adds host overhead)")
    ap.add_argument("--kernel-window", type=int, default=10,
                    help="steps of the window AFTER the timed region in which every kernel of roofline_kernels is bracketed by events")
    ap.add_argument("--cpu-rays", type=int, default=98304, help="rays drawn for the CPU baseline sample (~10 %% survive the occupancy mask)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--transport", default="fp32", choices=["fp32", "bf16"], help="wire format of the table-gradient exchange")
    ap.add_argument("--mlp-precision", default="fp16", choices=["fp16", "bf16"],
                    help="arithmetic type of the two MLPs (fp16 = the reference's tcnn configuration; bf16 = BASELINE.json configs[4])")
    ap.add_argument("--ab-pieces", default="", help="measurement aid: after the timed region, alternate TrainEngine.pipeline_pieces "
                    "over this comma-separated list (3 rounds x 40 steps each) and print ms/step per setting to stderr")
    ap.add_argument("--ab-overlap-vectors", action="store_true", help="measurement aid: after the timed region, alternate "
                    "TrainEngine.overlap_vector_scatter on / off on the same trajectory (40-step windows, four rounds)")
    ap.add_argument("--ab-env", default="", help="measurement aid: after the timed region, alternate this environment switch of the library "
                    "between 0 and 1 on the same trajectory (40-step windows, four rounds)")
    ap.add_argument("--ab-main-priority", action="store_true", help="measurement aid: after the timed region, alternate running the "
                    "training loop on torch's default stream and on a HIGH-priority stream (the sampler prefetch, the replacer and the "
                    "vector-gradient kernel stay on their normal-priority streams), 40-step windows, four rounds")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--launch-probe", action="store_true",
                    help="testing only: bring the ranks up, all-reduce a one, print the one line and leave (no GPU needed)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank keeps the reference's sample budget (global batch = N x the reference's); strong: the "
                         "budget is divided by N (global batch = the reference's, trainer.py:156-172)")
    ap.add_argument("--exchange", default="sharded", choices=["sharded", "allreduce"],
                    help="N > 1: table gradients by reduce-scatter + sharded Adam + all-gather of the fp16 tables, or by all-reduce")
    ap.add_argument("--exchange-groups", type=int, default=4,
                    help="N > 1: the table gradients are accumulated and exchanged in up to this many groups of temporal segments, "
                         "each group's collective under the accumulation of the next (1 = exchange after the whole scatter)")
    ap.add_argument("--no-exchange-signals", action="store_true",
                    help="N > 1: one accumulate launch PER segment group instead of one launch that signals each group's completion "
                         "to the stream its collective is issued from (TrainEngine.exchange_signalled)")
    ap.add_argument("--mlp-backward", default="fused", choices=["fused", "split"],
                    help="backward of the two MLPs as one kernel or as colour + density kernels (TrainEngine.mlp_backward)")
    ap.add_argument("--no-overlap-vectors", action="store_true",
                    help="measurement aid: run the vector-gradient scatter behind the table-gradient scatter instead of under it")
    ap.add_argument("--gradient-boundaries", default="fp16", choices=["fp32", "fp16"],
                    help="fp16 (default): round the gradient through half where the reference's modules hand each other half tensors "
                         "(TrainEngine.gradient_boundaries, include/hrf.h grad_boundary); fp32: keep fp32 from the loss to the tables")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the data-parallel step with every torch.distributed collective of it on whatever group exists, "
                         "even a group of ONE rank (degenerate but real RCCL calls): exercises the N > 1 code path on a one-GPU box")
    ap.add_argument("--table-scatter", default="auto", choices=["auto", "binned", "atomic"],
                    help="table-gradient scatter: radix partition + LDS accumulation (csrc/scatter.hip) or level-major atomics")
    return ap.parse_args()


def kernel_source_fingerprint() -> str:
    """SHA-256 over the kernel sources the library is built from (what a PMC summary under profiles/ is valid for)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "humanrf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def cpu_baseline(model, loader, n_rays: int):
    """The CPU oracle (a port: the reference has no CPU path at all) timed on the host
    cores on a bounded sample of the same workload: one prune + render + loss + backward over `n_rays` drawn
    rays, every core torch can see."""
    from oracle import hrf_oracle as O
    from tests.util import oracle_model_from
    # torch's intra-op pool degrades badly past a few dozen threads on these small index/gather ops (measured:
    # 256 threads were >100x slower than 8); `cores` reports the threads actually used.
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 16)
    torch.set_num_threads(cores)
    loader.batch_size = n_rays
    ib = next(loader)
    om = oracle_model_from(model, requires_grad=True)
    o, d = ib.ray_origins.cpu(), ib.ray_directions.cpu()
    fr, cm, rgba = ib.frame_numbers.cpu(), ib.camera_numbers.cpu(), ib.rgba.cpu()
    t0s, ri = ib.sample_distances.cpu(), ib.ray_indices.cpu()
    g = torch.Generator().manual_seed(0)
    params = [p for seg in om.tables for p in seg] + list(om.vectors) + list(om.sigma_w) + list(om.color_w)
    if om.camera_embeddings is not None:
        params.append(om.camera_embeddings)
    opt = torch.optim.Adam(params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)   # run.py:101
    jitter = torch.rand(t0s.shape[0], generator=g)
    # the pruning pass (no gradient: 96 % of the encoded samples) twice: on torch's thread pool, and with the oracle's OpenMP restatement
    # of the Decomposition4D forward (oracle/encode_oracle.c, bit-identical) on EVERY host core -- SURVEY.md 8(d)'s "OpenMP C++ for the
    # gather"; the render + backward + Adam part runs once, on torch
    t0 = time.perf_counter()
    t_j, _, vis, _ = O.prune_samples(om, o, d, fr, t0s, ri, jitter)
    dt_prune_torch = time.perf_counter() - t0
    O.C_ENCODE_THREADS = host_cores
    try:
        t0 = time.perf_counter()
        t_c, _, vis_c, _ = O.prune_samples(om, o, d, fr, t0s, ri, jitter)
        dt_prune_omp = time.perf_counter() - t0
    finally:
        O.C_ENCODE_THREADS = 0
    same = bool(torch.equal(vis, vis_c))
    t0 = time.perf_counter()
    t1, r1 = t_j[vis], ri[vis]
    bg = torch.rand(o.shape[0], 3, generator=g)
    color, acc = O.render(om, o, d, fr, cm, t1, r1, bg, True)
    loss, _ = O.training_loss(color, acc, rgba, bg)
    loss.backward()
    opt.step()
    dt_rest = time.perf_counter() - t0
    dt_torch, dt_omp = dt_prune_torch + dt_rest, dt_prune_omp + dt_rest
    return {"value": o.shape[0] / dt_omp, "unit": "rays/s", "cores": host_cores, "host_cores": host_cores, "kind": "port",
            "sample": f"{o.shape[0]} rays ({t0s.shape[0]} pre-prune, {int(vis.sum())} post-prune samples): prune + render + loss + backward + "
                      f"Adam step over every table; pruning pass {dt_prune_omp:.1f} s with the OpenMP gather on all {host_cores} host cores "
                      f"(same survivors as the torch pass: {same}), render / backward / Adam {dt_rest:.1f} s on {cores} torch threads",
            "torch_only": {"value": o.shape[0] / dt_torch, "cores": cores,
                           "note": f"the same sample with the pruning pass on torch's thread pool as well ({dt_prune_torch:.1f} s; the pool is "
                                   "slower beyond 16 threads on these gather-shaped ops): the figure rounds 1-5 reported"}}


@torch.no_grad()
def own_embedding_psnr(model, loader, cam, frame) -> float:
    """PSNR of a TRAINING camera's full image rendered with its own camera embedding (render(..., is_training=True) after an
    un-jittered prune): the number the zero-embedding validation render of the same camera is to be read against."""
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.inference import psnr_of_rendered_rays
    from humanrf_amd.volume_rendering import RenderOutput, prune_samples, render
    was_training = model.training
    model.eval()
    batches, outs = [], []
    try:
        for b in loader.validation_batches(cam, frame, 65536):
            batches.append(InputBatch(ray_masks=b.ray_masks, rgba=b.rgba, width=b.width, height=b.height))
            if b.num_rays == 0:
                dev = b.ray_masks.device
                outs.append(RenderOutput(color=torch.zeros(0, 3, device=dev), weights_sum=torch.zeros(0, 1, device=dev)))
                continue
            prune_samples(b, model, False)
            outs.append(render(b, model, 0.0, True))
    finally:
        model.train(was_training)
    rgba = torch.cat([b.rgba for b in batches], 0)
    return psnr_of_rendered_rays(RenderOutput.merge_render_outputs(outs), rgba, 0.0)


def build_scene(args, dev, rank, world):
    """Scene, capture, loader, segment sizes: shared by every trial."""
    from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
    from humanrf_amd.dataset.synthetic import HostCapture, ResidentCapture, SyntheticDataLoader, SyntheticScene
    frames = tuple(range(15, 15 + args.frames))  # presets.py:41
    scene = SyntheticScene(frames, num_cameras=args.cameras, width=args.image, height=args.image,
                           grid_resolution=args.grid, device=dev)
    if args.partitioning == "adaptive":
        segment_sizes = compute_adaptive_segment_sizes(scene.occupancy_grid, list(frames), 1.25)
    elif args.partitioning == "fixed":   # run.py:52-56
        segment_sizes = [args.segment_size] * ((len(frames) + args.segment_size - 1) // args.segment_size)
    else:
        segment_sizes = [len(frames)]
    val_cams = [c for c in VALIDATION_CAMERAS if c < args.cameras]
    train_cams = [c for c in range(args.cameras) if args.train_all_cameras or c not in val_cams]
    capture = None
    all_cams = list(range(args.cameras))
    if ResidentCapture.fits(scene, len(all_cams), int(args.capture_budget_gb * 2 ** 30)):
        capture = ResidentCapture(scene, all_cams)
    else:
        # the capture does not fit the HBM budget (1x scale: 290 GB; 1 000 frames: 361 GB): images go to pinned host memory, the
        # replacer's kernel reads them over the host link (HostCapture). When the host budget does not hold the whole rig either,
        # training uses the evenly spaced subset of the training cameras that fits.
        n_fit = HostCapture.cameras_that_fit(scene, int(args.host_capture_gb * 2 ** 30))
        if n_fit >= 4:
            if n_fit < len(train_cams):
                train_cams = [train_cams[(i * len(train_cams)) // n_fit] for i in range(n_fit)]
            capture = HostCapture(scene, train_cams)
    # data parallel: shared frame schedule, per-rank camera order and per-rank ray draws (torch seed in main)
    loader = SyntheticDataLoader(scene, batch_size=args.rays_initial, camera_numbers=train_cams, max_buffer_size=200,
                                 max_num_frames_per_batch=8, seed=123, camera_seed=123 + rank, capture=capture,
                                 frame_synchronous=True)
    iter(loader)
    return scene, loader, segment_sizes, val_cams, capture, frames


def build_engine(args, dev, rank, world, loader, frames, segment_sizes, model_seed):
    """One trajectory's model (random init from `model_seed`, identical on every rank) and training engine."""
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine
    model = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2,
                    log2_hashmap_size=args.log2_hashmap_size, n_levels=16, coarsest_resolution=32,
                    finest_resolution=2048, geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1,
                    n_hidden_layers_color=2, sh_degree=4, segment_sizes=tuple(segment_sizes),
                    camera_embedding_dim=args.emb, device=dev, seed=model_seed,  # identical replicas on every rank
                    mlp_precision=args.mlp_precision)
    transport = torch.bfloat16 if args.transport == "bf16" else torch.float32
    # --scaling strong: the reference's global sample budget is divided over the ranks (trainer.py:156-172 with
    # samples_max / N per rank); weak: every rank keeps the whole budget
    per_rank = args.samples_max if args.scaling == "weak" else max(args.samples_max // world, 16_384)
    exchange = args.exchange if transport == torch.float32 else "allreduce"
    def make(exchange_mode):
        return TrainEngine(model, loader, samples_max_batch_size=per_rank, rays_initial_batch_size=args.rays_initial,
                           world_size=world, transport_dtype=transport, table_scatter=args.table_scatter, exchange=exchange_mode,
                           force_collectives=args.force_collectives, gradient_boundaries=args.gradient_boundaries,
                           overlap_vector_scatter=not args.no_overlap_vectors, mlp_backward=args.mlp_backward,
                           exchange_groups=args.exchange_groups)
    if getattr(args, "exchange_fallback", None):
        exchange = "allreduce"
    try:
        eng = make(exchange)
    except RuntimeError as e:
        # The engine refuses to train on in-place collectives that fail its start-up probe, on every rank at once
        # (TableShardExchange.self_check). The bench then measures the exchange every backend has -- one all-reduce -- and says so
        # on its line instead of producing no number.
        if "self_check" not in str(e) or exchange != "sharded":
            raise
        args.exchange_fallback = str(e)
        print(f"[bench] sharded exchange refused ({e}); falling back to --exchange allreduce", file=sys.stderr, flush=True)
        eng = make("allreduce")
    eng.exchange_signalled = not args.no_exchange_signals
    return model, eng


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start the N ranks here --
    the same command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on a free port of 127.0.0.1, environment
    kept (HSA_ENABLE_IPC_MODE_LEGACY=0 included) -- and hand on its exit code. The ranks inherit this process's stdout, so the ONE
    JSON line rank 0 prints is this command's one line; the torchrun form the contract names keeps working (WORLD_SIZE is set
    there and this function is never reached)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")      # (torchrun would set 1 and warn: the replacer thread and the CPU side of a rank want a few)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks with torch.distributed.run on 127.0.0.1:{port}",
          file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def launch_probe(args, rank, world, json_fd) -> None:
    """--launch-probe (testing only, runs without a GPU): every rank joins the process group, the ranks all-reduce a one, rank 0
    prints the one JSON line. What tests/test_cpu_bench_contract.py drives through the plain `python bench.py --gpus 2` form."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group(args.backend if args.backend != "nccl" or torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
    one = torch.ones(1)
    dist.all_reduce(one)
    dist.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps({"launch_probe": True, "n_gpus": args.gpus, "world_size": world, "ranks_seen": int(one.item()),
                                       "backend": str(dist.get_backend())}) + "\n").encode())
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE line, the JSON: whatever libraries write to file descriptor 1 (RCCL prints a version banner when
    # its first communicator is created) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.launch_probe:
        return launch_probe(args, rank, world, json_fd)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1 or args.force_collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from humanrf_amd import _lib, ops
    _lib.lib()
    from humanrf_amd.inference import validate
    from humanrf_amd.trainer import TrainEngine

    torch.manual_seed(123 + rank)  # run_args.py:125; per-rank stream for ray sharding
    t_setup = time.perf_counter()
    scene, loader, segment_sizes, val_cams, capture, frames = build_scene(args, dev, rank, world)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    dp = world > 1 or args.force_collectives

    def sync():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loader.start_replacer(args.replacements_per_step)   # refills pool slots while the steps below run

    from types import SimpleNamespace

    def train(tr, n):
        for _ in range(n):
            tr.eng.train_iteration()
        tr.trained += n

    def measure(tr, n_steps, timed_kernels=()):
        """n_steps timed iterations bracketed by barrier + synchronize -> dict of raw counts (this rank).
        timed_kernels: span names timed with events on the launch stream; None = every span; () = none."""
        eng = tr.eng
        sync()
        if timed_kernels is None or len(timed_kernels):
            ops.TIMER = ops.KernelTimer(timed_kernels)
        col = eng.collector
        tot0 = col.totals.clone() if col is not None else None
        it0 = (col.iterations_prefetched, col.iterations_classic) if col is not None else (0, 0)
        sp0 = (col.march_launches, col.march_launch_rays, col.rays_used) if col is not None else (0, 0, 0)
        ld = getattr(tr, "loader", loader)
        rep0 = ld.replacements
        eng.exchange_events, eng.time_exchange = [], dp      # data parallel: an event pair per step around the gradient exchange
        rays = drawn = n1 = 0
        sums = torch.zeros(3, device=dev)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            st = eng.train_iteration()
            rays += st.num_rays; drawn += st.num_rays_drawn; n1 += st.num_samples
            sums += st.sums
        sync()
        dt = time.perf_counter() - t0
        tr.trained += n_steps
        timer = ops.TIMER.summary() if ops.TIMER is not None else {}
        ops.TIMER = None
        d = (col.totals - tot0).cpu().tolist() if col is not None else [0, 0]
        eng.time_exchange = False
        x_steps, x_ms, x_exposed = eng.exchange_ms()
        return {"exchange_ms": x_ms, "exchange_exposed_ms": x_exposed, "exchange_bytes": eng.exchange_bytes,
                "exchange_issue_log": list(eng.exchange_issue_log), "dt": dt, "rays": rays, "drawn": drawn, "n0": int(d[0]), "n_eval": int(d[1]), "n1": n1, "sums": sums,
                "timer": timer, "steps": n_steps, "replaced": ld.replacements - rep0, "trained_before": tr.trained - n_steps,
                "iters": (col.iterations_prefetched - it0[0], col.iterations_classic - it0[1]) if col is not None else (0, 0),
                "spec": (col.march_launches - sp0[0], col.march_launch_rays - sp0[1], col.rays_used - sp0[2]) if col is not None else (0, 0, 0)}

    def point(m):
        return {"steps_trained_before": m["trained_before"], "rays_per_s_this_rank": round(m["rays"] / m["dt"], 1),
                "ms_per_step": round(1e3 * m["dt"] / m["steps"], 3),
                "samples_per_ray_post": round(m["n1"] / max(m["rays"], 1), 2),
                "samples_per_ray_pre": round(m["n0"] / max(m["drawn"], 1), 2),
                "train_psnr_db": round(TrainEngine.psnr_from_sums(m["sums"], max(m["rays"], 1)), 2)}

    def val_pairs():
        """Held-out (camera, frame) pairs: the validation cameras in turn; every other view at a frame the pool is training on
        right now, the rest spread over the sequence."""
        pf = loader.frame_numbers_cuda.cpu()
        vframe = int(torch.mode(pf[pf >= 0]).values)
        return vframe, [(val_cams[i % len(val_cams)],
                         vframe if i % 2 == 0 else scene.frame_numbers[(i * 17) % len(scene.frame_numbers)])
                        for i in range(args.validation_views)]

    def reduce_stat(m):
        """(max-over-ranks time, rays / drawn rays / pre-prune / post-prune / encoded samples summed over the ranks)."""
        stat = torch.tensor([m["dt"], m["rays"], m["drawn"], m["n0"], m["n1"], m["n_eval"]], dtype=torch.float64, device=dev)
        if dp:
            import torch.distributed as dist
            mx = stat.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(stat, op=dist.ReduceOp.SUM)
            stat[0] = mx[0]
        return [float(x) for x in stat.tolist()]

    # ---- the headline: --trials independent trajectories (model seed, ray / background seed), each trained --pretrain steps,
    # warmed up and timed over exactly --steps steps; `value` is the median trial's. rays/s = 640 k / (visible samples per ray) /
    # (step time): the regime a model has reached after 2 000 steps moves the number by more than any kernel does (r04: 16.0 to
    # 20.1 visible samples per ray between runs of one build), so one trajectory is a draw, not a measurement.
    # Inside the timed region only the dominant kernel is bracketed by events (`roofline` needs its launch time from there): an event
    # pair costs the launch stream ~7 us of bubble (a barrier packet each way; round 6 trace: idle 0.105 -> 0.133 ms per step with nine
    # spans instead of five), so the other kernels of `roofline_kernels` are timed in a short window AFTER the timed region.
    timed = None if args.kernel_breakdown else {"prune_march"}
    WINDOW_SPANS = {"prune_march", "encode4d_fwd_save", "encode4d_bwd_tables", "encode4d_bwd_tables_accumulate", "encode4d_bwd_vectors",
                    "encode4d_fwd", "mlp_bwd", "color_mlp_fwd", "density_mlp_fwd"}
    n_trials = max(1, args.trials)
    trials, curve0 = [], None
    for k in range(n_trials):
        torch.manual_seed(123 + rank + 7919 * k)      # ray draws, backgrounds (per rank: ray sharding)
        model_k, eng_k = build_engine(args, dev, rank, world, loader, frames, segment_sizes, 1337 + k)
        tr = SimpleNamespace(model=model_k, eng=eng_k, trained=0, k=k, seeds={"torch": 123 + 7919 * k, "model": 1337 + k})
        # long-lived objects go to the permanent generation so the cyclic collector's periodic full passes do not stall the
        # launch thread for ~10 ms in the middle of a step (measured: 3 such stalls per 60 steps, always at the same launch)
        gc.collect()
        gc.freeze()
        if k == 0 and args.pretrain >= 16:   # the same loop from random initialisation (sigma ~ 100 everywhere)
            train(tr, 3)
            curve0 = point(measure(tr, 8))
        train(tr, max(args.pretrain - tr.trained, 0) + args.warmup)
        tr.m = measure(tr, args.steps, timed)
        tr.stat = reduce_stat(tr.m)
        tr.value = tr.stat[1] / tr.stat[0]
        trials.append(tr)
    order = sorted(range(n_trials), key=lambda i: trials[i].value)
    chosen = trials[order[(n_trials - 1) // 2]]       # the median (the lower middle one for an even count)
    for tr in trials:                                  # the other trajectories are done: free their engines
        if tr is not chosen:
            tr.eng = tr.model = None
    gc.unfreeze()                                      # (their objects sit in the permanent generation: cycles there are never collected)
    gc.collect()
    torch.cuda.empty_cache()
    gc.freeze()
    model, eng, m = chosen.model, chosen.eng, chosen.m
    curve = ([curve0] if curve0 is not None else []) + [point(m)]
    kw = measure(chosen, args.kernel_window, None if args.kernel_breakdown else WINDOW_SPANS) if args.kernel_window > 0 else None
    if args.ab_pieces and rank == 0:
        keep = eng.pipeline_pieces
        res = {}
        for rnd in range(3):
            for n in [int(x) for x in args.ab_pieces.split(",")]:
                eng.pipeline_pieces = n
                measure(chosen, 5)
                mm = measure(chosen, 40)
                res.setdefault(n, []).append(round(1e3 * mm["dt"] / mm["steps"], 3))
        eng.pipeline_pieces = keep
        print("AB pieces (ms/step per round):", res, file=sys.stderr, flush=True)
    if args.ab_overlap_vectors and rank == 0:
        keep = eng.overlap_vector_scatter
        res = {}
        for rnd in range(4):
            for on in (True, False):
                eng.overlap_vector_scatter = on
                measure(chosen, 5)
                mm = measure(chosen, 40)
                res.setdefault("second stream" if on else "same stream", []).append(
                    (round(1e3 * mm["dt"] / mm["steps"], 3), round(1e3 * mm["dt"] * 640_000 / max(mm["n1"], 1), 3)))
        eng.overlap_vector_scatter = keep
        print("AB vector half of the backward (ms/step, ms per 640k samples; per round):", res, file=sys.stderr, flush=True)
    if args.ab_env and rank == 0:
        # measurement aid: alternate an environment switch the library reads at launch time (a measurement build's) on one trajectory
        name = args.ab_env
        res = {}
        for rnd in range(4):
            for val in ("0", "1"):
                torch.cuda.synchronize()
                os.environ[name] = val
                measure(chosen, 5)
                mm = measure(chosen, 40)
                res.setdefault(f"{name}={val}", []).append((round(1e3 * mm["dt"] / mm["steps"], 3), round(1e3 * mm["dt"] * 640_000 / max(mm["n1"], 1), 3),
                                                            round(mm["n1"] / max(mm["rays"], 1), 2)))
        os.environ[name] = "0"
        print(f"AB {name} (ms/step, ms per 640k samples, samples per ray; per round):", res, file=sys.stderr, flush=True)
    if args.ab_main_priority and rank == 0:
        import contextlib
        hi = torch.cuda.Stream(device=dev, priority=-1)
        res = {}
        for rnd in range(4):
            for name in ("default stream", "high-priority stream"):
                torch.cuda.synchronize()
                ctx = torch.cuda.stream(hi) if name.startswith("high") else contextlib.nullcontext()
                with ctx:
                    measure(chosen, 5)
                    mm = measure(chosen, 40)
                    torch.cuda.synchronize()
                res.setdefault(name, []).append((round(1e3 * mm["dt"] / mm["steps"], 3), round(1e3 * mm["dt"] * 640_000 / max(mm["n1"], 1), 3),
                                                 round(mm["n1"] / max(mm["rays"], 1), 2)))
        print("AB main-stream priority (ms/step, ms per 640k samples, samples per ray; per round):", res, file=sys.stderr, flush=True)
        try:
            print("stream priority range:", torch.cuda.Stream.priority_range(), file=sys.stderr, flush=True)
        except Exception as e:
            print("priority_range unavailable:", e, file=sys.stderr, flush=True)
    skipped = eng.found_inf()
    validation = None
    def shared_pairs():
        """The held-out views, the same on every rank (rank 0's pool decides the frame): validation is sharded over the ranks."""
        got = val_pairs()
        if world > 1:
            box = [got]
            torch.distributed.broadcast_object_list(box, src=0)
            got = box[0]
        return got

    if not args.no_validation and val_cams and (rank == 0 or world > 1):
        # (data parallel: every rank renders its share of the views on its replica, the PSNRs are gathered -- SURVEY.md 8(e))
        loader.pause_replacing()
        vframe, pairs = shared_pairs()
        res = validate(model, loader, pairs, rays_batch_size=65536, world_size=world, rank=rank)
        if rank != 0:
            loader.continue_replacing()
    if not args.no_validation and rank == 0 and val_cams:
        validation = {"psnr_db_mean": round(res["psnr_mean"], 3), "psnr_db": [round(p, 3) for p in res["psnr"]],
                      "views": [{"camera": c, "frame": f} for c, f in pairs], "cameras_in_training": False,
                      "steps_trained": chosen.trained, "views_rendered_per_rank": res["images_rendered_here"] if world > 1 else len(pairs)}
        # diagnostic: a TRAINING camera rendered the same way (evaluation mode: zero camera embedding, humanrf.py:196-204)
        # tells a model that leans on its camera embeddings (low here too) from one that does not generalise (high here)
        tcam = loader.camera_numbers[0]
        validation["training_camera_eval_mode_psnr_db"] = round(validate(model, loader, [(tcam, vframe)], 65536)["psnr_mean"], 3)
        if args.emb > 0:
            validation["training_camera_own_embedding_psnr_db"] = round(own_embedding_psnr(model, loader, tcam, vframe), 3)
            # diagnostic, clearly not the reference's evaluation: the same held-out views rendered with the embedding of the NEAREST
            # TRAINING CAMERA in place of the zero vector model.eval() uses (humanrf.py:196-204 is left as it is). (The mean of
            # the training embeddings would say nothing: they start as N(0, 1) draws, their mean is ~0.)
            w = model.camera_embeddings.weight.data
            org = scene.all_camera_origins
            tc = torch.tensor(loader.camera_numbers, device=w.device)
            vc = torch.tensor(sorted({c for c, _ in pairs}), device=w.device)
            near = tc[torch.cdist(org[vc], org[tc]).argmin(dim=1)]
            saved = w[vc].clone()
            w[vc] = w[near]
            try:
                near_emb = [own_embedding_psnr(model, loader, c, f) for c, f in pairs]
            finally:
                w[vc] = saved
            validation["diagnostic_nearest_training_camera_embedding_psnr_db"] = [round(p, 3) for p in near_emb]
            validation["diagnostic_nearest_training_camera_embedding_psnr_db_mean"] = round(sum(near_emb) / len(near_emb), 3)
            validation["note"] = ("camera_embedding_dim > 0: validation renders with a zero embedding (humanrf.py:196-204); how much "
                                  "the colour network leans on the embeddings varies from run to run (DESIGN.md section 4, "
                                  "profiles/r03_psnr_variance_by_step_variant.txt); diagnostic_nearest_training_camera_embedding_* renders "
                                  "the same views with the embedding of the nearest training camera instead (not the reference's "
                                  "evaluation); --emb 0 is the paper's setting")
            validation["camera_embedding_rms"] = round(float(w[torch.tensor(loader.camera_numbers, device=w.device)].pow(2).mean().sqrt()), 4)
        loader.continue_replacing()
    later = [int(x) for x in args.curve.split(",") if x.strip()] if args.pretrain >= 16 else []
    for target in later:
        if target > chosen.trained + 40:
            train(chosen, target - chosen.trained - 20)
            pm = measure(chosen, 20)
            p = point(pm)
            if not args.no_validation and val_cams and (rank == 0 or world > 1):
                loader.pause_replacing()
                p["validation_psnr_db"] = round(validate(model, loader, shared_pairs()[1], 65536, world_size=world, rank=rank)["psnr_mean"], 3)
                p["validation_views"] = args.validation_views
                loader.continue_replacing()
            curve.append(p)
    loader.drain_replacer()

    # ---- secondary legs: driver-visible numbers for the other BASELINE.json configurations, one short trajectory
    # each (--other-pretrain training steps, 5 warm-up, 20 timed, replacer live), reported under `other_configs`; the headline fields
    # above are final before these run. Only on the default single-GPU line (the workload the driver runs).
    other_configs = []
    default_line = (world == 1 and not args.force_collectives and args.image == 752 and args.frames == 50 and args.mlp_precision == "fp16"
                    and args.partitioning == "adaptive" and args.log2_hashmap_size == 19 and args.pretrain >= 1000)
    if default_line and not args.no_other_configs:
        import copy as _copy

        def leg(label, baseline_config, a2, ld, fr, segs, cap, pretrain=None, psnr_views=None):
            torch.manual_seed(123 + 104729)
            model_l, eng_l = build_engine(a2, dev, rank, world, ld, fr, segs, 1337 + 17)
            tl = SimpleNamespace(model=model_l, eng=eng_l, trained=0, loader=ld)
            t_leg = time.perf_counter()
            train(tl, (a2.other_pretrain if pretrain is None else pretrain) + 5)
            ml = measure(tl, 20)
            st = reduce_stat(ml)
            rec = {"config": label, "baseline_config": baseline_config, "dtype": ("bf16 MLP operands" if a2.mlp_precision == "bf16" else "f16 MLP operands") + ", f16 tables, f32 accumulate",
                   "value": round(st[1] / st[0], 1), "unit": "rays/s", "ms_per_step": round(1e3 * st[0] / 20, 3), "steps": 20,
                   "pretrain_steps": tl.trained - 20, "samples_per_ray_post": round(st[4] / max(st[1], 1), 2),
                   "ms_per_640k_samples": round(1e3 * st[0] * 640_000 / max(st[4], 1), 3),
                   "train_psnr_db": round(TrainEngine.psnr_from_sums(ml["sums"], max(ml["rays"], 1)), 2),
                   "replacements_in_timed_region": ml["replaced"], "training_cameras": len(ld.camera_numbers),
                   "replacer_source": ("HBM-resident capture" if type(cap).__name__ == "ResidentCapture" else
                                       f"pinned host capture, {cap.images.numel() / 2 ** 30:.1f} GB" if cap is not None else "rendered on demand"),
                   "one_trajectory": True}
            if psnr_views:       # the PSNR half of the metric for this leg: the same held-out views as the headline's validation
                ld.pause_replacing()
                rv = validate(model_l, ld, psnr_views, rays_batch_size=65536)
                ld.continue_replacing()
                rec["validation_psnr_db"] = round(rv["psnr_mean"], 3)
                rec["validation_psnr_db_per_view"] = [round(p, 2) for p in rv["psnr"]]
            rec["leg_s"] = round(time.perf_counter() - t_leg, 1)
            tl.eng = tl.model = None
            return rec

        try:
            if args.emb > 0 and validation is not None:
                # the paper's setting (example_humanrf.py:19: "set to 0 for the numerical comparisons in the paper"): validation renders
                # every camera with a ZERO embedding, so with embeddings the novel-view PSNR spreads by several dB from run to run
                # (DESIGN.md section 4); without them it is a reading of the model. Same regime as the headline (--pretrain steps).
                a_e0 = _copy.copy(args); a_e0.emb = 0
                other_configs.append(leg("headline workload, camera_embedding_dim 0", "configs[1] at the paper's setting: the PSNR half "
                                         "of the metric without the camera embeddings' run-to-run spread", a_e0, loader, frames,
                                         segment_sizes, capture, pretrain=args.pretrain,
                                         psnr_views=[(v["camera"], v["frame"]) for v in validation["views"]]))
            a_bf = _copy.copy(args); a_bf.mlp_precision = "bf16"
            other_configs.append(leg("headline workload, bf16 MLP", "configs[4] arithmetic (fp16 hash tables + MFMA bf16 MLP) on the "
                                     "configs[1] workload", a_bf, loader, frames, segment_sizes, capture))
            loader.drain_replacer()
            loader.stop_replacer()
            gc.unfreeze(); gc.collect(); torch.cuda.empty_cache()
            a_1x = _copy.copy(args); a_1x.image = 3008; a_1x.host_capture_gb = min(args.host_capture_gb, 24.0)
            t_s = time.perf_counter()
            scene_x, loader_x, segs_x, _, cap_x, frames_x = build_scene(a_1x, dev, rank, world)
            torch.cuda.synchronize()
            loader_x.start_replacer(args.replacements_per_step)
            rec = leg("1x scale, 3008^2 px, 50 frames", "configs[2] (Actor01/Sequence1 1x full-res, 50 frames, 1 GPU)", a_1x, loader_x,
                      frames_x, segs_x, cap_x)
            rec["setup_s"] = round(time.perf_counter() - t_s - rec["leg_s"], 1)
            other_configs.append(rec)
            loader_x.drain_replacer(); loader_x.stop_replacer()
            del scene_x, loader_x, cap_x
            gc.collect(); torch.cuda.empty_cache()
        except Exception as e:      # a secondary leg must never cost the headline its line
            other_configs.append({"error": f"{type(e).__name__}: {e}"})

    dt_max, rays_all, drawn_all, n0_all, n1_all, n_eval_all = chosen.stat

    if rank == 0:
        timer, n_eval, n1 = m["timer"], m["n_eval"], m["n1"]
        # where a kernel's launch time comes from: the timed region (the dominant kernel) or the window behind it (everything else)
        SRC_TIMED = {"timer": timer, "steps": args.steps, "n_eval": n_eval, "n1": n1, "where": f"the {args.steps} timed steps"}
        SRC_WIN = SRC_TIMED if kw is None else {"timer": kw["timer"], "steps": kw["steps"], "n_eval": kw["n_eval"], "n1": kw["n1"],
                                                "where": f"a window of {kw['steps']} steps behind the timed region"}
        # HBM-side traffic of the gather / scatter kernels comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE,
        # TCC_ATOMIC: MI355X_MICROARCH.md) whose summary is committed with the fingerprint of the kernel sources it was taken
        # on (tools/make_traffic_json.py); a summary taken on other sources is not reported.
        traffic_json, traffic_src = None, None
        fp = kernel_source_fingerprint()
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
            cand = json.load(open(os.path.join(ROOT, "profiles", name)))
            if cand.get("kernel_sources_sha256") == fp:
                traffic_json, traffic_src = cand, f"profiles/{name} (rocprofv3 --pmc passes on kernel sources {fp[:12]})"
                break

        # `bound` names the roofline `frac` is priced against (the HBM byte roofline of SURVEY.md 8(d)); `limiter_from_profile` says
        # what the kernel was MEASURED to run into -- evidence taken on the default configuration (adaptive partitioning, log2_T
        # 19, fp16 MLP, binned scatter) with tools/kbench.py, so it is attached to lines of that configuration only (ADVICE r04:
        # a hard-coded label must not pose as a measurement of the run at hand).
        default_regime = (args.partitioning == "adaptive" and args.log2_hashmap_size == 19 and args.mlp_precision == "fp16"
                          and args.frames == 50 and args.image == 752 and eng.scatter_ws is not None)
        LIMITER = {
            "prune_march": {"limiter": "a balanced kernel: its gathers alone take 0.85 and everything else alone 0.66 of its time and the two "
                                       "overlap; not bytes (0.21 x algorithmic reach HBM), not L2 misses (no-miss bound -14 %), not L1 line "
                                       "look-ups (paired fetches: -24 % look-ups in isolation, +15..22 % time in place)",
                            "evidence": "round 6: offsets folded into an L2-resident window 0.307 -> 0.264 ns per encoded sample; clock-rotated "
                                        "level order (shipped) L2 misses -36..40 %, time -2..3 %; round 5 ablations",
                            "profiles": ["profiles/r06_l2_phase_go_nogo.txt", "profiles/r06_pair_fetch_no_go.txt",
                                         "profiles/r06_microbench_pair_loads.txt", "profiles/r05_gather_bound_ablations.txt"]},
            "encode4d_fwd_save": {"limiter": "the hash gather's request path: gathers alone 0.209 of the kernel's 0.222 ms; its no-miss bound is "
                                             "-13 %, most of which the clock-rotated level order (shipped) collects",
                                  "evidence": "round 5 ablation; round 6: 0.229 -> 0.199 ms with no L2 miss, 0.206-0.211 ms with the rotated order",
                                  "profiles": ["profiles/r06_l2_phase_go_nogo.txt", "profiles/r05_gather_bound_ablations.txt"]},
            "encode4d_bwd_tables": {"limiter": "k_scatter_emit: vector-ALU issue + LDS slot counters + scattered 12-byte stores; "
                                               "k_scatter_accumulate: per-workgroup phases at one 128 KB workgroup per CU",
                                    "evidence": "SQ counters (VALU 62-70 % / 30 %); two accumulate workgroups per CU: -19 %, more reads in "
                                                "flight per wavefront: no change",
                                    "profiles": ["profiles/r04_sq_k_scatter_emit_k_scatter_accumulate_rewritten.txt",
                                                 "profiles/r05_scatter_variants.txt"]},
        }

        for tsrc in ({id(SRC_TIMED): SRC_TIMED, id(SRC_WIN): SRC_WIN}).values():
            tt = tsrc["timer"]
            if "encode4d_bwd_tables_accumulate" in tt and "encode4d_bwd_tables" in tt:
                # data parallel: the scatter runs as emit + one accumulate launch per segment group (the groups' collectives in between)
                tt["encode4d_bwd_tables"]["ms_total"] += tt["encode4d_bwd_tables_accumulate"]["ms_total"]

        def line(span, kname, unit_key, bytes_per_unit, tkey=None, src=None):
            src = SRC_WIN if src is None else src
            e = src["timer"].get(span)
            if e is None or e["ms_total"] <= 0:
                return None
            units = src[unit_key]
            achieved = units * bytes_per_unit / (e["ms_total"] * 1e-3) / 1e9
            traffic = None
            if traffic_json is not None and tkey in traffic_json:
                t = traffic_json[tkey]
                traffic = round((t["fetch_bytes_per_encoded_sample"] + t["write_bytes_per_encoded_sample"]) * units /
                                max(e["launches"], 1))
            limiter = LIMITER.get(span) if default_regime else None
            if span == "encode4d_bwd_tables" and eng.scatter_ws is None:
                limiter = {"limiter": "memory-side atomic requests (0.86 of the 21.1 G/s the chip retires)", "evidence": "PMC TCC_ATOMIC",
                           "profiles": ["profiles/r02_microbench_scatter_probe.txt"]}
            return {"bound": "hbm", "limiter_from_profile": limiter, "kernel": kname,
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_over_algorithmic": (round(traffic / max(units * bytes_per_unit / max(e["launches"], 1), 1), 3)
                                                 if traffic is not None else None),
                    "traffic_unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE)",
                    "traffic_source": traffic_src if traffic is not None else
                                      "not reported: no PMC summary under profiles/ was taken on these kernel sources",
                    "algorithmic_bytes_per_unit": bytes_per_unit, "unit_of_work": "encoded sample",
                    "algorithmic_bytes_per_launch": round(units * bytes_per_unit / max(e["launches"], 1)),
                    "launches": e["launches"], "avg_launch_ms": round(e["ms_total"] / max(e["launches"], 1), 4),
                    "ms_per_step": round(e["ms_total"] / src["steps"], 4), "timed_over": src["where"]}

        kernels = [
            line("prune_march", "k_prune_march (hash gather + sigma_net + visibility, prune pass)", "n_eval", ENC_BYTES_PER_SAMPLE,
                 "k_prune_march", src=SRC_TIMED),
            line("encode4d_fwd_save", "k_encode4d_fwd<save> (hash gather + compose, render pass)", "n1", ENC_BYTES_PER_SAMPLE,
                 "k_encode4d_fwd"),
            line("encode4d_bwd_tables", ("k_scatter_emit + k_scatter_accumulate (table-gradient scatter, binned)"
                                         if eng.scatter_ws is not None else "k_encode4d_bwd_tables_lm (table-gradient scatter, atomics)"),
                 "n1", BWD_BYTES_PER_SAMPLE, "table_scatter"),
        ]
        kernels = [k for k in kernels if k is not None]
        # The MLP kernels against the matrix-core peak (north_star: "MFMA utilisation on the MLP against gfx950 peaks"): flops from
        # SURVEY.md 8(d) -- sigma_net 6 144 per sample, colour network 2 (K 64 + 64 64 + 64 16) with K = 32 (no embedding) or 48; the
        # backward recomputes the forward and forms input and weight gradients: 3 x the forward. MfmaUtil (rocprofv3 --pmc) comes from
        # the newest profiles/*_mfma.json taken on these kernel sources.
        color_flops = 2 * (model.color_in_pad * 64 + 64 * 64 + 64 * 16)
        mfma_json = None
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_mfma.json")), reverse=True):
            cand = json.load(open(os.path.join(ROOT, "profiles", name)))
            if cand.get("kernel_sources_sha256") == fp:
                mfma_json, mfma_src = cand, f"profiles/{name}"
                break

        def mfma_line(span, kname, unit_key, flops_per_unit, ukey, note, src=None):
            src = SRC_WIN if src is None else src
            e = src["timer"].get(span)
            units = src[unit_key]
            if e is None or e["ms_total"] <= 0 or units <= 0:
                return None
            achieved = units * flops_per_unit / (e["ms_total"] * 1e-3) / 1e12
            util = (mfma_json or {}).get(ukey)
            return {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_PEAK_TFLOPS, 5), "flops_per_unit": flops_per_unit, "unit_of_work": "sample",
                    "launches": e["launches"], "avg_launch_ms": round(e["ms_total"] / max(e["launches"], 1), 4),
                    "ms_per_step": round(e["ms_total"] / src["steps"], 4), "timed_over": src["where"],
                    "mfma_util_percent_pmc": util, "mfma_util_source": (mfma_src if util is not None else
                                                                         "not reported: no MfmaUtil pass under profiles/ on these kernel sources"),
                    "note": note}
        mlp_kernels = [
            mfma_line("mlp_bwd", "k_mlp_bwd (both networks: forward recompute + input and weight gradients)", "n1",
                      3 * (SIGMA_FLOPS + color_flops), "k_mlp_bwd",
                      "one wavefront per SIMD (176 weight-gradient accumulator registers): bound by its dependent MFMA -> convert -> MFMA "
                      "chains, not by matrix-core issue (profiles/r04_sq_k_mlp_bwd.txt)"),
            mfma_line("color_mlp_fwd", "k_color_fwd (SH16 + identity encoding + colour network)", "n1", color_flops, "k_color_fwd",
                      "bound by its per-sample streams (h 32 B in, rgb 6 B out, ray gathers), not by the matrix cores"),
            mfma_line("density_mlp_fwd", "k_density_fwd (sigma_net + truncated_exp, render pass)", "n1", SIGMA_FLOPS, "k_density_fwd",
                      "64 B in, 36 B out per sample: an HBM stream"),
            mfma_line("prune_march", "k_prune_march: its sigma_net share", "n_eval", SIGMA_FLOPS, "k_prune_march",
                      "the march is bound by its hash gathers; the MFMA share is 6 144 of its flops per encoded sample", src=SRC_TIMED),
        ]
        mlp_kernels = [k for k in mlp_kernels if k is not None]
        for k in kernels:   # memory-side atomic requests of the scatter (PMC TCC_ATOMIC_sum) against the 21.1 G/s the chip retires
            if "table-gradient scatter" in k["kernel"] and traffic_json is not None:
                per = traffic_json.get("table_scatter", {}).get("l2_atomic_requests_per_sample")
                if per is not None:
                    rate = per * SRC_WIN["n1"] / max(SRC_WIN["timer"]["encode4d_bwd_tables"]["ms_total"] * 1e-3, 1e-12) / 1e9
                    k["atomic_requests"] = {"per_sample": per, "achieved_G_per_s": round(rate, 2), "ceiling_G_per_s": 21.1,
                                            "frac": round(rate / 21.1, 3),
                                            "source": f"PMC TCC_ATOMIC_sum, {traffic_src}; ceiling: profiles/r01_microbench_atomic_rates.txt"}
        # `roofline` = the kernel that takes the most time per step (the others stay in roofline_kernels)
        roofline = max(kernels, key=lambda k: k["ms_per_step"]) if kernels else None
        # the whole step against the HBM peak: algorithmic bytes of the three passes over the step time
        step_bytes = (n_eval_all * ENC_BYTES_PER_SAMPLE + n1_all * (ENC_BYTES_PER_SAMPLE + BWD_BYTES_PER_SAMPLE)) / max(args.steps, 1)
        step_algorithmic = {"bytes_per_step": round(step_bytes), "achieved": round(step_bytes * args.steps / dt_max / 1e9, 1),
                            "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                            "frac": round(step_bytes * args.steps / dt_max / 1e9 / (HBM_PEAK_GBS * world), 4),
                            "note": "prune-pass gather + render-pass gather + gradient scatter, all ranks"}
        breakdown = {k: round(v["ms_total"] / SRC_WIN["steps"], 3) for k, v in sorted(SRC_WIN["timer"].items())}
        breakdown["prune_march"] = round(timer["prune_march"]["ms_total"] / args.steps, 3) if "prune_march" in timer else breakdown.get("prune_march")
        used_share = m["spec"][2] / max(m["spec"][1], 1) if m["spec"][1] > m["spec"][2] > 0 else 1.0
        scale = ({752: "4x", 3008: "1x"}).get(args.image, "custom scale")
        out = {
            "metric": "training rays/sec", "value": round(rays_all / dt_max, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "pretrain_steps": args.pretrain,
            "ms_per_step": round(1e3 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": ("f16 tables/MLP operands, f32 accumulate + master weights" if args.mlp_precision == "fp16" else
                                         "f16 tables, bf16 MLP operands, f32 accumulate + master weights"),
            "data": f"synthetic ActorsHQ-shaped scene, random-init weights trained for {chosen.m['trained_before']} steps "
                    "before the timed region",
            "config": {"workload": f"Actor01/Sequence1-shaped {scale}, {args.frames} frames, {args.cameras}-camera rig "
                                   f"({len(loader.camera_numbers)} training cameras), {args.image}^2 px, grid {args.grid}^3, "
                                   f"segments {list(segment_sizes)}, log2_T {args.log2_hashmap_size}, emb {args.emb}",
                       "samples_max_batch_size": eng.samples_max, "rays_initial_batch_size": args.rays_initial,
                       "parallelism": f"ray-sharded dp{world}" + ("" if world == 1 and not args.force_collectives else
                                      (f": per-GPU sample budget fixed, so the global batch is {world}x the reference's; "
                                       if args.scaling == "weak" else
                                       f": the reference's sample budget divided over the ranks ({eng.samples_max} per GPU); ") +
                                      ("tables replicated, gradients reduce-scattered, Adam on the owned 1/N of every segment, "
                                       "fp16 tables all-gathered" if "reduce_scatter_tensor" in eng.collectives_used else
                                       "tables replicated, Adam on the owned 1/N of every segment; this backend has no tensor "
                                       "reduce-scatter / all-gather: all_reduce + list all_gather stood in"
                                       if any("stand-in" in c for c in eng.collectives_used) else
                                       f"tables replicated, one {args.transport} gradient all-reduce per step") +
                                      ", restricted to the segments whose frames are in the pools")},
            # the trials the headline is the median of (whole-job numbers: time = max over ranks, rays summed over ranks)
            "value_is": f"median of {n_trials} independent trajectories" if n_trials > 1 else "one trajectory (--trials 1)",
            "value_min": round(min(t.value for t in trials), 1), "value_max": round(max(t.value for t in trials), 1),
            "trials": [{"seeds": t.seeds, "value": round(t.value, 1), "ms_per_step": round(1e3 * t.stat[0] / args.steps, 3),
                        "samples_per_ray_post": round(t.stat[4] / max(t.stat[1], 1), 2),
                        "samples_rendered_per_s": round(t.stat[4] / t.stat[0], 1),
                        "ms_per_640k_samples": round(1e3 * t.stat[0] * 640_000 * world / max(t.stat[4], 1), 3),
                        "train_psnr_db": round(TrainEngine.psnr_from_sums(t.m["sums"], max(t.m["rays"], 1)), 2),
                        "chosen": t is chosen} for t in trials],
            # regime-independent companions of `value`: a step renders ~640 k samples whatever the rays' length
            "samples_rendered_per_s": round(n1_all / dt_max, 1),
            "ms_per_640k_samples": round(1e3 * dt_max * 640_000 * world / max(n1_all, 1), 3),
            "rays_drawn_per_s": round(drawn_all / dt_max, 1),
            # the march's counters include the rays marched speculatively beyond what the batch-growing loop used
            # (`drawn_rays_marched_over_used` below); the per-second figures are scaled to the used share, as the
            # reference's definition (samples of the batches that enter the merge, trainer.py:145-172) has it
            "samples_pre_prune_per_s": round(n0_all * used_share / dt_max, 1), "samples_post_prune_per_s": round(n1_all / dt_max, 1),
            "samples_encoded_by_prune_per_s": round(n_eval_all * used_share / dt_max, 1),
            "samples_per_ray_pre": round(n0_all * used_share / max(drawn_all, 1), 2), "samples_per_ray_post": round(n1_all / max(rays_all, 1), 2),
            "train_psnr_db": round(TrainEngine.psnr_from_sums(m["sums"], max(m["rays"], 1)), 3),
            "skipped_step_flag": bool(skipped),
            "grad_scaler": {"skipped_steps_total": int(eng.opt_state[1].item()), "scale_now": eng.grad_scale,
                            "internal_scale": eng.internal_grad_scale,
                            "note": "torch.amp.GradScaler semantics on the device (init 65536, backoff 0.5, growth interval "
                                    "100000 as example_humanrf.py) x tcnn's loss_scale 128"},
            "kernel_ms_per_step": breakdown,
            "roofline": roofline,
            "roofline_kernels": kernels + mlp_kernels,
            "step_algorithmic": step_algorithmic,
            "regime_curve": curve,
            "collector_iterations": {"prefetched": m["iters"][0], "classic": m["iters"][1]},
            "prune_march_launches_per_step": round(m["spec"][0] / args.steps, 3),
            "drawn_rays_marched_over_used": round(m["spec"][1] / max(m["spec"][2], 1), 4),
            "replacer": {"thread": True, "replacements_in_timed_region": m["replaced"],
                         "per_step": args.replacements_per_step,
                         "source": ("rendered on demand" if capture is None else
                                    f"pinned host capture ({len(capture.camera_numbers)} cameras x {len(capture.frame_numbers)} frames, "
                                    f"{capture.images.numel() / 2 ** 30:.1f} GB), read by the replacement kernel over the host link"
                                    if type(capture).__name__ == "HostCapture" else "HBM-resident capture")},
            "setup_s": round(setup_s, 1),
            "gradient_boundaries": args.gradient_boundaries,
        }
        if world > 1 or args.force_collectives:
            # what actually ran (TableShardExchange / allreduce_gradients record every torch.distributed call they issue)
            out["gradient_exchange_ms_per_step"] = (round(m["exchange_ms"], 4) if m["exchange_ms"] is not None else None)
            out["gradient_exchange_exposed_ms_per_step"] = (round(m["exchange_exposed_ms"], 4) if m["exchange_exposed_ms"] is not None
                                                            else None)
            out["gradient_exchange_bytes_per_rank_last_step"] = int(m["exchange_bytes"])
            out["gradient_exchange_issue_order_last_step"] = [[ph, list(sg)] for ph, sg in m["exchange_issue_log"]]
            out["gradient_exchange_accumulate_mode"] = eng.exchange_issue_log_mode
            out["gradient_exchange_groups"] = eng.exchange_groups
            out["gradient_exchange_note"] = ("rank 0, mean over the timed steps. ms_per_step (issued): the first table collective is "
                                             "handed to the backend -> the compute stream has waited for all of them and for the small "
                                             "all-reduce of vectors / MLPs / flags; the accumulate launches of the later segment groups "
                                             "and the vector-gradient kernel run inside this window. exposed: the tail of that window "
                                             "in which the compute stream had nothing left to run. bytes: table-gradient payload this "
                                             "rank handed to the reduce-scatter / all-reduce (+ the fp16 all-gather of the previous "
                                             "step's tables when sharded). The all-gather of the fp16 tables is waited for by the next "
                                             "step's march and is not in the window")
            if getattr(args, "exchange_fallback", None):
                out["exchange_fallback"] = {"asked": args.exchange, "ran": "allreduce", "reason": args.exchange_fallback}
            out["collectives"] = {"backend": torch.distributed.get_backend(), "world_size": world,
                                  "calls": sorted(eng.collectives_used),
                                  "forced_on_one_rank": bool(args.force_collectives and world == 1)}
        if other_configs:
            out["other_configs"] = other_configs
        if validation is not None:
            out["validation"] = validation
            out["validation_psnr_db"] = validation["psnr_db_mean"]
        if world == 1 and not args.no_cpu_baseline:
            loader.stop_replacer()
            out["cpu_baseline"] = cpu_baseline(model, loader, args.cpu_rays)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    loader.stop_replacer()
    if world > 1 or args.force_collectives:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
