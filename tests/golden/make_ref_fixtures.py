#!/usr/bin/env python3
"""Freezes outputs of the REFERENCE's own code, executed here from /root/reference through oracle/ref_harness.py, into
tests/golden/ref_*.npz. Run from the repo root in the build container (the only place /root/reference exists):

    python tests/golden/make_ref_fixtures.py [host] [field] [render]

What runs is the reference's Python unmodified -- humanrf/input.py, utils/activation.py, utils/loss.py,
adaptive_temporal_partitioning.py, scene_representation/humanrf.py (constructor, density, forward),
scene_representation/decomposition4d.py (forward + its autograd Function), volume_rendering.py (prune_samples, render),
trainer.py (train_step, _calculate_losses) with torch.optim.Adam / LambdaLR / GradScaler wired as run.py:101-104 and
trainer.py:74 wire them -- over stand-ins for tinycudann / nerfacc / the nvcc-built compose op whose arithmetic is
oracle/hrf_oracle.py (oracle/ref_stubs.py says exactly what is and is not reference code).
The fixtures travel to the GPU box; tests/test_cpu_ref_fixtures.py checks the oracle and the host-side product code
against them, tests/test_gpu_ref_fixtures.py checks the HIP path against them."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import hrf_oracle as O  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from tests import refcases as RC  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(arrays)} arrays")


# ------------------------------------------------------------------------------------------------ host-level functions
def merge_case_batches(ref_cls, seed: int):
    """A list of InputBatch objects of class `ref_cls` with the invariants the loader guarantees (sorted ray_indices,
    ray_masks.sum() == num_rays, data_loader.py:631-660) -> (batches, max_num_samples)."""
    g = RC.rng(seed)
    nb = int(g.integers(1, 5))
    batches = []
    total = 0
    for _ in range(nb):
        r0 = int(g.integers(3, 40))
        mask = g.random(r0) < 0.7
        if not mask.any():
            mask[0] = True
        R = int(mask.sum())
        counts = g.integers(0, 9, R)
        ray_idx = np.repeat(np.arange(R), counts)
        n = int(counts.sum())
        total += n
        frames = g.integers(15, 22, (R, 1)).astype(np.int32)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        batches.append(ref_cls(
            ray_origins=t(g.random((R, 3), dtype=np.float32)), ray_directions=t(g.random((R, 3), dtype=np.float32)),
            minmaxes=t(g.random((R, 2), dtype=np.float32)), rgba=t(g.random((R, 4), dtype=np.float32)),
            ray_masks=t(mask.reshape(-1, 1)), frame_numbers=t(frames),
            unique_frame_numbers=t(np.unique(frames).reshape(-1, 1)), camera_numbers=t(g.integers(0, 160, (R, 1)).astype(np.int32)),
            sample_distances=t(g.random((n, 1), dtype=np.float32)), ray_indices=t(ray_idx.astype(np.int64)), width=48, height=40))
    mode = seed % 3
    max_n = None if mode == 0 else (max(1, total // 2) if mode == 1 else total + 5)
    return batches, max_n


BATCH_FIELDS = ("ray_origins", "ray_directions", "minmaxes", "rgba", "ray_masks", "frame_numbers", "unique_frame_numbers",
                "camera_numbers", "sample_distances", "ray_indices")
FRAME_TABLE_CASES = [
    (tuple(range(15, 65)), (6, 6, 6, 12, 6, 6, 12)),   # sum 54 > 50: last segment clipped (humanrf.py:80-81)
    (tuple(range(15, 27)), (12,)),
    (tuple(range(15, 65)), (25, 25)),
    (tuple(range(10, 100, 3)), (12, 12, 6)),           # non-contiguous frame numbers
    (tuple(range(15, 265)), (100, 100, 50)),
]
ATP_CASES = [(50, 1.25), (50, 1.1), (50, 1.02), (250, 1.25), (250, 1.04), (37, 1.25), (6, 1.25), (5, 1.25)]


def atp_grids(n_frames: int, G: int = 32):
    from humanrf_amd.dataset.synthetic import SyntheticScene
    frames = list(range(15, 15 + n_frames))
    scene = SyntheticScene(frames, num_cameras=1, width=8, height=8, grid_resolution=G, device="cpu")
    return frames, {f: scene.occupancy_grid(f).numpy() for f in frames}


def make_host(ref):
    out = {}
    for seed in range(12):
        batches, max_n = merge_case_batches(ref.InputBatch, 7000 + seed)
        merged = ref.merge_input_batches(batches, max_num_samples=max_n)
        for k in BATCH_FIELDS:
            out[f"merge{seed}_{k}"] = getattr(merged, k).numpy()
    x = torch.tensor([-30.0, -15.0, -3.5, 0.0, 0.25, 7.0, 15.0, 16.0, 40.0], requires_grad=True)
    y = ref.truncated_exp(x)
    w = torch.linspace(0.5, 2.0, x.numel())
    (y * w).sum().backward()
    out["texp_x"], out["texp_y"], out["texp_w"], out["texp_dx"] = x.detach().numpy(), y.detach().numpy(), w.numpy(), x.grad.numpy()
    g = RC.rng(42)
    pred = torch.from_numpy(np.concatenate([g.random(60, dtype=np.float32) * 1.4 - 0.2, [0.0, 1.0, 1e-12, 1 - 1e-7]]).astype(np.float32))
    target = torch.from_numpy((g.random(pred.numel()) < 0.5).astype(np.float32))
    out["bce_pred"], out["bce_target"], out["bce_out"] = pred.numpy(), target.numpy(), ref.bce_loss(pred, target).numpy()
    out["segsize_table"] = np.array([ref.get_segment_size(n) for n in range(1, 131)], np.int32)
    out["final_segsize_table"] = np.array([ref.get_final_segment_size(n) for n in range(1, 101)], np.int32)
    for i, (frames, segs) in enumerate(FRAME_TABLE_CASES):
        m = RH.make_model(ref, frames, segs, log2_T=10, emb=0)   # tiny tables: only the constructor's lookups matter
        out[f"ft{i}_f2s"] = m.frame_numbers_to_segment_numbers.numpy()
        out[f"ft{i}_f2l"] = m.frame_numbers_to_normalized_local_frame_numbers.numpy()
    grids_cache = {}
    for i, (nf, thr) in enumerate(ATP_CASES):
        if nf not in grids_cache:
            grids_cache[nf] = atp_grids(nf)
            # the grids themselves are stored (bit-packed): they come out of libm-dependent synthetic geometry
            out[f"atpgrids{nf}"] = np.packbits(np.stack([grids_cache[nf][1][f] == 255 for f in grids_cache[nf][0]]))
        frames, grids = grids_cache[nf]
        segs = ref.compute_adaptive_segment_sizes(RH.GridDataset(grids), frames, thr)
        out[f"atp{i}_sizes"] = np.array(segs, np.int32)
        out[f"atp{i}_popcounts"] = np.array([(grids[f] == 255).sum() for f in frames], np.int64)
        print("adaptive partition", nf, thr, "->", segs)
    _save("ref_host.npz", **out)


# ------------------------------------------------------------------------------------------------ field cases
def make_field(ref, name: str):
    f0, nf, segs, log2_T, emb, n = RC.FIELD_CASES[name]
    inp = RC.field_inputs(name)
    sd = RC.seeded_reference_state(segs, log2_T, emb, seed=500 + len(name))
    model = RH.make_model(ref, inp["sorted_frames"], segs, log2_T=log2_T, emb=emb)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("frame_numbers" in k for k in missing.missing_keys), missing
    out = {}
    pos, frames, cams, dirs = inp["positions"], inp["frames"], inp["cams"], inp["directions"]
    uniq = torch.unique(frames).view(-1, 1)
    # Decomposition4D.forward of every segment on the samples of that segment (decomposition4d.py:124-135)
    seg_of = model.frame_numbers_to_segment_numbers[frames.view(-1).long()]
    feats = torch.zeros(n, 32, dtype=torch.half)
    with torch.no_grad():
        for s in range(len(segs)):
            sel = seg_of == s
            if sel.any():
                tloc = model.frame_numbers_to_normalized_local_frame_numbers[frames.view(-1).long()[sel]].unsqueeze(-1)
                feats[sel] = model.feature_grids[s](pos[sel] + 0.5, tloc)
    out["d4_features"] = feats.numpy()
    # HumanRF.density / forward (humanrf.py:158-208), training and evaluation form
    qi = ref.QueryInput(is_training=True, positions=pos, directions=dirs, frame_numbers=frames, unique_frame_numbers=uniq,
                        camera_numbers=cams)
    with torch.no_grad():
        dq = model.density(qi)
        out["density"], out["geo"] = dq.density.numpy(), dq.geometry_features.half().numpy()
        qe = ref.QueryInput(is_training=False, positions=pos, directions=dirs, frame_numbers=frames, unique_frame_numbers=uniq,
                            camera_numbers=cams)
        out["radiance_eval"] = model(qe).radiance.half().numpy()
    q = model(qi)
    out["radiance"] = q.radiance.detach().half().numpy()
    loss = (q.density * inp["a"]).sum() + (q.radiance * inp["b"]).sum()
    loss.backward()
    out["loss"] = np.array([float(loss)])
    out["g_sigma"] = model.sigma_net.params.grad.numpy()
    out["g_color"] = model.color_net.params.grad.numpy()
    if emb > 0:
        out["g_emb"] = model.camera_embeddings.weight.grad.numpy()
    norms = np.zeros((len(segs), 4, 16))
    for s in range(len(segs)):
        fg = model.feature_grids[s]
        out[f"g_vec{s}"] = fg.vectors.grad[:, ::16, :].numpy()
        out[f"g_vec{s}_norm"] = np.array([float(fg.vectors.grad.double().norm())])
        levels = O.hashgrid_levels(16, RC.segment_log2(segs[s], log2_T), 32, RC.PLS)
        for e, nm in enumerate(RC.ENC_NAMES):
            gr = getattr(fg, f"{nm}_encoding").params.grad
            for l, lv in enumerate(levels):
                norms[s, e, l] = float(gr[2 * lv.offset:2 * (lv.offset + lv.size)].double().norm())
            nz = torch.nonzero(gr).view(-1).numpy()
            pick = nz[RC.sample_indices(nz.size, 768, seed=31 * s + e)] if nz.size else nz
            out[f"g_tab{s}_{e}_idx"] = pick.astype(np.int64)
            out[f"g_tab{s}_{e}_val"] = gr[pick].numpy()
            out[f"g_tab{s}_{e}_nnz"] = np.array([nz.size], np.int64)
    out["g_tab_level_norms"] = norms
    _save(f"ref_field_{name}.npz", **out)


# ------------------------------------------------------------------------------------------------ prune / render / train_step
RENDER_FRAMES = tuple(range(15, 27))
RENDER_SEGS, RENDER_LOG2T, RENDER_EMB = (6, 6), 16, 2


def render_sampler_inputs():
    """Sampler inputs of a tiny capture (stored in the fixture: the synthetic images go through libm)."""
    from humanrf_amd.dataset.synthetic import SyntheticScene
    scene = SyntheticScene(RENDER_FRAMES, num_cameras=5, width=28, height=24, grid_resolution=40, device="cpu")
    slots = [(0, 15), (1, 18), (3, 22), (4, 26)]
    rgba = torch.stack([scene.render_rgba(c, f) for c, f in slots]).reshape(-1, 4).numpy()
    grids = np.stack([scene.occupancy_grid(f).numpy() for _, f in slots])
    P = scene.width * scene.height
    idx = RC.rng(123).integers(0, len(slots) * P, 220).astype(np.int64)
    cams = np.array([c for c, _ in slots], np.int32)
    frames = np.array([f for _, f in slots], np.int32)
    return dict(rgba=rgba, grids=grids, idx=idx, cams=cams, frames=frames,
                inverse_krs=scene.all_inverse_krs[cams].numpy(), camera_origins=scene.all_camera_origins[cams].numpy(),
                aabb=scene.aabb.numpy(), W=np.int64(scene.width), H=np.int64(scene.height), G=np.int64(scene.grid_resolution))


def make_render(ref):
    inp = render_sampler_inputs()
    s = O.sampler_get_data(inp["rgba"], None, inp["frames"], inp["cams"], list(inp["grids"]), np.ones(4, bool), inp["idx"],
                           inp["inverse_krs"], inp["camera_origins"], inp["aabb"], int(inp["G"]), int(inp["W"]), int(inp["H"]),
                           4e-4, False, True, True)
    org, dirs, rgba, frames, cams, minmax, ray_mask, t, ray = s
    out = {"in_" + k: v for k, v in inp.items()}
    for nm, a in zip(("origins", "dirs", "rgba_s", "frames_s", "cams_s", "minmax", "ray_mask", "t", "ray"), s):
        out["smp_" + nm] = a
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))

    def batch():
        return ref.InputBatch(ray_origins=tt(org), ray_directions=tt(dirs), minmaxes=tt(minmax), rgba=tt(rgba),
                              ray_masks=tt(ray_mask).view(-1, 1), frame_numbers=tt(frames).view(-1, 1),
                              unique_frame_numbers=torch.unique(tt(frames)).view(-1, 1), camera_numbers=tt(cams).view(-1, 1),
                              sample_distances=tt(t).view(-1, 1).clone(), ray_indices=tt(ray).long(), width=int(inp["W"]),
                              height=int(inp["H"]))

    sd = RC.seeded_reference_state(RENDER_SEGS, RENDER_LOG2T, RENDER_EMB, seed=77, table_scale=0.3, vec_scale=0.4)
    model = RH.make_model(ref, RENDER_FRAMES, RENDER_SEGS, log2_T=RENDER_LOG2T, emb=RENDER_EMB)
    model.load_state_dict(sd, strict=False)
    # evaluation form: no jitter, background 0, zero embedding (trainer.py:283-308)
    ib = batch()
    ref.prune_samples(ib, model, False)
    out["eval_t"], out["eval_ray"] = ib.sample_distances.numpy(), ib.ray_indices.numpy()
    with torch.no_grad():
        ro = ref.render(ib, model, 0, False)
    out["eval_color"], out["eval_acc"] = ro.color.numpy(), ro.weights_sum.numpy()
    # training form: the jitter prune_samples draws is reproduced by re-seeding (volume_rendering.py:63-64)
    ib = batch()
    torch.manual_seed(2024)
    out["jitter"] = torch.rand_like(ib.sample_distances).numpy()
    torch.manual_seed(2024)
    ref.prune_samples(ib, model, True)
    out["train_t"], out["train_ray"] = ib.sample_distances.numpy().copy(), ib.ray_indices.numpy().copy()
    # three train_steps on that batch (trainer.py:229-255), backgrounds reproduced the same way
    tr = RH.make_trainer(ref, model)
    names = [n for n, _ in model.named_parameters()]
    picks = {n: RC.sample_indices(p.numel(), 4096, seed=len(n)) for n, p in model.named_parameters()}
    for step in range(3):
        torch.manual_seed(3000 + step)
        out[f"bg{step}"] = torch.rand_like(ib.rgba[..., 0:3]).numpy()
        torch.manual_seed(3000 + step)
        tr.optimizer.zero_grad(set_to_none=True)                     # trainer.py:174
        loss, info = tr.train_step(ib)
        assert tr.scaler.get_scale() == 65536.0, "the reference's GradScaler backed off: a gradient overflowed fp16"
        out[f"loss{step}"] = np.array([float(loss), info["photometric"], info["psnr"], info["mask_loss"]])
        for n, p in model.named_parameters():
            st = tr.optimizer.state[p]
            out[f"s{step}|{n}|p"] = p.detach().view(-1)[picks[n]].numpy().copy()
            out[f"s{step}|{n}|m"] = st["exp_avg"].view(-1)[picks[n]].numpy().copy()
            out[f"s{step}|{n}|v"] = st["exp_avg_sq"].view(-1)[picks[n]].numpy().copy()
            if step == 0:
                out[f"nnz|{n}"] = np.array([int((st["exp_avg"] != 0).sum())], np.int64)
        out[f"lr{step}"] = np.array([tr.optimizer.param_groups[0]["lr"]])
    out["param_names"] = np.array(names)
    # full-image assembly and its PSNR (trainer.py:517-526, 372-389 via _calculate_losses :218-223) on the evaluation output
    full = ref.InputBatch(ray_masks=tt(ray_mask).view(-1, 1), rgba=tt(rgba), width=int(inp["idx"].shape[0]), height=1)
    ro = ref.RenderOutput(color=tt(out["eval_color"]), weights_sum=tt(out["eval_acc"]))
    out["eval_image"] = tr.combine_rays_to_image(full, ro, 0).numpy()
    gt_rgb = tt(rgba)[..., 0:3] * tt(rgba)[..., 3:4]
    _, info = tr._calculate_losses(ro, gt_rgb, tt(rgba)[..., 3:4])
    out["eval_psnr"] = np.array([info["psnr"]])
    _save("ref_render.npz", **out)
    make_skip_steps(ref, inp)


SKIP_SEQUENCE = ("A", "B", "A", "C")   # A: rays of segment 0 only, B: segment 1 only, C: both


def skip_batches(inp):
    """Sampler outputs for the three ray subsets of the untouched-segment scenario (slots 0,1 -> frames 15,18 -> segment 0;
    slots 2,3 -> frames 22,26 -> segment 1 with RENDER_SEGS = (6, 6))."""
    P = int(inp["W"]) * int(inp["H"])
    sel = {"A": inp["idx"][inp["idx"] // P < 2], "B": inp["idx"][inp["idx"] // P >= 2], "C": inp["idx"]}
    res = {}
    for key, idx in sel.items():
        res[key] = O.sampler_get_data(inp["rgba"], None, inp["frames"], inp["cams"], list(inp["grids"]), np.ones(4, bool), idx,
                                      inp["inverse_krs"], inp["camera_origins"], inp["aabb"], int(inp["G"]), int(inp["W"]),
                                      int(inp["H"]), 4e-4, False, True, True)
    return res


def make_skip_steps(ref, inp):
    """Trainer.train_step over batches that leave a temporal segment untouched: the reference only runs the segments of
    the batch's frames (humanrf.py:159-179), the others keep grad None (zero_grad(set_to_none=True), trainer.py:174) and
    torch.optim.Adam skips them -- moments, values and per-parameter step counts frozen."""
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    sd = RC.seeded_reference_state(RENDER_SEGS, RENDER_LOG2T, RENDER_EMB, seed=78, table_scale=0.3, vec_scale=0.4)
    model = RH.make_model(ref, RENDER_FRAMES, RENDER_SEGS, log2_T=RENDER_LOG2T, emb=RENDER_EMB)
    model.load_state_dict(sd, strict=False)
    tr = RH.make_trainer(ref, model)
    out = {}
    batches = {}
    for key, s in skip_batches(inp).items():
        org, dirs, rgba, frames, cams, minmax, ray_mask, t, ray = s
        ib = ref.InputBatch(ray_origins=tt(org), ray_directions=tt(dirs), minmaxes=tt(minmax), rgba=tt(rgba),
                            ray_masks=tt(ray_mask).view(-1, 1), frame_numbers=tt(frames).view(-1, 1),
                            unique_frame_numbers=torch.unique(tt(frames)).view(-1, 1), camera_numbers=tt(cams).view(-1, 1),
                            sample_distances=tt(t).view(-1, 1).clone(), ray_indices=tt(ray).long(), width=int(inp["W"]),
                            height=int(inp["H"]))
        ref.prune_samples(ib, model, False)      # pruned with the INITIAL model, once; the steps reuse the sample sets
        out[f"{key}_t"], out[f"{key}_ray"] = ib.sample_distances.numpy().copy(), ib.ray_indices.numpy().copy()
        batches[key] = ib
    picks = {n: RC.sample_indices(p.numel(), 2048, seed=len(n) + 1) for n, p in model.named_parameters()}
    for step, key in enumerate(SKIP_SEQUENCE):
        ib = batches[key]
        torch.manual_seed(4000 + step)
        out[f"bg{step}"] = torch.rand_like(ib.rgba[..., 0:3]).numpy()
        torch.manual_seed(4000 + step)
        tr.optimizer.zero_grad(set_to_none=True)
        loss, info = tr.train_step(ib)
        assert tr.scaler.get_scale() == 65536.0
        out[f"loss{step}"] = np.array([float(loss), info["photometric"]])
        for n, p in model.named_parameters():
            st = tr.optimizer.state.get(p, {})
            out[f"s{step}|{n}|p"] = p.detach().view(-1)[picks[n]].numpy().copy()
            out[f"s{step}|{n}|t"] = np.array([int(st["step"]) if "step" in st else 0], np.int64)
            if "exp_avg" in st:
                out[f"s{step}|{n}|m"] = st["exp_avg"].view(-1)[picks[n]].numpy().copy()
                out[f"s{step}|{n}|v"] = st["exp_avg_sq"].view(-1)[picks[n]].numpy().copy()
    out["param_names"] = np.array([n for n, _ in model.named_parameters()])
    _save("ref_steps_skip.npz", **out)


# ------------------------------------------------------------------------------------------------ one step with weak gradients
WEAK_DRAWS, WEAK_SEED = 6000, 321


def weak_sampler_outputs():
    """The tiny capture of the render case, but 6 000 drawn rays (with repetition: 2 688 pixels): the loss is a mean over the
    rays, so every per-sample gradient is ~25 times smaller than in ref_render.npz and a part of the table entries only
    receives contributions below the half floor at the GradScaler's scale."""
    inp = render_sampler_inputs()
    P = int(inp["W"]) * int(inp["H"])
    idx = RC.rng(WEAK_SEED).integers(0, 4 * P, WEAK_DRAWS).astype(np.int64)
    s = O.sampler_get_data(inp["rgba"], None, inp["frames"], inp["cams"], list(inp["grids"]), np.ones(4, bool), idx,
                           inp["inverse_krs"], inp["camera_origins"], inp["aabb"], int(inp["G"]), int(inp["W"]), int(inp["H"]),
                           4e-4, False, True, True)
    return inp, s


def make_weak_step(ref):
    """ONE Trainer.train_step (trainer.py:229-255; GradScaler 65536, torch.optim.Adam) on a batch whose gradients reach below
    the floor of the reference's half gradient tensors (decomposition4d.py:8-39): which sampled table entries the step moves,
    and where to. Pins the half-gradient-boundary rule (oracle.half_gradient, include/hrf.h grad_boundary) on the CPU."""
    inp, s = weak_sampler_outputs()
    org, dirs, rgba, frames, cams, minmax, ray_mask, t, ray = s
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    ib = ref.InputBatch(ray_origins=tt(org), ray_directions=tt(dirs), minmaxes=tt(minmax), rgba=tt(rgba),
                        ray_masks=tt(ray_mask).view(-1, 1), frame_numbers=tt(frames).view(-1, 1),
                        unique_frame_numbers=torch.unique(tt(frames)).view(-1, 1), camera_numbers=tt(cams).view(-1, 1),
                        sample_distances=tt(t).view(-1, 1).clone(), ray_indices=tt(ray).long(), width=int(inp["W"]),
                        height=int(inp["H"]))
    sd = RC.seeded_reference_state(RENDER_SEGS, RENDER_LOG2T, RENDER_EMB, seed=79, table_scale=0.1, vec_scale=0.4)
    model = RH.make_model(ref, RENDER_FRAMES, RENDER_SEGS, log2_T=RENDER_LOG2T, emb=RENDER_EMB)
    model.load_state_dict(sd, strict=False)
    ref.prune_samples(ib, model, False)                              # evaluation form: no jitter to reproduce
    out = {"t": ib.sample_distances.numpy().copy(), "ray": ib.ray_indices.numpy().copy(),
           "num_rays": np.array([ib.ray_origins.shape[0]], np.int64)}
    tr = RH.make_trainer(ref, model)
    torch.manual_seed(5000)
    out["bg"] = torch.rand_like(ib.rgba[..., 0:3]).numpy()
    torch.manual_seed(5000)
    tr.optimizer.zero_grad(set_to_none=True)
    from oracle import ref_stubs
    assert ref_stubs.HALF_GRAD_OUTPUTS        # the compose op's four gradients are at::Half in the reference (see ref_stubs)
    loss, info = tr.train_step(ib)
    assert tr.scaler.get_scale() == 65536.0
    out["loss"] = np.array([float(loss), info["photometric"]])
    names = [n for n, _ in model.named_parameters() if "encoding.params" in n]
    for n, p in model.named_parameters():
        if n in names:
            pick = RC.sample_indices(p.numel(), 8192, seed=len(n) + 2)
            out[f"{n}|p"] = p.detach().view(-1)[pick].numpy().copy()
            g = tr.optimizer.state[p]["exp_avg"].view(-1)
            out[f"{n}|nnz"] = np.array([int((g != 0).sum())], np.int64)
    out["param_names"] = np.array(names)
    _save("ref_step_weak.npz", **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["host", "field", "render", "weak"]
    ref = RH.load()
    if "host" in what:
        make_host(ref)
    if "field" in what:
        for name in RC.FIELD_CASES:
            make_field(ref, name)
    if "render" in what:
        make_render(ref)
    if "weak" in what:
        make_weak_step(ref)
