"""The REFERENCE'S OWN SOURCE over libhrf_hip.so (north_star: "run.py and trainer.py use it as a drop-in").

oracle/ref_harness.load("hip") imports the reference's modules (from /root/reference, or from the byte copies
oracle/snapshot_reference.py leaves under oracle/_ref/reference for the GPU box) with humanrf_amd's drop-in modules
registered under the names the reference imports:
    tinycudann, nerfacc, humanrf.scene_representation.tensor_composition_native,
    actorshq.dataset.ray_sampler_native, actorshq.dataset.occupancy_grid_native
so every kernel these loop bodies reach is the HIP library's:
    HumanRF.forward / density                      humanrf/scene_representation/humanrf.py:158-208
    prune_samples, render                          humanrf/volume_rendering.py:42-150
    Trainer.train_step (+ torch Adam, GradScaler)  humanrf/trainer.py:229-255
    the loop body of Trainer.train as a whole      humanrf/trainer.py:138-177 (its statements, compiled from its source)
    DataLoader.__next__, training branch           actorshq/dataset/data_loader.py:539-575,631-660
Each is compared with humanrf_amd's own surface for the same call (the fused kernels) on the same inputs."""
import threading

import numpy as np
import os

import pytest
import torch

from oracle import ref_harness as RH
from tests import refcases as RC
from tests.test_gpu_ref_fixtures import _load, _rel_cos, _set_difference
from tests.util import make_model, small_scene

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not RH.available(), reason="reference sources not on this machine (run __graft_entry__.build() "
                                                            "where /root/reference exists: it snapshots them to oracle/_ref)")]
DEV = "cuda"
FRAMES = tuple(range(15, 27))


@pytest.fixture(scope="module")
def ref():
    ns = RH.load("hip")
    import tinycudann
    import nerfacc
    assert tinycudann.__name__ == "humanrf_amd.compat.tinycudann" and nerfacc.__name__ == "humanrf_amd.compat.nerfacc"
    return ns


def _pair(ref, segs=(6, 6), log2_T=19, emb=2, table_scale=0.3):
    """humanrf_amd's HumanRF and the reference's HumanRF (over the drop-in modules) holding the same parameters."""
    m = make_model(DEV, segs, FRAMES, log2_T=log2_T, emb=emb, table_scale=table_scale)
    rm = RH.make_model(ref, FRAMES, segs, log2_T=log2_T, emb=emb).to(DEV)
    missing = rm.load_state_dict({k: v.to(DEV) for k, v in m.reference_state_dict().items()}, strict=False)
    assert not missing.unexpected_keys, missing
    return m, rm


def _ref_batch(ref, ib):
    """The reference's InputBatch dataclass holding copies of a humanrf_amd batch."""
    c = lambda t: None if t is None else t.clone()
    return ref.InputBatch(ray_origins=c(ib.ray_origins), ray_directions=c(ib.ray_directions), minmaxes=c(ib.minmaxes),
                          rgba=c(ib.rgba), ray_masks=c(ib.ray_masks), frame_numbers=c(ib.frame_numbers),
                          camera_numbers=c(ib.camera_numbers), unique_frame_numbers=c(ib.unique_frame_numbers),
                          sample_distances=c(ib.sample_distances), ray_indices=c(ib.ray_indices), width=ib.width, height=ib.height)


def _loader(batch=3000, seed=4):
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    scene = small_scene(DEV, G=64, W=96, H=80, frames=FRAMES, num_cameras=8)
    ld = SyntheticDataLoader(scene, batch_size=batch, max_buffer_size=16, max_num_frames_per_batch=4, seed=seed)
    iter(ld)
    return scene, ld


def test_reference_humanrf_forward_over_the_dropin_modules(ref):
    """(was test_gpu_compat_tcnn.py::test_reference_source_runs_on_the_compat_modules, skipped on every GPU run so far)"""
    name = "seg12_T15"
    fx, inp = _load(f"ref_field_{name}.npz"), RC.field_inputs(name)
    m = RH.make_model(ref, inp["sorted_frames"], (12,), log2_T=15, emb=2).to(DEV)
    m.load_state_dict({k: v.to(DEV) for k, v in RC.seeded_reference_state((12,), 15, 2, seed=500 + len(name)).items()}, strict=False)
    with torch.autocast("cuda"):
        q = m(ref.QueryInput(is_training=True, positions=inp["positions"].to(DEV), directions=inp["directions"].to(DEV),
                             frame_numbers=inp["frames"].to(DEV), unique_frame_numbers=torch.unique(inp["frames"]).view(-1, 1).to(DEV),
                             camera_numbers=inp["cams"].to(DEV)))
    assert np.allclose(q.density.detach().cpu().numpy(), fx["density"], rtol=2e-2, atol=1e-3)
    assert np.abs(q.radiance.detach().float().cpu().numpy() - fx["radiance"].astype(np.float32)).max() <= 4e-3


@pytest.mark.parametrize("n_levels,geo,emb,hidden", [(8, 7, 2, 3), (12, 15, 0, 1), (5, 4, 3, 2)])
def test_reference_humanrf_with_other_knobs_over_the_dropin_modules(ref, n_levels, geo, emb, hidden):
    """the drop-in tcnn modules take the knob shapes too (sigma_net with 2 n_levels < 32 inputs, the colour network with fewer
    geometry features and 1..3 hidden layers), so THE REFERENCE'S HumanRF class runs over them with n_levels / geometry_feature_dim /
    n_hidden_layers_color other than the defaults -- against humanrf_amd's HumanRF holding the same parameters: outputs, and every parameter
    gradient of a loss (the state dict round trip is the hand-over)."""
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.scene_representation.query_io import QueryInput
    sizes = (6, 6)
    kw = dict(density_scale=100, sorted_frame_numbers=tuple(FRAMES), n_features_per_level=2, log2_hashmap_size=15, n_levels=n_levels,
              coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=geo, n_neurons=64, n_hidden_layers_density=1,
              n_hidden_layers_color=hidden, sh_degree=4, segment_sizes=sizes, camera_embedding_dim=emb)
    m = HumanRF(device=DEV, seed=9, **kw)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        m.table_params.copy_(((torch.rand(m.table_params.numel(), generator=g) * 2 - 1) * 0.5).to(DEV))
    rm = ref.HumanRF(**kw).to(DEV)
    sd = m.reference_state_dict()
    for k, v in rm.state_dict().items():          # the reference's module tree over the drop-ins has the reference's shapes
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), (k, tuple(v.shape))
    missing = rm.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys, missing
    g = torch.Generator().manual_seed(11)
    n = 1200
    pos = (torch.rand(n, 3, generator=g) - 0.5).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(DEV)
    fr = torch.tensor(FRAMES, dtype=torch.int32)[torch.randint(0, len(FRAMES), (n,), generator=g)].reshape(-1, 1).to(DEV)
    cams = torch.randint(0, 160, (n, 1), generator=g, dtype=torch.int32).to(DEV)
    a, b = torch.rand(n, generator=g).to(DEV), torch.rand(n, 3, generator=g).to(DEV)
    uniq = torch.unique(fr).reshape(-1, 1)
    with torch.autocast("cuda"):
        q_ref = rm(ref.QueryInput(is_training=True, positions=pos, directions=dirs, frame_numbers=fr, unique_frame_numbers=uniq,
                                  camera_numbers=cams))
    q = m(QueryInput(is_training=True, positions=pos, directions=dirs, frame_numbers=fr, unique_frame_numbers=uniq, camera_numbers=cams))
    d_ref, d = q_ref.density.detach().float(), q.density.detach().float()
    assert float(((d - d_ref).abs() / d_ref.abs().clamp_min(1e-3)).max()) <= 2e-2
    assert float((q.radiance.detach().float() - q_ref.radiance.detach().float()).abs().max()) <= 4e-3
    assert tuple(q_ref.geometry_features.shape) == tuple(q.geometry_features.shape) == (n, geo)
    ((q_ref.density.float() * a).sum() * 1e-3 + (q_ref.radiance.float() * b).sum()).backward()
    ((q.density * a).sum() * 1e-3 + (q.radiance * b).sum()).backward()

    def close(mine, theirs, name, rel=5e-2):
        mine, theirs = mine.detach().double().reshape(-1), theirs.detach().double().reshape(-1)
        assert mine.shape == theirs.shape, (name, mine.shape, theirs.shape)
        assert float((mine - theirs).norm() / theirs.norm().clamp_min(1e-30)) <= rel, name
    pad = m.sigma_in_pad
    g_s = m.sigma_params.grad
    close(torch.cat([g_s[:2048].reshape(64, 32)[:, :pad].reshape(-1), g_s[2048:]]), rm.sigma_net.params.grad, "sigma_net")
    close(m.color_params.grad, rm.color_net.params.grad, "color_net")
    if emb:
        close(m.camera_embeddings.weight.grad, rm.camera_embeddings.weight.grad, "camera_embeddings")
    off, F = 0, 2 * n_levels
    for s, entries in enumerate(m.entries_per_segment):
        close(m.vectors.grad[s][..., :F], rm.feature_grids[s].vectors.grad, f"vectors {s}")
        for nm in ("xyz", "xyt", "yzt", "xzt"):
            close(m.table_params.grad[off * 2:(off + entries) * 2], getattr(rm.feature_grids[s], f"{nm}_encoding").params.grad, f"tables {s} {nm}")
            off += entries


def test_reference_prune_and_render_equal_the_fused_path(ref):
    """volume_rendering.py:42-150 of the reference (per-segment Decomposition4D modules, boolean-mask plumbing, nerfacc calls)
    and humanrf_amd.volume_rendering (fused march / encode / composite) on one batch: survivors, colours, opacities and the
    parameter gradients of a rendering loss."""
    from humanrf_amd.volume_rendering import prune_samples, render
    m, rm = _pair(ref)
    _, ld = _loader()
    torch.manual_seed(1)
    ib = next(ld)
    rb = _ref_batch(ref, ib)
    assert ib.num_samples > 20_000
    for training in (False, True):
        a, b = _ref_batch(ref, ib), _ref_batch(ref, ib)
        a_own = type(ib)(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(ib).items() if not k.startswith("_")})
        torch.manual_seed(11)
        with torch.autocast("cuda"):
            ref.prune_samples(b, rm, training)
        torch.manual_seed(11)
        prune_samples(a_own, m, training)
        assert 0 < b.sample_distances.numel() < ib.num_samples
        d = _set_difference(a_own.sample_distances.cpu().numpy(), a_own.ray_indices.cpu().numpy(),
                            b.sample_distances.cpu().numpy(), b.ray_indices.cpu().numpy())
        assert d <= 0.005, (training, d)
        del a
    # render + backward on the reference's survivors (training form: camera embeddings on)
    own = type(ib)(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(ib).items() if not k.startswith("_")})
    own.sample_distances, own.ray_indices = b.sample_distances.clone(), b.ray_indices.clone()
    g = torch.Generator(device=DEV).manual_seed(3)
    bg = torch.rand(ib.num_rays, 3, device=DEV, generator=g)
    coef = torch.rand(ib.num_rays, 3, device=DEV, generator=g) + 0.5
    with torch.autocast("cuda"):
        ro_ref = ref.render(b, rm, bg, True)
    ro = render(own, m, bg, True)
    assert float((ro.color - ro_ref.color.float()).abs().max()) <= 2e-3
    assert float((ro.weights_sum - ro_ref.weights_sum.float()).abs().max()) <= 2e-3
    S = 1024.0   # a loss scale, as under GradScaler: tcnn-shaped modules pass fp16 gradients between them
    ((ro_ref.color.float() * coef).sum() * S).backward()
    ((ro.color * coef).sum() * S).backward()
    sd_g = {k: p.grad for k, p in rm.named_parameters() if p.grad is not None}
    from tests.util import record_parity
    T = "test_reference_prune_and_render_equal_the_fused_path"
    names = ("xyz", "xyt", "yzt", "xzt")
    off = 0
    for s, entries in enumerate(m.entries_per_segment):
        rel, cos = _rel_cos(m.vectors.grad[s].cpu().numpy(), sd_g[f"feature_grids.{s}.vectors"].cpu().numpy())
        record_parity(T, f"vectors segment {s}", rel, cos, 1e-2)
        assert cos >= 0.999 and rel <= 1e-2, ("vectors", s, rel, cos)      # measured <= 1.7e-4 (profiles/r06_gradient_parity_measured.txt)
        for nm in names:
            own_g = m.table_params.grad[off * 2:(off + entries) * 2]
            rel, cos = _rel_cos(own_g.cpu().numpy(), sd_g[f"feature_grids.{s}.{nm}_encoding.params"].cpu().numpy())
            record_parity(T, f"tables segment {s} {nm}", rel, cos, 1e-2)
            assert cos >= 0.999 and rel <= 1e-2, (s, nm, rel, cos)          # measured <= 3.3e-4
            off += entries
    for own_p, key in ((m.sigma_params, "sigma_net.params"), (m.color_params, "color_net.params"),
                       (m.camera_embeddings.weight, "camera_embeddings.weight")):
        rel, cos = _rel_cos(own_p.grad.cpu().numpy(), sd_g[key].cpu().numpy())
        record_parity(T, key, rel, cos, 1e-2)
        assert cos >= 0.999 and rel <= 1e-2, (key, rel, cos)                # measured <= 1.8e-5


@pytest.mark.parametrize("boundaries", ["fp32", "fp16"])
def test_reference_trainer_train_step_equals_the_engine(ref, boundaries):
    """gradient_boundaries="fp16" rounds the gradient through half where the reference's modules hand each other half tensors
    (include/hrf.h grad_boundary): the entries whose whole gradient sits below that floor then stay put as in the reference --
    at most 0.1 % of the moved entries move on one side only (measured 0.003 %); "fp32" (contributions of any size reach Adam)
    moves up to 20 % more (measured 9.8 %).

    One Trainer.train_step of the reference (trainer.py:229-255: random background, render, Huber + 1e-3 BCE,
    torch.cuda GradScaler, torch.optim.Adam, LambdaLR) over the drop-in modules, and one TrainEngine.train_step (explicit
    kernel chain, device GradScaler, fused Adam) from the same parameters on the same batch and background."""
    from humanrf_amd.trainer import TrainEngine
    from humanrf_amd.volume_rendering import prune_samples
    m, rm = _pair(ref, table_scale=0.1)
    _, ld = _loader(batch=2500, seed=7)
    torch.manual_seed(2)
    ib = next(ld)
    prune_samples(ib, m, False)
    rb = _ref_batch(ref, ib)
    p0 = {k: v.detach().clone() for k, v in m.reference_state_dict().items()}
    tr = RH.make_trainer(ref, rm, growth_interval=100_000, device="cuda")
    torch.manual_seed(5)
    with torch.autocast("cuda"):
        loss, info = tr.train_step(rb)
    assert tr.scaler.get_scale() == 65536.0 and torch.isfinite(loss)
    eng = TrainEngine(m, loader=None, samples_max_batch_size=60_000, rays_initial_batch_size=2500,
                      gradient_boundaries=boundaries)
    torch.manual_seed(5)
    eng.loss_sums.zero_()
    eng.train_step(ib)
    assert eng.found_inf() == 0 and eng.optimizer_steps() == [1, 1, 1]
    sums = eng.loss_sums.cpu()
    R = ib.num_rays
    own_loss = float(sums[0]) / (3 * R) + 1e-3 * float(sums[1]) / R
    assert abs(own_loss - float(loss)) <= 1e-2 * abs(float(loss)) + 1e-6, (own_loss, float(loss))
    assert abs(info["psnr"] - TrainEngine.psnr_from_sums(eng.loss_sums, R)) <= 0.05
    # parameters after the step. Adam's first step moves an entry by lr * g / (|g| + eps): compare entry by entry where the
    # reference's gradient stands clear of its fp16 noise, and require untouched entries to be untouched on both sides.
    p1 = m.reference_state_dict()
    ref_p1 = {k: v.detach() for k, v in rm.state_dict().items()}
    ref_g = {k: p.grad for k, p in rm.named_parameters()}
    checked = 0
    for k, before in p0.items():
        if not before.dtype.is_floating_point:
            continue
        du = (p1[k] - before).reshape(-1)
        dr = (ref_p1[k].float() - before).reshape(-1)
        g = ref_g[k].reshape(-1).float() if ref_g.get(k) is not None else torch.zeros_like(dr)
        # Entries only ONE side moved. The reference's modules hand each other fp16 gradient tensors (tcnn's outputs and
        # the compose op are half, at the GradScaler's scale): a contribution below 2^-24 / 65536 = 9e-13 is (all but) zero
        # there, and Adam leaves an entry whose gradient is that small where it is (eps = 1e-15). humanrf_amd's fused backward stays in
        # fp32 from the loss to the tables, such an entry keeps its tiny gradient and Adam's first step moves it by lr like
        # any other (a stated deviation, DESIGN.md section 2: ~10 % of the touched entries of a step at this scale).
        # The other direction must not happen, and never on an entry with a clear gradient.
        only_ref = (dr != 0) & (du == 0)
        only_own = (du != 0) & (dr == 0)
        assert int(only_ref.sum()) <= 1e-3 * max(int((dr != 0).sum()), 1000), (k, "moved in the reference only", int(only_ref.sum()))
        own_limit = 0.2 if boundaries == "fp32" else 0.001        # measured: 9.8 % / 0.003 % of the moved entries
        diag = os.environ.get("HRF_TEST_DIAG")
        if diag:
            with open(diag, "a") as f:
                f.write(f"dropin train_step [{boundaries}] {k}: moved ref {int((dr != 0).sum())} own {int((du != 0).sum())} "
                        f"only_ref {int(only_ref.sum())} only_own {int(only_own.sum())}\n")
        assert int(only_own.sum()) <= own_limit * max(int((dr != 0).sum()), 50), (k, "moved here only", int(only_own.sum()))
        # the reference's gradient there: zero, or so far below Adam's eps = 1e-15 that its step rounds away
        assert float(g[only_own].abs().max() if bool(only_own.any()) else 0.0) <= 1e-16
        clear = g.abs() > 0.05 * g.abs().max()
        if int(clear.sum()) == 0:
            continue
        assert int(((only_ref | only_own) & clear).sum()) == 0, (k, "an entry with a clear gradient moved on one side only")
        err = (du[clear] - dr[clear]).abs()
        assert float(err.max()) <= 0.1 * 1e-2, (k, float(err.max()))       # lr = 1e-2: every clear entry within 10 % of a step
        assert float(err.mean()) <= 0.01 * 1e-2, (k, float(err.mean()))
        checked += int(clear.sum())
    assert checked > 10_000


def test_reference_dataloader_next_over_the_hip_sampler(ref):
    """DataLoader.__next__'s training branch (data_loader.py:539-575,631-660) -- the reference's source, object built
    without its file-reading __init__ -- calling ray_sampler_native.get_samples_occupancy_minmax = the HIP sampler, on the
    pool tables of a SyntheticDataLoader; equals SyntheticDataLoader.__next__ under the same torch seed."""
    _, ld = _loader(batch=4096)
    DL = ref.DataLoader
    import actorshq.dataset.ray_sampler_native as sampler_mod
    assert sampler_mod.__name__ == "humanrf_amd.dataset.ray_sampler_native"
    rl = DL.__new__(DL)
    rl.mode = DL.Mode.TRAINING
    rl.device = DEV
    rl.batch_size = 4096
    rl.buffer_size, rl.num_pixels_per_camera, rl.resolution = ld.buffer_size, ld.num_pixels_per_camera, ld.resolution
    rl.data_lock = threading.Lock()
    rl.ray_sampler_func = sampler_mod.get_samples_occupancy_minmax          # data_loader.py:173-175
    rl.pixel_colors_cpu, rl.light_mask_cpu = ld.pixel_colors, ld.light_mask  # (the pool may live in HBM: INTEGRATION.md)
    for name in ("frame_numbers_cuda", "camera_numbers_cuda", "grid_texture_objects_cuda", "landscape_mode_cuda",
                 "inverse_krs_cuda", "camera_origins_cuda", "aabb", "occupancy_grid_resolution"):
        setattr(rl, name, getattr(ld, name))
    rl.filter_light_bloom = False
    rl.iternum = 0
    torch.manual_seed(31)
    rb = next(rl)
    torch.manual_seed(31)
    ld.batch_size = 4096
    ib = next(ld)
    assert rb.num_rays == ib.num_rays > 100 and rb.num_samples == ib.num_samples > 1000
    for f in ("ray_origins", "ray_directions", "minmaxes", "rgba", "ray_masks", "frame_numbers", "camera_numbers",
              "sample_distances", "ray_indices"):
        assert torch.equal(getattr(rb, f), getattr(ib, f)), f
    assert sorted(rb.unique_frame_numbers.view(-1).tolist()) == sorted(ib.unique_frame_numbers.view(-1).tolist())
    assert rl.iternum == 4096


def _reference_loop_body(ref):
    """The statements of the reference's Trainer.train from `training_data_loader.batch_size = rays_initial_batch_size` to
    `step_loss = loss.item()` (humanrf/trainer.py:138-177: the batch-growing loop over next(loader) + prune_samples, merge_input_batches,
    zero_grad, train_step under autocast), cut out of the reference's source at run time and compiled as they stand."""
    import inspect
    import textwrap
    src = inspect.getsource(ref.Trainer.train).splitlines()
    a = next(i for i, line in enumerate(src) if "rays_initial_batch_size" in line)
    b = next(i for i, line in enumerate(src) if "step_loss = loss.item()" in line)
    assert 25 <= b - a <= 60, (a, b)
    body = textwrap.dedent("\n".join(src[a:b + 1]))
    for needle in ("while True:", "next(training_data_loader_iter)", "prune_samples(", "merge_input_batches(", "self.train_step("):
        assert needle in body, needle
    return compile(body, "<humanrf/trainer.py:138-177>", "exec")


class _LoaderBridge:
    """SyntheticDataLoader seen through the reference's InputBatch dataclass (the loop sets .batch_size and calls next())."""

    def __init__(self, ref, loader):
        self.ref, self.loader = ref, loader

    batch_size = property(lambda self: self.loader.batch_size, lambda self, v: setattr(self.loader, "batch_size", int(v)))

    def __iter__(self):
        return self

    def __next__(self):
        return _ref_batch(self.ref, next(self.loader))


def test_reference_training_loop_body_over_the_dropins_trains_like_the_engine(ref):
    """VERDICT r04, missing item 6: the loop body of the reference's Trainer.train as a whole -- its own statements, compiled from
    its source -- over the drop-in loader / modules for 60 steps from the standard initialisation, next to 60
    TrainEngine.train_iteration() calls (StepCollector: one speculative march + hrf_batch_plan instead of the loop) from the
    same initial parameters on an identically seeded loader. The two draw different rays (the collector prefetches its draws and
    jitters with a counter-based stream), so the comparison is of what the loop is for: the sample budget it fills, the rays it
    needs for that, the iterations it takes, and where training gets to."""
    from humanrf_amd.trainer import TrainEngine
    SMAX, R0, STEPS = 120_000, 256, 60
    m, rm = _pair(ref, table_scale=None)
    _, ld_ref = _loader(batch=R0, seed=9)
    _, ld_own = _loader(batch=R0, seed=9)
    tr = RH.make_trainer(ref, rm, growth_interval=100_000, device="cuda")
    tr.config.training.rays_initial_batch_size = R0
    tr.config.training.samples_max_batch_size = SMAX
    body = _reference_loop_body(ref)
    bridge = _LoaderBridge(ref, ld_ref)
    scope = {"self": tr, "training_data_loader": bridge, "training_data_loader_iter": iter(bridge)}
    glob = vars(ref.modules.trainer)
    rm.train()
    torch.manual_seed(21)
    ref_rows = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # torch.cuda.amp.autocast() is the reference's spelling
        for _ in range(STEPS):
            exec(body, glob, scope)
            ib = scope["input_batch"]
            assert ib.num_samples <= int(SMAX * 1.1)
            assert scope["total_num_samples"] >= 0.9 * SMAX
            ref_rows.append((scope["total_num_rays"], ib.num_samples, len(scope["input_batches"]), scope["losses_info"]["psnr"]))
    assert tr.scaler.get_scale() == 65536.0

    eng = TrainEngine(m, loader=ld_own, samples_max_batch_size=SMAX, rays_initial_batch_size=R0)
    torch.manual_seed(21)
    own_rows = []
    for _ in range(STEPS):
        st = eng.train_iteration()
        assert st.num_samples <= int(SMAX * 1.1)
        own_rows.append((st.num_rays_drawn, st.num_samples, 0, TrainEngine.psnr_from_sums(st.sums, st.num_rays)))
    assert eng.found_inf() == 0
    col = eng.collector
    own_iters = (col.iterations_prefetched + col.iterations_classic) / STEPS
    R, O = np.array(ref_rows, dtype=np.float64), np.array(own_rows, dtype=np.float64)
    late = slice(STEPS // 2, STEPS)
    summary = dict(ref_rays=R[late, 0].mean(), own_rays=O[late, 0].mean(), ref_samples=R[late, 1].mean(), own_samples=O[late, 1].mean(),
                   ref_iters=R[:, 2].mean(), own_iters=own_iters, ref_psnr0=R[:5, 3].mean(), own_psnr0=O[:5, 3].mean(),
                   ref_psnr=R[-10:, 3].mean(), own_psnr=O[-10:, 3].mean())
    diag = os.environ.get("HRF_TEST_DIAG")
    if diag:
        with open(diag, "a") as f:
            f.write(f"reference loop body vs engine, {STEPS} steps: " + " ".join(f"{k}={v:.3f}" for k, v in summary.items()) + "\n")
    assert summary["ref_iters"] >= 1.9, summary          # (the configuration makes the loop grow its batch: 256 rays are a fifth of the budget)
    # the budget is filled the same way: samples per merged batch and drawn rays per step (second half of the run)
    assert abs(summary["ref_samples"] - summary["own_samples"]) <= 0.05 * SMAX, summary
    assert abs(summary["ref_rays"] - summary["own_rays"]) <= 0.15 * summary["ref_rays"], summary
    assert abs(summary["ref_iters"] - summary["own_iters"]) <= 0.6, summary
    # and training gets to the same place
    assert summary["ref_psnr"] >= summary["ref_psnr0"] + 2.0 and summary["own_psnr"] >= summary["own_psnr0"] + 2.0, summary
    assert abs(summary["ref_psnr"] - summary["own_psnr"]) <= 1.5, summary
