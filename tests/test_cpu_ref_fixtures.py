"""CPU checks against outputs of the REFERENCE's own code (tests/golden/ref_*.npz, frozen by
tests/golden/make_ref_fixtures.py from /root/reference through oracle/ref_harness.py):

  * the host-side product code (merge_input_batches, truncated_exp, bce_loss, segment sizing, frame tables, adaptive
    temporal partitioning) must reproduce them exactly;
  * the ORACLE must reproduce them: this is what pins oracle/hrf_oracle.py's composition (per-segment dispatch,
    +0.5, local time, compose pairing, density scale, colour-network input, jitter, alpha, visibility call, weights,
    accumulation, background, loss, optimizer wiring) to the reference instead of to our reading of it;
  * where /root/reference is present (build container), the same comparisons run LIVE on extra random cases.

Still unpinned after this (stated in DESIGN.md section 2): the arithmetic inside tinycudann's kernels, nerfacc's scan order, and
the CUDA texture unit -- the reference repository does not contain them."""
import os

import numpy as np
import pytest
import torch

from oracle import hrf_oracle as O
from oracle import ref_harness as RH
from tests import refcases as RC
from tests.golden import make_ref_fixtures as GEN

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_reference = pytest.mark.skipif(not RH.available(), reason="/root/reference only exists in the build container")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


# ------------------------------------------------------------------------------------------------ host-level product code
def _assert_batch_equal(merged, get):
    for k in GEN.BATCH_FIELDS:
        a, b = getattr(merged, k).numpy(), get(k)
        if k == "unique_frame_numbers":   # torch.unique(sorted=False): order is unspecified (input.py:50-53)
            a, b = np.sort(a.reshape(-1)), np.sort(b.reshape(-1))
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), k


def test_merge_input_batches_equals_reference_outputs():
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.input import merge_input_batches
    fx = _load("ref_host.npz")
    for seed in range(12):
        batches, max_n = GEN.merge_case_batches(InputBatch, 7000 + seed)
        merged = merge_input_batches(batches, max_num_samples=max_n)
        _assert_batch_equal(merged, lambda k: fx[f"merge{seed}_{k}"])


def test_truncated_exp_bce_and_segment_sizes_equal_reference_outputs():
    from humanrf_amd import adaptive_temporal_partitioning as atp
    from humanrf_amd.utils.activation import truncated_exp
    from humanrf_amd.utils.loss import bce_loss
    fx = _load("ref_host.npz")
    x = torch.from_numpy(fx["texp_x"]).requires_grad_()
    y = truncated_exp(x)
    (y * torch.from_numpy(fx["texp_w"])).sum().backward()
    assert np.array_equal(y.detach().numpy(), fx["texp_y"]) and np.array_equal(x.grad.numpy(), fx["texp_dx"])
    yo = O.truncated_exp(torch.from_numpy(fx["texp_x"]))
    assert np.array_equal(yo.numpy(), fx["texp_y"])
    out = bce_loss(torch.from_numpy(fx["bce_pred"]), torch.from_numpy(fx["bce_target"])).numpy()
    assert np.array_equal(out, fx["bce_out"])
    assert [atp.get_segment_size(n) for n in range(1, 131)] == fx["segsize_table"].tolist()
    assert [atp.get_final_segment_size(n) for n in range(1, 101)] == fx["final_segsize_table"].tolist()


def test_frame_tables_equal_reference_constructor():
    from humanrf_amd.scene_representation import hashgrid
    fx = _load("ref_host.npz")
    for i, (frames, segs) in enumerate(GEN.FRAME_TABLE_CASES):
        f2s, f2l = hashgrid.frame_tables(frames, segs)
        assert np.array_equal(f2s, fx[f"ft{i}_f2s"]) and f2s.dtype == fx[f"ft{i}_f2s"].dtype
        assert np.array_equal(f2l, fx[f"ft{i}_f2l"]) and f2l.dtype == fx[f"ft{i}_f2l"].dtype


def test_adaptive_partitioning_equals_reference_outputs():
    from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
    fx = _load("ref_host.npz")
    G = 32
    for i, (nf, thr) in enumerate(GEN.ATP_CASES):
        bits = np.unpackbits(fx[f"atpgrids{nf}"])[:nf * G ** 3].reshape(nf, G, G, G)
        grids = {15 + k: torch.from_numpy(bits[k] * np.uint8(255)) for k in range(nf)}
        assert [int(g.sum()) // 255 for g in grids.values()] == fx[f"atp{i}_popcounts"].tolist()
        sizes = compute_adaptive_segment_sizes(lambda f: grids[f], list(range(15, 15 + nf)), thr)
        assert sizes == fx[f"atp{i}_sizes"].tolist(), (nf, thr)


# ------------------------------------------------------------------------------------------------ oracle vs reference: field
def _field_oracle(name, requires_grad):
    from humanrf_amd.scene_representation import hashgrid
    inp = RC.field_inputs(name)
    sd = RC.seeded_reference_state(inp["segment_sizes"], inp["log2_T"], inp["emb"], seed=500 + len(name))
    f2s, f2l = hashgrid.frame_tables(inp["sorted_frames"], inp["segment_sizes"])
    om = RC.oracle_model_from_state(sd, inp["sorted_frames"], inp["segment_sizes"], inp["log2_T"], inp["emb"],
                                    torch.from_numpy(f2s), torch.from_numpy(f2l), requires_grad)
    return inp, om


def _rel(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("name", list(RC.FIELD_CASES))
def test_oracle_field_equals_reference_humanrf(name):
    """HumanRF.density / forward and Decomposition4D.forward as the reference composes them (humanrf.py:158-208,
    decomposition4d.py:124-135) == oracle.model_density / model_forward, values exactly, gradients to fp16-gradient noise."""
    fx = _load(f"ref_field_{name}.npz")
    inp, om = _field_oracle(name, requires_grad=True)
    pos, frames, cams, dirs = inp["positions"], inp["frames"], inp["cams"], inp["directions"]
    with torch.no_grad():
        feats = O.model_features(om, pos, frames)
        assert np.array_equal(feats.half().numpy(), fx["d4_features"])
        _, rgb_eval = O.model_forward(om, pos, dirs, frames, cams, False)
        assert np.array_equal(rgb_eval.half().numpy(), fx["radiance_eval"])
    sigma, geo, _ = O.model_density(om, pos, frames)
    assert np.array_equal(sigma.detach().numpy(), fx["density"]) and np.array_equal(geo.detach().half().numpy(), fx["geo"])
    sigma, rgb = O.model_forward(om, pos, dirs, frames, cams, True)
    assert np.array_equal(rgb.detach().half().numpy(), fx["radiance"])
    loss = (sigma * inp["a"]).sum() + (rgb * inp["b"]).sum()
    assert abs(float(loss) - float(fx["loss"][0])) <= 1e-4 * abs(float(fx["loss"][0]))
    loss.backward()
    # the reference's gradients pass through real fp16 tensors (features, geometry features): ~2^-11 relative noise per
    # element that the oracle's straight-through rounding does not have
    tol = 4e-3
    assert _rel(torch.cat([w.grad.reshape(-1) for w in om.sigma_w]).numpy(), fx["g_sigma"]) <= tol
    assert _rel(torch.cat([w.grad.reshape(-1) for w in om.color_w]).numpy(), fx["g_color"]) <= tol
    if inp["emb"] > 0:
        assert _rel(om.camera_embeddings.grad.numpy(), fx["g_emb"]) <= tol
    for s in range(len(inp["segment_sizes"])):
        assert _rel(om.vectors[s].grad[:, ::16, :].numpy(), fx[f"g_vec{s}"]) <= tol
        for e in range(4):
            g = om.tables[s][e].grad.reshape(-1)
            assert int((g != 0).sum()) == int(fx[f"g_tab{s}_{e}_nnz"][0]), "a different set of table entries was touched"
            idx = fx[f"g_tab{s}_{e}_idx"]
            if idx.size:
                assert _rel(g[idx].numpy(), fx[f"g_tab{s}_{e}_val"]) <= tol
            for l, lv in enumerate(om.levels[s]):
                n_ref = fx["g_tab_level_norms"][s, e, l]
                n_or = float(g[2 * lv.offset:2 * (lv.offset + lv.size)].double().norm())
                assert abs(n_or - n_ref) <= tol * max(n_ref, 1e-12), (s, e, l)


# ------------------------------------------------------------------------------------------------ oracle vs reference: render + step
def _render_case():
    from humanrf_amd.scene_representation import hashgrid
    fx = _load("ref_render.npz")
    sd = RC.seeded_reference_state(GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, seed=77, table_scale=0.3, vec_scale=0.4)
    f2s, f2l = hashgrid.frame_tables(GEN.RENDER_FRAMES, GEN.RENDER_SEGS)
    return fx, sd, torch.from_numpy(f2s), torch.from_numpy(f2l)


def test_oracle_sampler_equals_fixture_inputs():
    """The sampler outputs the reference's prune/render were fed with are the C oracle's (regenerated here)."""
    fx = _load("ref_render.npz")
    s = O.sampler_get_data(fx["in_rgba"], None, fx["in_frames"], fx["in_cams"], list(fx["in_grids"]), np.ones(4, bool),
                           fx["in_idx"], fx["in_inverse_krs"], fx["in_camera_origins"], fx["in_aabb"], int(fx["in_G"]),
                           int(fx["in_W"]), int(fx["in_H"]), 4e-4, False, True, True)
    for nm, a in zip(("origins", "dirs", "rgba_s", "frames_s", "cams_s", "minmax", "ray_mask", "t", "ray"), s):
        assert np.array_equal(a, fx["smp_" + nm]), nm


def test_oracle_prune_and_render_equal_reference_volume_rendering():
    """prune_samples / render of humanrf/volume_rendering.py:42-150 == oracle.prune_samples / render."""
    fx, sd, f2s, f2l = _render_case()
    om = RC.oracle_model_from_state(sd, GEN.RENDER_FRAMES, GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, f2s, f2l)
    t = lambda k: torch.from_numpy(fx[k])
    o, d, fr, cm = t("smp_origins"), t("smp_dirs"), t("smp_frames_s"), t("smp_cams_s")
    t0, ray = t("smp_t"), t("smp_ray").long()
    # evaluation
    tj, _, vis, _ = O.prune_samples(om, o, d, fr, t0, ray, None)
    assert np.array_equal(tj[vis].numpy(), fx["eval_t"]) and np.array_equal(ray[vis].numpy(), fx["eval_ray"])
    with torch.no_grad():
        color, acc = O.render(om, o, d, fr, cm, tj[vis], ray[vis], torch.zeros(o.shape[0], 3), False)
    assert np.allclose(color.numpy(), fx["eval_color"], atol=1e-6) and np.allclose(acc.numpy(), fx["eval_acc"], atol=1e-6)
    # training
    tj, _, vis, _ = O.prune_samples(om, o, d, fr, t0, ray, t("jitter"))
    assert np.array_equal(tj[vis].numpy(), fx["train_t"]) and np.array_equal(ray[vis].numpy(), fx["train_ray"])
    assert 0.05 < vis.float().mean() < 0.95, "degenerate case: pruning keeps everything or nothing"


def oracle_train_steps(fx, sd, f2s, f2l, steps=3, half_gradient_scale=0.0):
    """Three optimizer steps of oracle autograd + torch.optim.Adam + LambdaLR on the fixture's batch
    (what tests/test_gpu_ref_fixtures.py also compares the HIP engine with) -> {param name: (p, m, v)} per step."""
    om = RC.oracle_model_from_state(sd, GEN.RENDER_FRAMES, GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, f2s, f2l)
    om.half_gradient_scale = half_gradient_scale
    # fp32 masters as the optimizer sees them; the oracle reads fp16 copies of tables / MLP weights (tcnn)
    masters = {k: v.clone().requires_grad_() for k, v in sd.items()}
    opt = torch.optim.Adam(list(masters.values()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: 0.5 ** min(step / 50_001, 1))
    t = lambda k: torch.from_numpy(fx[k])
    o, d, fr, cm, rgba = t("smp_origins"), t("smp_dirs"), t("smp_frames_s"), t("smp_cams_s"), t("smp_rgba_s")
    tt, ray = t("train_t"), t("train_ray")
    kin = 16 * ((31 + GEN.RENDER_EMB + 15) // 16)
    out = []
    for step in range(steps):
        opt.zero_grad(set_to_none=True)
        for s in range(len(GEN.RENDER_SEGS)):
            om.vectors[s] = masters[f"feature_grids.{s}.vectors"]
            om.tables[s] = [O.round_half(masters[f"feature_grids.{s}.{nm}_encoding.params"]).reshape(-1, 2) for nm in RC.ENC_NAMES]
        sw, cw = O.round_half(masters["sigma_net.params"]), O.round_half(masters["color_net.params"])
        om.sigma_w = [sw[:2048].reshape(64, 32), sw[2048:].reshape(16, 64)]
        om.color_w = [cw[:64 * kin].reshape(64, kin), cw[64 * kin:64 * kin + 4096].reshape(64, 64), cw[64 * kin + 4096:].reshape(16, 64)]
        om.camera_embeddings = masters["camera_embeddings.weight"]
        bg = t(f"bg{step}")
        color, acc = O.render(om, o, d, fr, cm, tt, ray, bg, True)
        loss, photo = O.training_loss(color, acc, rgba, bg)
        loss.backward()
        opt.step()
        sched.step()
        out.append((float(loss), float(photo), {k: (p.detach().clone(), opt.state[p]["exp_avg"].clone(),
                                                    opt.state[p]["exp_avg_sq"].clone()) for k, p in masters.items()}))
    return out


def test_oracle_train_step_equals_reference_trainer():
    """Trainer.train_step x3 (trainer.py:229-255: random background, render, Huber + 1e-3 BCE, GradScaler, Adam, LambdaLR)
    == oracle.render + oracle.training_loss + autograd + torch.optim.Adam, to the noise of the reference's fp16 gradients."""
    fx, sd, f2s, f2l = _render_case()
    # (the oracle differentiates the unscaled loss: its half gradient tensors around the compose op sit at the GradScaler's 65536)
    res = oracle_train_steps(fx, sd, f2s, f2l, half_gradient_scale=65536.0)
    names = [str(n) for n in fx["param_names"]]
    for step, (loss, photo, state) in enumerate(res):
        assert abs(loss - fx[f"loss{step}"][0]) <= 2e-5 * max(abs(fx[f"loss{step}"][0]), 1e-3) + 1e-7
        assert abs(photo - fx[f"loss{step}"][1]) <= 2e-5 * max(abs(fx[f"loss{step}"][1]), 1e-3) + 1e-7
        for n in names:
            pick = RC.sample_indices(state[n][0].numel(), 4096, seed=len(n))
            p, m, v = (x.view(-1)[pick].numpy() for x in state[n])
            assert _rel(m, fx[f"s{step}|{n}|m"]) <= 6e-3, (step, n, "exp_avg")
            assert _rel(np.sqrt(v), np.sqrt(fx[f"s{step}|{n}|v"])) <= 6e-3, (step, n, "exp_avg_sq")
            # Adam's first steps move every touched parameter by ~lr * sign(g): compare the update, not the value
            p0 = sd[n].view(-1)[pick].numpy()
            du, dr = p - p0, fx[f"s{step}|{n}|p"] - p0
            touched = np.abs(dr) > 0
            agree = np.mean(np.sign(du[touched]) == np.sign(dr[touched])) if touched.any() else 1.0
            assert agree >= 0.995, (step, n, agree)
            assert _rel(du, dr) <= 0.08, (step, n, _rel(du, dr))


def test_half_gradient_boundaries_move_exactly_the_entries_the_reference_moves():
    """The rule TrainEngine(gradient_boundaries="fp16") and include/hrf.h's grad_boundary implement, pinned on the CPU against the
    reference's own Trainer.train_step (fixture ref_step_weak.npz: GradScaler 65536, the compose op's real torch.half tensors,
    1 239 rays so that part of the table entries only receives contributions below the half floor): with the gradients of the
    compose output and of its four per-encoding inputs rounded through half at the GradScaler's scale (oracle.half_gradient), the
    oracle's first Adam step moves the SAME table entries as the reference's; with fp32 gradients throughout (the fused backward
    of round 3) it moves entries the reference leaves alone -- Adam (eps 1e-15) steps any non-zero gradient by lr."""
    fx = _load("ref_step_weak.npz")
    _, smp = GEN.weak_sampler_outputs()
    org, dirs, rgba, frames, cams = (torch.from_numpy(np.ascontiguousarray(a)) for a in smp[:5])
    assert org.shape[0] == int(fx["num_rays"][0])
    from humanrf_amd.scene_representation import hashgrid
    f2s, f2l = (torch.from_numpy(a) for a in hashgrid.frame_tables(GEN.RENDER_FRAMES, GEN.RENDER_SEGS))
    sd = RC.seeded_reference_state(GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, seed=79, table_scale=0.1, vec_scale=0.4)
    names = [str(n) for n in fx["param_names"]]
    kin = 16 * ((31 + GEN.RENDER_EMB + 15) // 16)
    tt, ray, bg = torch.from_numpy(fx["t"]), torch.from_numpy(fx["ray"]), torch.from_numpy(fx["bg"])
    moved_ref, only_own, only_ref = 0, {"fp32": 0, "fp16": 0}, {"fp32": 0, "fp16": 0}
    for mode, scale in (("fp32", 0.0), ("fp16", 65536.0)):
        om = RC.oracle_model_from_state(sd, GEN.RENDER_FRAMES, GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, f2s, f2l)
        om.half_gradient_scale = scale
        masters = {k: v.clone().requires_grad_() for k, v in sd.items()}
        opt = torch.optim.Adam(list(masters.values()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        for s in range(len(GEN.RENDER_SEGS)):
            om.vectors[s] = masters[f"feature_grids.{s}.vectors"]
            om.tables[s] = [O.round_half(masters[f"feature_grids.{s}.{nm}_encoding.params"]).reshape(-1, 2) for nm in RC.ENC_NAMES]
        sw, cw = O.round_half(masters["sigma_net.params"]), O.round_half(masters["color_net.params"])
        om.sigma_w = [sw[:2048].reshape(64, 32), sw[2048:].reshape(16, 64)]
        om.color_w = [cw[:64 * kin].reshape(64, kin), cw[64 * kin:64 * kin + 4096].reshape(64, 64), cw[64 * kin + 4096:].reshape(16, 64)]
        om.camera_embeddings = masters["camera_embeddings.weight"]
        color, acc = O.render(om, org, dirs, frames, cams, tt, ray, bg, True)
        loss, _ = O.training_loss(color, acc, rgba, bg)
        assert abs(float(loss) - fx["loss"][0]) <= 2e-5 * abs(fx["loss"][0]) + 1e-7
        loss.backward()
        opt.step()
        for n in names:
            pick = RC.sample_indices(sd[n].numel(), 8192, seed=len(n) + 2)
            p0 = sd[n].view(-1)[pick].numpy()
            du, dr = masters[n].detach().view(-1)[pick].numpy() - p0, fx[f"{n}|p"] - p0
            if mode == "fp32":
                moved_ref += int((dr != 0).sum())
            only_own[mode] += int(((du != 0) & (dr == 0)).sum())
            only_ref[mode] += int(((du == 0) & (dr != 0)).sum())
    assert moved_ref > 10_000
    # half boundaries: the same entries (measured: 39 001 moved, none on one side only); fp32 gradients: 258 entries move here only
    assert only_own["fp16"] <= 5 and only_ref["fp16"] <= 5, (moved_ref, only_own, only_ref)
    assert only_ref["fp32"] <= 5 and only_own["fp32"] >= 100, (moved_ref, only_own, only_ref)
    print("moved by the reference (sampled):", moved_ref, "here only:", only_own, "there only:", only_ref)


def oracle_skip_steps(fx, sd, f2s, f2l):
    """The untouched-segment scenario through oracle autograd + torch.optim.Adam: parameters of a segment no ray of the
    batch belongs to are not part of the graph, get grad None and are skipped by Adam, like in the reference."""
    from tests.golden.make_ref_fixtures import SKIP_SEQUENCE, skip_batches
    om = RC.oracle_model_from_state(sd, GEN.RENDER_FRAMES, GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, f2s, f2l)
    om.half_gradient_scale = 65536.0
    masters = {k: v.clone().requires_grad_() for k, v in sd.items()}
    opt = torch.optim.Adam(list(masters.values()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: 0.5 ** min(step / 50_001, 1))
    inp = {k[3:]: fx[k] for k in fx.files if k.startswith("in_")}
    smp = skip_batches(inp)
    kin = 16 * ((31 + GEN.RENDER_EMB + 15) // 16)
    t = torch.from_numpy
    out = []
    for step, key in enumerate(SKIP_SEQUENCE):
        org, dirs, rgba, frames, cams = (t(np.ascontiguousarray(a)) for a in smp[key][:5])
        opt.zero_grad(set_to_none=True)
        for s in range(len(GEN.RENDER_SEGS)):
            om.vectors[s] = masters[f"feature_grids.{s}.vectors"]
            om.tables[s] = [O.round_half(masters[f"feature_grids.{s}.{nm}_encoding.params"]).reshape(-1, 2) for nm in RC.ENC_NAMES]
        sw, cw = O.round_half(masters["sigma_net.params"]), O.round_half(masters["color_net.params"])
        om.sigma_w = [sw[:2048].reshape(64, 32), sw[2048:].reshape(16, 64)]
        om.color_w = [cw[:64 * kin].reshape(64, kin), cw[64 * kin:64 * kin + 4096].reshape(64, 64), cw[64 * kin + 4096:].reshape(16, 64)]
        om.camera_embeddings = masters["camera_embeddings.weight"]
        bg = t(fx_skip_bg(step))
        color, acc = O.render(om, org, dirs, frames, cams, t(_SKIP[f"{key}_t"]), t(_SKIP[f"{key}_ray"]), bg, True)
        loss, photo = O.training_loss(color, acc, rgba, bg)
        loss.backward()
        opt.step()
        sched.step()
        snap = lambda x: None if x is None else x.clone()
        out.append((float(loss), {k: (p.detach().clone(), snap(opt.state[p].get("exp_avg")), snap(opt.state[p].get("exp_avg_sq")),
                                      int(opt.state[p]["step"]) if "step" in opt.state[p] else 0) for k, p in masters.items()}))
    return out


_SKIP = None


def fx_skip_bg(step):
    return _SKIP[f"bg{step}"]


def test_oracle_untouched_segments_equal_reference_trainer():
    """Segments without rays in the batch: no gradient, Adam skips them, their step count does not advance
    (humanrf.py:159-179, trainer.py:174, torch.optim.Adam) -- reference Trainer.train_step vs oracle + torch Adam."""
    global _SKIP
    _SKIP = _load("ref_steps_skip.npz")
    fx_in = _load("ref_render.npz")
    from humanrf_amd.scene_representation import hashgrid
    sd = RC.seeded_reference_state(GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, seed=78, table_scale=0.3, vec_scale=0.4)
    f2s, f2l = hashgrid.frame_tables(GEN.RENDER_FRAMES, GEN.RENDER_SEGS)
    res = oracle_skip_steps(fx_in, sd, torch.from_numpy(f2s), torch.from_numpy(f2l))
    names = [str(n) for n in _SKIP["param_names"]]
    expected_steps = {0: {"0": 1, "1": 0}, 1: {"0": 1, "1": 1}, 2: {"0": 2, "1": 1}, 3: {"0": 3, "1": 2}}
    for step, (loss, state) in enumerate(res):
        assert abs(loss - _SKIP[f"loss{step}"][0]) <= 2e-5 * max(abs(_SKIP[f"loss{step}"][0]), 1e-3) + 1e-7
        for n in names:
            p, m, v, tcount = state[n]
            assert tcount == int(_SKIP[f"s{step}|{n}|t"][0]), (step, n)
            if n.startswith("feature_grids."):
                assert tcount == expected_steps[step][n.split(".")[1]]
            pick = RC.sample_indices(p.numel(), 2048, seed=len(n) + 1)
            p0 = sd[n].view(-1)[pick].numpy()
            du, dr = p.view(-1)[pick].numpy() - p0, _SKIP[f"s{step}|{n}|p"] - p0
            if tcount == 0:
                assert not du.any() and not dr.any()          # never stepped: bit-identical to the initial values
                continue
            assert _rel(m.view(-1)[pick].numpy(), _SKIP[f"s{step}|{n}|m"]) <= 6e-3, (step, n)
            assert _rel(du, dr) <= 0.08, (step, n, _rel(du, dr))


def test_image_assembly_and_psnr_equal_reference_trainer():
    """combine_rays_to_image (trainer.py:517-526) and the PSNR of _calculate_losses (trainer.py:218-223)."""
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.inference import combine_rays_to_image, psnr_of_rendered_rays
    from humanrf_amd.volume_rendering import RenderOutput
    fx = _load("ref_render.npz")
    t = torch.from_numpy
    full = InputBatch(ray_masks=t(fx["smp_ray_mask"]).view(-1, 1), rgba=t(fx["smp_rgba_s"]), width=int(fx["in_idx"].shape[0]), height=1)
    ro = RenderOutput(color=t(fx["eval_color"]), weights_sum=t(fx["eval_acc"]))
    assert np.array_equal(combine_rays_to_image(full, ro, 0).numpy(), fx["eval_image"])
    assert abs(psnr_of_rendered_rays(ro, full.rgba, 0.0) - float(fx["eval_psnr"][0])) <= 1e-4


# ------------------------------------------------------------------------------------------------ live against /root/reference
@needs_reference
def test_live_reference_modules_are_the_reference_files():
    ref = RH.load()
    for mod in vars(ref.modules).values():
        assert os.path.abspath(mod.__file__).startswith(os.path.abspath(RH.REFERENCE_ROOT))
    import nerfacc
    import tinycudann
    assert getattr(tinycudann, "__stub__", False) and getattr(nerfacc, "__stub__", False)


@needs_reference
def test_live_host_functions_against_imported_reference():
    from humanrf_amd import adaptive_temporal_partitioning as atp
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.input import merge_input_batches
    from humanrf_amd.scene_representation import hashgrid
    from humanrf_amd.utils.activation import truncated_exp
    from humanrf_amd.utils.loss import bce_loss
    ref = RH.load()
    for seed in range(100, 160):
        mine, max_n = GEN.merge_case_batches(InputBatch, seed)
        theirs, _ = GEN.merge_case_batches(ref.InputBatch, seed)
        a, b = merge_input_batches(mine, max_n), ref.merge_input_batches(theirs, max_num_samples=max_n)
        _assert_batch_equal(a, lambda k: getattr(b, k).numpy())
    g = RC.rng(5)
    x = torch.from_numpy((g.standard_normal(500) * 12).astype(np.float32))
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    truncated_exp(xa).sum().backward(); ref.truncated_exp(xb).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    pred, tgt = torch.from_numpy(g.random(500, dtype=np.float32) * 1.2 - 0.1), torch.from_numpy((g.random(500) < 0.5).astype(np.float32))
    assert torch.equal(bce_loss(pred, tgt), ref.bce_loss(pred, tgt))
    for trial in range(20):
        nf = int(g.integers(1, 140))
        frames = tuple(np.sort(g.choice(np.arange(5, 400), nf, replace=False)).tolist())
        segs, left = [], nf
        while left > 0:
            s = int(g.choice(ref.PREDEFINED_SEGMENT_SIZES))
            segs.append(s); left -= s
        m = RH.make_model(ref, frames, segs, log2_T=10, emb=0)
        f2s, f2l = hashgrid.frame_tables(frames, segs)
        assert np.array_equal(f2s, m.frame_numbers_to_segment_numbers.numpy())
        assert np.array_equal(f2l, m.frame_numbers_to_normalized_local_frame_numbers.numpy())
    G = 12
    for trial in range(25):
        nf = int(g.integers(1, 120))
        base = g.random((G, G, G)) < 0.15
        grids = {}
        for k in range(nf):
            base = base | (g.random((G, G, G)) < float(g.choice([0.0, 0.002, 0.02])))
            if g.random() < 0.08:
                base = g.random((G, G, G)) < 0.15
            grids[k] = (base * np.uint8(255)).astype(np.uint8)
        thr = float(g.choice([1.02, 1.1, 1.25, 1.6]))
        theirs = ref.compute_adaptive_segment_sizes(RH.GridDataset(grids), list(range(nf)), thr)
        mine = atp.compute_adaptive_segment_sizes(lambda f: torch.from_numpy(grids[f]), list(range(nf)), thr)
        assert mine == theirs, (trial, nf, thr)


@needs_reference
def test_live_fixtures_are_current():
    """Regenerating a fixture from the reference gives the committed file's content (guards against a stale fixture)."""
    ref = RH.load()
    fx = _load("ref_field_seg12_T15.npz")
    inp = RC.field_inputs("seg12_T15")
    sd = RC.seeded_reference_state(inp["segment_sizes"], inp["log2_T"], inp["emb"], seed=500 + len("seg12_T15"))
    model = RH.make_model(ref, inp["sorted_frames"], inp["segment_sizes"], log2_T=inp["log2_T"], emb=inp["emb"])
    model.load_state_dict(sd, strict=False)
    qi = ref.QueryInput(is_training=True, positions=inp["positions"], directions=inp["directions"], frame_numbers=inp["frames"],
                        unique_frame_numbers=torch.unique(inp["frames"]).view(-1, 1), camera_numbers=inp["cams"])
    with torch.no_grad():
        q = model(qi)
    assert np.array_equal(q.density.numpy(), fx["density"]) and np.array_equal(q.radiance.half().numpy(), fx["radiance"])


def test_reference_state_dict_keys_match_the_reference_module_tree():
    """Key names / shapes of humanrf_amd's reference_state_dict() == state_dict() of the reference's HumanRF module tree
    (recorded in the fixture as `param_names`; the flat layout INSIDE each tcnn `params` vector stays upstream knowledge)."""
    from tests.util import make_model
    fx = _load("ref_render.npz")
    m = make_model("cpu", GEN.RENDER_SEGS, GEN.RENDER_FRAMES, log2_T=GEN.RENDER_LOG2T, emb=GEN.RENDER_EMB)
    mine = m.reference_state_dict()
    names = {str(n) for n in fx["param_names"]}
    assert names | {"frame_numbers_to_segment_numbers", "frame_numbers_to_normalized_local_frame_numbers"} == set(mine)
    sd = RC.seeded_reference_state(GEN.RENDER_SEGS, GEN.RENDER_LOG2T, GEN.RENDER_EMB, seed=77)
    for n in names:
        assert mine[n].shape == sd[n].shape, n
