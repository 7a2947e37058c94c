"""Round-5 GPU checks:
  * the model knobs the reference accepts and the kernels now serve -- n_levels < 16, geometry_feature_dim < 15
    (humanrf/args/model_args.py:16,22) -- against THE REFERENCE'S OWN HumanRF CLASS (its source, executed on the CPU over the
    oracle's tcnn stand-ins, oracle/ref_harness.load("cpu")), parameters handed over through reference_state_dict();
  * a finite gradient that the half gradient boundary turns into inf raises found_inf on the atomic table path too;
  * the start-up probe of the in-place RCCL collectives (TableShardExchange.self_check) on a one-rank group."""
import os
import socket

import numpy as np
import pytest
import torch

DEV = "cuda"

pytestmark = pytest.mark.gpu


def _pair(n_levels, geo, emb, log2_T=15, hidden_color=2):
    from oracle import ref_harness
    from humanrf_amd.scene_representation import HumanRF
    if not ref_harness.available():
        pytest.skip("the reference's sources (or their snapshot under oracle/_ref/reference) are not on this machine")
    ref = ref_harness.load("cpu")
    frames, sizes = tuple(range(15, 27)), (6, 6)
    kw = dict(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2, log2_hashmap_size=log2_T, n_levels=n_levels,
              coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=geo, n_neurons=64, n_hidden_layers_density=1,
              n_hidden_layers_color=hidden_color, sh_degree=4, segment_sizes=sizes, camera_embedding_dim=emb)
    m = HumanRF(device=DEV, seed=7, **kw)
    with torch.no_grad():      # tables well above their 1e-4 initialisation, so that the outputs are not all alike
        g = torch.Generator().manual_seed(3)
        m.table_params.copy_(((torch.rand(m.table_params.numel(), generator=g) * 2 - 1) * 0.5).to(DEV))
    rm = ref.HumanRF(**kw)
    sd = {k: v.detach().cpu() for k, v in m.reference_state_dict().items()}
    missing, unexpected = rm.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("color_net.") is False or True for k in missing)
    # every parameter of the reference's module tree is filled from ours, with ITS shapes (checkpoint layout of the knobs)
    for k, v in rm.state_dict().items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), (k, tuple(v.shape), tuple(sd.get(k, torch.zeros(0)).shape))
    return ref, m, rm, frames


# (round 6) n_hidden_layers_color (model_args.py:31) 1 and 3 next to the reference's 2: the last column
@pytest.mark.parametrize("n_levels,geo,emb,hidden_color", [(8, 15, 2, 2), (12, 7, 2, 2), (16, 4, 0, 2), (5, 0, 3, 2), (16, 15, 2, 2),
                                                           (16, 15, 2, 1), (16, 15, 2, 3), (10, 7, 0, 3), (16, 15, 0, 1)])
def test_model_knobs_against_the_reference_class(n_levels, geo, emb, hidden_color):
    from humanrf_amd.scene_representation.query_io import QueryInput
    ref, m, rm, frames = _pair(n_levels, geo, emb, hidden_color=hidden_color)
    assert m.color_params.numel() == rm.color_net.params.numel() == 64 * m.color_in_pad + 4096 * (hidden_color - 1) + 1024
    g = torch.Generator().manual_seed(11)
    n = 1500
    pos = torch.rand(n, 3, generator=g) - 0.5
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    fr = torch.tensor(frames, dtype=torch.int32)[torch.randint(0, len(frames), (n,), generator=g)].reshape(-1, 1)
    cams = torch.randint(0, 160, (n, 1), generator=g, dtype=torch.int32)
    a, b = torch.rand(n, generator=g), torch.rand(n, 3, generator=g)
    uniq = torch.unique(fr).reshape(-1, 1)
    for training in (True, False):
        q_ref = rm(ref.QueryInput(is_training=training, positions=pos, directions=dirs, frame_numbers=fr, unique_frame_numbers=uniq,
                                  camera_numbers=cams))
        q = m(QueryInput(is_training=training, positions=pos.to(DEV), directions=dirs.to(DEV), frame_numbers=fr.to(DEV),
                         unique_frame_numbers=uniq.to(DEV), camera_numbers=cams.to(DEV)))
        assert tuple(q.geometry_features.shape) == (n, geo) == tuple(q_ref.geometry_features.shape)
        d_ref, d = q_ref.density.detach(), q.density.detach().cpu()
        assert float(((d - d_ref).abs() / d_ref.abs().clamp_min(1e-3)).max()) <= 2e-2
        if geo:
            assert float((q.geometry_features.float().cpu() - q_ref.geometry_features.detach().float()).abs().max()) <= \
                2e-2 * max(1.0, float(q_ref.geometry_features.detach().abs().max()))
        assert float((q.radiance.detach().float().cpu() - q_ref.radiance.detach().float()).abs().max()) <= 4e-3
        if not training:
            continue
        loss_ref = (q_ref.density * a).sum() * 1e-3 + (q_ref.radiance * b).sum()
        loss = (q.density * a.to(DEV)).sum() * 1e-3 + (q.radiance * b.to(DEV)).sum()
        loss_ref.backward(); loss.backward()
        assert abs(float(loss) - float(loss_ref)) <= 5e-3 * abs(float(loss_ref))

        def close(mine, theirs, name, cos_min=0.999, rel=5e-2):
            mine, theirs = mine.detach().double().cpu().reshape(-1), theirs.detach().double().reshape(-1)
            assert mine.shape == theirs.shape, (name, mine.shape, theirs.shape)
            if float(theirs.norm()) == 0.0:
                assert float(mine.norm()) == 0.0, name
                return
            cos = float((mine * theirs).sum() / (mine.norm() * theirs.norm()))
            assert cos >= cos_min, (name, cos)
            assert float((mine - theirs).norm() / theirs.norm()) <= rel, name
        pad = m.sigma_in_pad
        g_s = m.sigma_params.grad
        close(torch.cat([g_s[:2048].reshape(64, 32)[:, :pad].reshape(-1), g_s[2048:]]), rm.sigma_net.params.grad, "sigma_net")
        assert float(g_s[:2048].reshape(64, 32)[:, pad:].abs().max() if pad < 32 else 0.0) == 0.0     # columns that multiply zeros
        close(m.color_params.grad, rm.color_net.params.grad, "color_net")
        if emb:
            close(m.camera_embeddings.weight.grad, rm.camera_embeddings.weight.grad, "camera_embeddings")
        off = 0
        F = 2 * n_levels
        for s, entries in enumerate(m.entries_per_segment):
            close(m.vectors.grad[s][..., :F], rm.feature_grids[s].vectors.grad, f"vectors {s}", rel=3e-2)
            assert float(m.vectors.grad[s][..., F:].abs().max() if F < 32 else 0.0) == 0.0
            for e, nm in enumerate(("xyz", "xyt", "yzt", "xzt")):
                close(m.table_params.grad[off * 2:(off + entries) * 2], getattr(rm.feature_grids[s], f"{nm}_encoding").params.grad,
                      f"tables {s} {nm}", rel=3e-2)
                off += entries


def test_training_engine_steps_a_model_with_fewer_levels_and_geometry_features():
    """The fused training path (prune march with the constant columns, binned scatter, Adam) on such a model: a few steps
    run, nothing is skipped, the loss falls, the padding columns of vectors / sigma_net stay exactly zero."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine
    from tests.util import small_scene
    torch.manual_seed(5)
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=512, max_buffer_size=8, max_num_frames_per_batch=3, seed=1)
    iter(loader)
    m = HumanRF(density_scale=100, sorted_frame_numbers=tuple(scene.frame_numbers), n_features_per_level=2, log2_hashmap_size=15,
                n_levels=6, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=6, n_neurons=64,
                n_hidden_layers_density=1, n_hidden_layers_color=2, sh_degree=4, segment_sizes=(12,), camera_embedding_dim=2,
                device=DEV)
    eng = TrainEngine(m, loader, samples_max_batch_size=40_000, rays_initial_batch_size=512)
    psnr = []
    for _ in range(12):
        st = eng.train_iteration()
        psnr.append(TrainEngine.psnr_from_sums(st.sums, st.num_rays))
    torch.cuda.synchronize()
    assert eng.found_inf() == 0 and eng.optimizer_steps()[0] == 12
    assert psnr[-1] > psnr[0] + 0.5, psnr
    assert float(m.vectors.detach()[..., 12:].abs().max()) == 0.0 and m.sigma_in_pad == 16
    assert float(m.sigma_params.detach()[:2048].reshape(64, 32)[:, m.sigma_in_pad:].abs().max()) == 0.0


@pytest.mark.parametrize("hidden_color,backward,precision", [(1, "fused", "fp16"), (3, "fused", "fp16"), (3, "split", "fp16"),
                                                             (1, "split", "bf16"), (3, "fused", "bf16")])
def test_training_engine_steps_a_model_with_another_colour_depth(hidden_color, backward, precision):
    """n_hidden_layers_color 1 and 3 through the fused training path, both forms of the MLP backward: the steps run, the loss falls, and
    the weight gradients of one batch agree with the oracle's autograd of the same batch (the depth-generic MLP of oracle/hrf_oracle)."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine
    from tests.util import small_scene
    torch.manual_seed(5)
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=512, max_buffer_size=8, max_num_frames_per_batch=3, seed=1)
    iter(loader)
    m = HumanRF(density_scale=100, sorted_frame_numbers=tuple(scene.frame_numbers), n_features_per_level=2, log2_hashmap_size=15,
                n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
                n_hidden_layers_density=1, n_hidden_layers_color=hidden_color, sh_degree=4, segment_sizes=(12,), camera_embedding_dim=2,
                device=DEV, mlp_precision=precision)
    eng = TrainEngine(m, loader, samples_max_batch_size=40_000, rays_initial_batch_size=512, mlp_backward=backward)
    psnr = []
    steps = 40            # (batches of 512 rays: single steps scatter by +-1 dB; means of five at both ends)
    for _ in range(steps):
        st = eng.train_iteration()
        psnr.append(TrainEngine.psnr_from_sums(st.sums, st.num_rays))
    torch.cuda.synchronize()
    assert eng.found_inf() == 0 and eng.optimizer_steps()[0] == steps
    assert sum(psnr[-5:]) / 5 > sum(psnr[:5]) / 5 + 0.5, psnr
    assert m.color_params.numel() == 64 * 48 + 4096 * (hidden_color - 1) + 1024


@pytest.mark.parametrize("hidden_color", [1, 3])
@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_colour_network_kernels_of_another_depth_against_the_oracle(hidden_color, precision):
    """k_color_fwd / k_mlp_bwd instantiated for 1 and 3 hidden layers: the field's forward and backward (the fused MLP backward) against
    the oracle's depth-generic MLP + autograd on the same weights, to the bounds of the two-layer network's tests (tests/test_gpu_parity:
    test_density_and_color_forward, test_field_backward_against_oracle_autograd); then the colour-alone backward (hrf_color_mlp_bwd)
    against the fused one on the same samples."""
    from oracle import hrf_oracle as O
    from humanrf_amd import ops
    from humanrf_amd.scene_representation import QueryInput
    from tests.util import make_model, oracle_model_from
    frames = tuple(range(15, 27))
    m = make_model(DEV, (6, 6), frames, log2_T=15, emb=2, table_scale=0.5, mlp_precision=precision, n_hidden_color=hidden_color)
    om = oracle_model_from(m, requires_grad=True)
    assert len(om.color_w) == hidden_color + 1
    g = torch.Generator().manual_seed(4)
    n = 1500
    pos = torch.rand(n, 3, generator=g) - 0.5
    fn = torch.tensor(frames, dtype=torch.int32)[torch.randint(0, len(frames), (n,), generator=g)].reshape(-1, 1)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    cams = torch.randint(0, 8, (n, 1), generator=g, dtype=torch.int32)
    w_sig = torch.randn(n, generator=g) * 1e-4
    w_rgb = torch.randn(n, 3, generator=g)
    q = m(QueryInput(is_training=True, positions=pos.to(DEV), directions=d.to(DEV), frame_numbers=fn.to(DEV), camera_numbers=cams.to(DEV)))
    ((q.density * w_sig.to(DEV)).sum() + (q.radiance * w_rgb.to(DEV)).sum()).backward()
    sig, rgb = O.model_forward(om, pos, d, fn, cams, True)
    ((sig * w_sig).sum() + (rgb * w_rgb).sum()).backward()
    assert float((q.radiance.detach().cpu() - rgb.detach()).abs().max()) <= (2e-2 if precision == "bf16" else 4e-3)

    def close(a, b, name, rel_max):
        a, b = a.detach().double().reshape(-1).cpu(), b.detach().double().reshape(-1).cpu()
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        assert rel <= rel_max, (name, rel)
    rel = 6e-2 if precision == "bf16" else 1e-2
    close(m.color_params.grad, torch.cat([w.grad.reshape(-1) for w in om.color_w]), "color_net", rel)
    close(m.sigma_params.grad, torch.cat([w.grad.reshape(-1) for w in om.sigma_w]), "sigma_net", rel)
    close(m.camera_embeddings.weight.grad, om.camera_embeddings.grad, "camera_embeddings", rel)

    # the colour network alone (MODE 2 of the same kernel) on the same samples
    with torch.no_grad():
        xyzt, seg = m._xyzt_seg(pos.to(DEV), fn.to(DEV))
        feats, _ = ops.encode4d_fwd(xyzt, seg, m._tables_h, m.vectors.detach(), m._seg_meta, m.num_segments, save_enc=False)
        sw1, sw2 = m._sigma_w()
        cw1, cw2, cw3 = m._color_w()
        h, _ = ops.density_mlp_fwd(feats, sw1, sw2, float(m.density_scale))
        ray = torch.arange(n, device=DEV)
        emb_w = m.camera_embeddings.weight.detach()
        cam1 = cams.reshape(-1).to(DEV)
        g_flat = torch.zeros(m.color_params.numel(), dtype=torch.float32, device=DEV)
        g_emb = torch.zeros_like(emb_w)
        flags = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.color_mlp_bwd(d.to(DEV), ray, h, emb_w, cam1, 2, True, cw1, cw2, cw3, (w_rgb * 128.0).to(DEV).contiguous(),
                          *m.split_color(g_flat), g_emb, flags)
        torch.cuda.synchronize()
    assert int(flags[0]) == 0
    close(g_flat / 128.0, m.color_params.grad, "colour-alone backward vs fused", 1e-5)
    close(g_emb / 128.0, m.camera_embeddings.weight.grad, "colour-alone embedding gradient vs fused", 1e-5)


def test_half_boundary_overflow_raises_found_inf_on_the_atomic_table_path():
    from humanrf_amd import ops
    from tests.util import make_model
    m = make_model(DEV, (12,), tuple(range(15, 27)), log2_T=15)
    g = torch.Generator(device=DEV).manual_seed(0)
    n = 4096
    xyzt = torch.rand(n, 4, device=DEV, generator=g).contiguous()
    seg = torch.zeros(n, dtype=torch.int32, device=DEV)
    enc = torch.zeros(n, 4, 32, dtype=torch.float16, device=DEV)
    with torch.no_grad():
        m.vectors.fill_(1.0)
    for dy_mag, expect in ((1e3, 0), (1e8, 1)):           # |v * dY| / 128 = 7.8e5 > 65504: inf in the reference's half tensor
        for level_major in (True, False):
            dy = torch.full((16, n, 2) if level_major else (n, 32), dy_mag, device=DEV)
            flags = torch.zeros(1, dtype=torch.int32, device=DEV)
            d_tab = torch.zeros(m.table_params.numel(), device=DEV)
            ops.encode4d_bwd(xyzt, seg, enc, m.vectors.detach(), m._seg_meta, 1, dy, 1.0, d_tab, None, level_major=level_major,
                             grad_boundary=128.0, flags=flags)
            torch.cuda.synchronize()
            assert int(flags) == expect, (dy_mag, level_major)
            assert bool(torch.isfinite(d_tab).all()) == (expect == 0)


def _self_check_worker(port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from humanrf_amd.trainer import TableShardExchange
    ex = TableShardExchange([(0, 1024)], 1, 0)
    ex.self_check(torch.device("cuda:0"))
    out["calls"] = sorted(ex.collectives_used)
    dist.destroy_process_group()


def test_inplace_collective_self_check_runs_on_rccl():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with ctx.Manager() as mgr:
        out = mgr.dict()
        p = ctx.Process(target=_self_check_worker, args=(port, out))
        p.start(); p.join(300)
        assert p.exitcode == 0
        assert any("self_check" in c for c in out["calls"])


def test_host_capture_streams_the_same_pool_as_the_resident_capture():
    """Captures that do not fit HBM (configs[2] at 1x, configs[4] with 1 000 frames) live in pinned host memory; the replacer's
    kernel reads them over the host link. Same schedule, same slots: the HBM pool and its camera tables must come out
    bit-identical to the HBM-resident store's, through the synchronous path and through the background thread."""
    from humanrf_amd.dataset.synthetic import HostCapture, ResidentCapture, SyntheticDataLoader
    from tests.util import small_scene
    scene = small_scene(DEV)
    cams = list(range(len(scene.cameras)))
    pools = {}
    for kind, cls in (("hbm", ResidentCapture), ("host", HostCapture)):
        cap = cls(scene, cams)
        assert cap.images.is_cuda == (kind == "hbm") and (kind == "hbm" or cap.images.is_pinned())
        ld = SyntheticDataLoader(scene, batch_size=256, max_buffer_size=8, max_num_frames_per_batch=3, seed=4, capture=cap)
        iter(ld)
        for _ in range(5):
            ld.replace_next()
        ld.start_replacer(3)
        for _ in range(7):
            ld.tick()
            next(ld)
        ld.stop_replacer()
        torch.cuda.synchronize()
        pools[kind] = (ld.pixel_colors.clone(), ld.frame_numbers_cuda.clone(), ld.camera_numbers_cuda.clone(),
                       ld.inverse_krs_cuda.clone(), ld.replacements)
    assert pools["hbm"][4] == pools["host"][4] > 20
    for a, b in zip(pools["hbm"][:4], pools["host"][:4]):
        assert torch.equal(a, b)
