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


def _pair(n_levels, geo, emb, log2_T=15):
    from oracle import ref_harness
    from humanrf_amd.scene_representation import HumanRF
    if not ref_harness.available():
        pytest.skip("the reference's sources (or their snapshot under oracle/_ref/reference) are not on this machine")
    ref = ref_harness.load("cpu")
    frames, sizes = tuple(range(15, 27)), (6, 6)
    kw = dict(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2, log2_hashmap_size=log2_T, n_levels=n_levels,
              coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=geo, n_neurons=64, n_hidden_layers_density=1,
              n_hidden_layers_color=2, sh_degree=4, segment_sizes=sizes, camera_embedding_dim=emb)
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


@pytest.mark.parametrize("n_levels,geo,emb", [(8, 15, 2), (12, 7, 2), (16, 4, 0), (5, 0, 3), (16, 15, 2)])
def test_model_knobs_against_the_reference_class(n_levels, geo, emb):
    from humanrf_amd.scene_representation.query_io import QueryInput
    ref, m, rm, frames = _pair(n_levels, geo, emb)
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
