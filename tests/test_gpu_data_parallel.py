"""Two ranks of the training engine on one GPU (gloo rendezvous on 127.0.0.1): rays are sharded (different loader seeds),
the gradient exchange + fused optimizer must keep the replicas bit-identical, and the exchanged gradient must be the
mean of the per-rank gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, transport, results, synchronous=False, exchange="allreduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.trainer import TrainEngine
    from tests.util import make_model, small_scene
    scene = small_scene("cuda")
    model = make_model("cuda", (6, 6), tuple(scene.frame_numbers), log2_T=15, emb=2)          # same seed on every rank
    if synchronous:   # shared frame schedule, per-rank cameras: only the segments of the pool's frames are exchanged
        loader = SyntheticDataLoader(scene, batch_size=512, max_buffer_size=4, max_num_frames_per_batch=2, seed=10,
                                     camera_seed=50 + rank, frame_synchronous=True)
    else:
        loader = SyntheticDataLoader(scene, batch_size=512, max_buffer_size=8, max_num_frames_per_batch=3, seed=10 + rank)
    iter(loader)
    eng = TrainEngine(model, loader, samples_max_batch_size=30_000, rays_initial_batch_size=512, world_size=world,
                      transport_dtype=transport, exchange=exchange)
    assert (eng.shards is not None) == (exchange == "sharded")
    rays, exchanged = [], []
    for it in range(6 if synchronous else 4):
        st = eng.train_iteration()
        rays.append(st.num_rays)
        r = eng._exchange_ranges()
        exchanged.append(None if r is None else sum(b - a for a, b in r))
        if synchronous:
            for _ in range(3):
                eng.replace_next()      # lockstep replacement: the pools move on to other frames / segments
    torch.cuda.synchronize()
    half_before = model._tables_h.double().sum().cpu()
    if exchange == "sharded":
        # a rank's fp32 masters are current on its own shards only: serialising the model now is an ERROR (never a hidden
        # collective: `if rank == 0: torch.save(model.state_dict())` must not deadlock) ...
        assert not eng.masters_fresh
        for call in (model.state_dict, model.reference_state_dict):
            try:
                call()
                raise AssertionError("stale masters were serialised")
            except RuntimeError as e:
                assert "gather_master_tables" in str(e)
    eng.gather_master_tables()     # ... until every rank has gathered (collective)
    assert eng.masters_fresh
    if rank == 0:
        model.state_dict(); model.reference_state_dict()      # one rank alone may serialise now: no collective is issued
    eng.gather_master_tables()     # (fresh: issues nothing, so this unmatched-looking extra call on every rank is harmless)
    torch.cuda.synchronize()
    # the fp16 tables the kernels read equal the (now complete) masters cast to fp16 on every rank
    assert torch.equal(model._tables_h[:model.table_params.numel()], model.table_params.detach().half())
    results[f"x{rank}"] = (exchanged, eng._big, eng.optimizer_steps())
    from humanrf_amd import ops as _ops
    results[f"mode{rank}"] = (eng.exchange_issue_log_mode, bool(_ops.can_stream_wait_value()), len(eng._exchange_groups(eng._exchange_segments())))
    chk = torch.stack([model.table_params.detach().double().sum(), model.table_params.detach().double().abs().sum(),
                       model.vectors.detach().double().sum(), model.sigma_params.detach().double().sum(),
                       model.color_params.detach().double().sum(), model.camera_embeddings.weight.detach().double().sum(),
                       model._tables_h.double().sum()]).cpu()
    results[rank] = (chk, rays, int(eng.found_inf()))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("transport,exchange", [(torch.bfloat16, "allreduce"), (torch.float32, "allreduce"), (torch.float32, "sharded")])
def test_two_ranks_stay_identical(transport, exchange):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, transport, results, False, exchange)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        (c0, r0, s0), (c1, r1, s1) = results[0], results[1]
        modes = [results["mode0"], results["mode1"]]
    for mode, can_wait, n_groups in modes:
        # two segments, both in the pools: two groups, accumulated by ONE launch that signals each group's completion to the stream its
        # collective waits on (ABI 10) -- the replicas below stayed bit-identical THROUGH that path, not through a fallback
        if can_wait and n_groups > 1:
            assert mode == "one accumulate launch, signalled per group", mode
    assert torch.equal(c0, c1), (c0, c1)                  # replicas identical after 4 steps, to the last bit
    assert r0 != r1                                        # ... although they trained on different rays
    assert s0 == 0 and s1 == 0
    assert float(c0[1]) > 0


@pytest.mark.parametrize("exchange", ["allreduce", "sharded"])
def test_two_ranks_exchange_only_the_pool_segments(exchange):
    """Frame-synchronous pools: the table-gradient exchange covers the segments whose frames are in the pools (a strict
    subset of the tables in some steps), replicas still stay bit-identical, and the per-segment Adam step counts agree --
    with the all-reduce and with reduce-scatter + sharded Adam + all-gather of the fp16 tables."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, torch.float32, results, True, exchange)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        (c0, r0, s0), (c1, r1, s1) = results[0], results[1]
        (x0, big, steps0), (x1, _, steps1) = results["x0"], results["x1"]
    assert torch.equal(c0, c1), (c0, c1)
    assert r0 != r1 and s0 == 0 and s1 == 0
    assert x0 == x1 and all(v is not None for v in x0)
    assert min(x0) < big, "every step exchanged all tables: the touched-segment restriction never engaged"
    assert steps0 == steps1 and steps0[0] == 6 and min(steps0[1:]) < 6


def _validate_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.inference import validate
    from tests.util import make_model, small_scene
    scene = small_scene("cuda")
    model = make_model("cuda", (6, 6), tuple(scene.frame_numbers), log2_T=15, emb=2, table_scale=0.3)
    loader = SyntheticDataLoader(scene, batch_size=512, max_buffer_size=4, max_num_frames_per_batch=2, seed=10, camera_seed=50 + rank)
    iter(loader)
    fr = list(scene.frame_numbers)
    pairs = [(0, fr[0]), (1, fr[3]), (2, fr[5]), (0, fr[7]), (3, fr[1])]
    sharded = validate(model, loader, pairs, rays_batch_size=2048, world_size=world, rank=rank)
    alone = validate(model, loader, pairs, rays_batch_size=2048) if rank == 0 else None
    results[rank] = (sharded["psnr"], sharded["images_rendered_here"], None if alone is None else alone["psnr"])
    dist.destroy_process_group()


def test_validation_is_sharded_over_the_ranks_and_gathers_every_image():
    """SURVEY.md 8(e), last bullet (the reference validates on its one GPU, trainer.py:257-370): with N ranks inference.validate deals
    the images out round-robin, and every rank returns the full per-image PSNR list -- equal to what one rank alone computes."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_validate_worker, args=(r, 2, port, results)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        (p0, n0, alone), (p1, n1, _) = results[0], results[1]
    assert n0 == 3 and n1 == 2
    assert p0 == p1 and len(p0) == 5
    assert all(abs(a - b) < 1e-9 for a, b in zip(p0, alone)) and all(5.0 < v < 80.0 for v in p0)
