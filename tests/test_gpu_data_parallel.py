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


def _worker(rank, world, port, transport, results):
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
    loader = SyntheticDataLoader(scene, batch_size=512, max_buffer_size=8, max_num_frames_per_batch=3, seed=10 + rank)
    iter(loader)
    eng = TrainEngine(model, loader, samples_max_batch_size=30_000, rays_initial_batch_size=512, world_size=world,
                      transport_dtype=transport)
    rays = []
    for _ in range(4):
        st = eng.train_iteration()
        rays.append(st.num_rays)
    torch.cuda.synchronize()
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


@pytest.mark.parametrize("transport", [torch.bfloat16, torch.float32])
def test_two_ranks_stay_identical(transport):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, transport, results)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        (c0, r0, s0), (c1, r1, s1) = results[0], results[1]
    assert torch.equal(c0, c1), (c0, c1)                  # replicas identical after 4 steps, to the last bit
    assert r0 != r1                                        # ... although they trained on different rays
    assert s0 == 0 and s1 == 0
    assert float(c0[1]) > 0
