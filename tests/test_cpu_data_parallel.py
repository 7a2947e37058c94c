"""World-size-2 gloo test of the data-parallel gradient exchange; runs on CPU."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, transport, results):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanrf_amd.trainer import allreduce_gradients
    big = 1000
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(big + 37, generator=g)
    flat[-1] = float(rank == 1)           # found_inf flag raised on rank 1 only
    mine = flat.clone()
    if transport == "split":   # the engine's form: head without waiting, then the tail, sum only
        wire = torch.empty(big, dtype=torch.bfloat16)
        pending = allreduce_gradients(flat, big, world, None, torch.bfloat16, wire=wire, average=False, tail=False, wait=False)
        allreduce_gradients(flat, big, world, None, torch.bfloat16, wire=wire, average=False, head=False)
        pending()
        flat /= world
    else:
        allreduce_gradients(flat, big, world, None, transport)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    results[rank] = (flat, torch.stack(gathered).mean(0))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(transport):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, transport, results)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        return {k: v for k, v in results.items()}


def test_allreduce_gradients_fp32_is_exact_mean():
    res = _run(torch.float32)
    for rank in (0, 1):
        got, want = res[rank]
        assert torch.allclose(got, want, atol=1e-6)
    assert torch.equal(res[0][0], res[1][0])          # replicas stay identical
    assert float(res[0][0][-1]) > 0                    # found_inf propagates to every rank


def test_allreduce_gradients_bf16_transport():
    res = _run(torch.bfloat16)
    for rank in (0, 1):
        got, want = res[rank]
        assert torch.allclose(got[:1000], want[:1000], rtol=2e-2, atol=2e-2)   # bf16 wire format: 8-bit mantissa
        assert torch.allclose(got[1000:], want[1000:], atol=1e-6)               # small tail travels in fp32
    assert torch.equal(res[0][0], res[1][0])


def test_split_exchange_equals_one_shot():
    res = _run("split")
    for rank in (0, 1):
        got, want = res[rank]
        assert torch.allclose(got[:1000], want[:1000], rtol=2e-2, atol=2e-2)
        assert torch.allclose(got[1000:], want[1000:], atol=1e-6)
    assert torch.equal(res[0][0], res[1][0])
    assert float(res[0][0][-1]) > 0


def test_single_rank_is_a_noop():
    from humanrf_amd.trainer import allreduce_gradients
    x = torch.arange(10.0)
    allreduce_gradients(x, 5, 1)
    assert torch.equal(x, torch.arange(10.0))
