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


# ------------------------------------------------------------------------------------------------
# Sharded exchange: reduce-scatter -> Adam on the owned shard -> all-gather of the 16-bit tables.
# Four gloo ranks on the CPU; the optimizer step is torch's own formula on the shard (the fused HIP optimizer needs a GPU).
# ------------------------------------------------------------------------------------------------
def _adam_first_step(p, g, lr=1e-2, b1=0.9, b2=0.99, eps=1e-15):
    m = (1 - b1) * g
    v = (1 - b2) * g * g
    return p - lr * (m / (1 - b1)) / ((v / (1 - b2)).sqrt() + eps)


def _shard_worker(rank, world, port, results):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from humanrf_amd.trainer import TableShardExchange
    ranges = [(0, 64), (64, 64 + 160), (224, 224 + 96)]          # three "segments" of table values (multiples of 8)
    total = ranges[-1][1]
    ex = TableShardExchange(ranges, world, rank)
    assert [b - a for a, b in ex.own_ranges] == [(b - a) // world for a, b in ranges]
    params = torch.linspace(-1.0, 1.0, total)                     # identical replicas
    p16 = params.half()
    grads = torch.randn(total, generator=torch.Generator().manual_seed(7 + rank))
    local = grads.clone()
    touched = [0, 2]                                              # segment 1 has no gradient anywhere this step
    ex.reduce_scatter(grads, touched)()
    for s in touched:                                            # Adam on the owned shard only, with the MEAN gradient
        oa, ob = ex.own_ranges[s]
        params[oa:ob] = _adam_first_step(params[oa:ob], grads[oa:ob] / world)
        p16[oa:ob] = params[oa:ob].half()
    assert ex.tensor_collectives is False                         # gloo: the stand-ins, chosen by the backend (no try / except)
    finish = ex.all_gather(p16, touched, wait=False)             # the engine's form: issued behind the optimizer, waited for by
    assert callable(finish)                                      # the next reader of the tables (HumanRF._refresh_half)
    finish()
    assert ex.all_gather(params, touched) is None                # what gather_master_tables() does before a checkpoint
    assert ex.collectives_used == {"all_reduce (gloo stand-in for reduce_scatter_tensor)",
                                   "all_gather (gloo stand-in for all_gather_into_tensor)"}
    everyone = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(everyone, local)
    results[rank] = (params, p16, torch.stack(everyone).sum(0) / world)
    dist.destroy_process_group()


def test_sharded_exchange_four_ranks_equals_replicated_adam():
    world = 4
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_shard_worker, args=(r, world, port, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = {k: v for k, v in results.items()}
    p0 = torch.linspace(-1.0, 1.0, 320)
    mean_g = res[0][2]
    want = p0.clone()
    for a, b in ((0, 64), (224, 320)):
        want[a:b] = _adam_first_step(p0[a:b], mean_g[a:b])
    for r in range(world):
        params, p16, _ = res[r]
        assert torch.allclose(params, want, atol=1e-6), r        # every rank ends with the replicated optimizer's result
        assert torch.equal(p16, want.half()) or torch.allclose(p16.float(), want, atol=1e-3)
        assert torch.equal(params[64:224], p0[64:224])           # the untouched segment did not move
        assert torch.equal(params, res[0][0]) and torch.equal(p16, res[0][1])   # replicas identical


def test_shard_ranges_cover_every_segment_once():
    from humanrf_amd.trainer import TableShardExchange
    ranges = [(0, 8 * 524288), (8 * 524288, 8 * 524288 + 8 * 1019904)]
    for world in (1, 2, 4, 8):
        owned = [TableShardExchange(ranges, world, r).own_ranges for r in range(world)]
        for s, (a, b) in enumerate(ranges):
            pieces = sorted(o[s] for o in owned)
            assert pieces[0][0] == a and pieces[-1][1] == b
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
            assert len({pb - pa for pa, pb in pieces}) == 1
    import pytest
    with pytest.raises(ValueError):
        TableShardExchange([(0, 100)], 8, 0)
