"""Host side of the training-engine options added in round 2: the bf16 MLP variant (the oracle's bf16 mode is what its
docstring says, the weight dtype selects the kernels' arithmetic, the Adam descriptors carry the bf16 flag), the record layout
of the device-side GradScaler, and how a batch is cut into pipelined pieces. No GPU, no compute through the library."""
import ctypes
import pytest
import torch

from oracle import hrf_oracle as O


def test_oracle_bf16_mode_rounds_to_bf16_everywhere():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(257, 32, generator=g).half().float()
    w = [torch.randn(64, 32, generator=g).mul(0.2).bfloat16().float(), torch.randn(16, 64, generator=g).mul(0.2).bfloat16().float()]
    y16 = O.mlp(x, w, "None", "fp16")
    yb = O.mlp(x, w, "None", "bf16")
    assert torch.equal(yb, yb.bfloat16().float())          # bf16 values ...
    assert torch.equal(yb, yb.half().float())              # ... that the half output tensor holds exactly
    assert not torch.equal(yb, y16)
    assert float((yb - y16).abs().max()) < 0.05 * float(y16.abs().max())
    # restated by hand: rounding after the input, the hidden layer and the output
    h = (x.bfloat16().float() @ w[0].t()).relu().bfloat16().float()
    assert torch.equal(yb, (h @ w[1].t()).bfloat16().float().half().float())
    # default precision is the reference's
    assert torch.equal(O.mlp(x, w, "None"), y16)


def test_weight_dtype_selects_the_mode_and_descriptor_flag():
    from humanrf_amd import ops, _lib
    a16, ab = torch.zeros(8, dtype=torch.float16), torch.zeros(8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="float16 or bfloat16"):
        ops._mlp_mode(torch.zeros(8))
    with pytest.raises(RuntimeError, match="share one 16-bit type"):
        ops._mlp_mode(a16, ab)
    with pytest.raises(RuntimeError, match="expected device"):     # dtype accepted, CPU tensor refused by the device check
        ops._mlp_mode(ab)
    f = [torch.zeros(8) for _ in range(4)]
    raw = ops.adam_descriptors([(f[0], f[1], f[2], f[3], a16, 0), (f[0], f[1], f[2], f[3], ab, 1),
                                (f[0], f[1], f[2], f[3], None, 2)], "cpu")
    recs = (_lib.AdamTensor * 3).from_buffer_copy(bytes(raw.numpy()))
    assert [r.reserved for r in recs] == [0, 1, 0] and [r.group for r in recs] == [0, 1, 2]
    assert ctypes.sizeof(_lib.AdamTensor) == 56
    with pytest.raises(RuntimeError):
        ops.adam_descriptors([(f[0], f[1], f[2], f[3], torch.zeros(8), 0)], "cpu")


def test_grad_scaler_record_layout():
    """hrf_grad_scaler (include/hrf.h) <-> the ctypes mirror: 32 bytes, torch.amp.GradScaler's defaults."""
    from humanrf_amd import ops, _lib
    assert ctypes.sizeof(_lib.GradScaler) == 32
    st = ops.grad_scaler_state(ops.grad_scaler("cpu"))
    assert st == {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "growth_tracker": 0}
    st = ops.grad_scaler_state(ops.grad_scaler("cpu", init_scale=128.0, growth_interval=100_000))
    assert st["scale"] == 128.0 and st["growth_interval"] == 100_000


def test_model_precision_option():
    from humanrf_amd.scene_representation import HumanRF
    kw = dict(density_scale=100, sorted_frame_numbers=tuple(range(4)), n_features_per_level=2, log2_hashmap_size=12,
              n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
              n_hidden_layers_density=1, n_hidden_layers_color=2, sh_degree=4, segment_sizes=(4,), camera_embedding_dim=0,
              device="cpu")
    m = HumanRF(**kw, mlp_precision="bf16")
    assert m._sigma_h.dtype == torch.bfloat16 and m._tables_h.dtype == torch.float16
    m._refresh_half()
    assert torch.equal(m._sigma_h, m.sigma_params.detach().bfloat16())
    assert HumanRF(**kw)._sigma_h.dtype == torch.float16
    with pytest.raises(ValueError):
        HumanRF(**kw, mlp_precision="fp8")


def test_pipeline_pieces_are_ray_aligned_partitions():
    """TrainEngine._pieces (host logic): the cut points the collector hands over become 1 / 2 / 4 contiguous pieces that cover
    the batch; anything degenerate falls back to one piece."""
    from types import SimpleNamespace
    from humanrf_amd.trainer import TrainEngine
    eng = TrainEngine.__new__(TrainEngine)
    eng.world_size, eng.pipeline_min_samples = 1, 1000
    ib = SimpleNamespace(num_rays=400, num_samples=8000, _cuts=[(100, 2100), (200, 4000), (300, 6500)])
    eng.pipeline_pieces = 1
    assert eng._pieces(ib) == [(0, 400, 0, 8000)]
    eng.pipeline_pieces = 2
    assert eng._pieces(ib) == [(0, 200, 0, 4000), (200, 400, 4000, 8000)]
    eng.pipeline_pieces = 4
    p = eng._pieces(ib)
    assert p == [(0, 100, 0, 2100), (100, 200, 2100, 4000), (200, 300, 4000, 6500), (300, 400, 6500, 8000)]
    assert all(a[1] == b[0] and a[3] == b[2] for a, b in zip(p, p[1:]))
    ib._cuts = None
    assert eng._pieces(ib) == [(0, 400, 0, 8000)]                       # batch assembled from several chunks
    ib._cuts = [(100, 2100), (100, 2100), (300, 6500)]
    assert eng._pieces(ib) == [(0, 400, 0, 8000)]                       # an empty piece: one pass
    ib._cuts = [(100, 2100), (200, 4000), (300, 6500)]
    eng.world_size = 2
    assert eng._pieces(ib) == [(0, 400, 0, 8000)]                       # data parallel: one pass (the exchange is interleaved)
    eng.world_size, eng.pipeline_min_samples = 1, 10_000
    assert eng._pieces(ib) == [(0, 400, 0, 8000)]                       # too small to fill the chip in pieces


def test_scatter_workspace_size_follows_the_documented_layout():
    """hrf_scatter_workspace_bytes is host arithmetic (no device call): header tables + per-tile tables + counters + maxima +
    record queues + (round 6) one maximum per (segment, level, encoding) of csrc/scatter.hip's layout (tiles of 1024 samples, one spare
    tile per segment, counters for up to 64 queues per (level, encoding), 16 levels x 4 encodings x
    8192 records of 12 bytes per tile), 256-byte aligned pieces; 0 for a degenerate request."""
    from humanrf_amd import _lib, ops
    lib = _lib.lib()
    al = lambda x: (x + 255) // 256 * 256

    def want(n, segs):
        tiles = (n + 1023) // 1024 + segs
        return (2 * al((segs + 1) * 4) + 3 * al(tiles * 4) + al(16 * 4 * 64 * tiles * 4) + al(16 * 4 * tiles * 4)
                + al(tiles * 16 * 4 * 8192 * 12) + al(segs * 16 * 4 * 4))
    for n, segs in ((1, 1), (1024, 1), (1025, 7), (704_000, 7), (704_000, 142), (2_000_000, 1024)):
        assert int(lib.hrf_scatter_workspace_bytes(n, segs)) == want(n, segs), (n, segs)
    assert int(lib.hrf_scatter_workspace_bytes(0, 3)) == 0 and int(lib.hrf_scatter_workspace_bytes(100, 0)) == 0
    assert ops.ScatterWorkspace.supports(65536, 7) and ops.ScatterWorkspace.supports(1, 1024)
    assert ops.ScatterWorkspace.supports(1 << 19, 1)      # 64 chunks: log2_hashmap_size 19 on a 100-frame segment
    assert not ops.ScatterWorkspace.supports((1 << 19) + 1, 7) and not ops.ScatterWorkspace.supports(4096, 1025)
    assert not ops.ScatterWorkspace.supports(0, 1)


def test_committed_traffic_summary_belongs_to_the_committed_kernel_sources():
    """bench.py reports roofline.traffic only from a profiles/*_traffic.json whose fingerprint equals the SHA-256 of the kernel
    sources in the tree; the newest committed summary must be that one (re-run tools/measure.sh pmc after touching csrc/)."""
    import json
    import os
    import bench
    prof = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "profiles")
    names = sorted(f for f in os.listdir(prof) if f.endswith("_traffic.json"))
    assert names, "no PMC traffic summary committed"
    newest = json.load(open(os.path.join(prof, names[-1])))
    assert len(bench.kernel_source_fingerprint()) == 64
    if newest.get("kernel_sources_sha256") != bench.kernel_source_fingerprint():
        pytest.skip("kernel sources changed since the last PMC pass: bench.py will report roofline.traffic = null until "
                    "tools/measure.sh pmc is re-run")
    for k in ("k_prune_march", "k_encode4d_fwd", "table_scatter"):
        assert newest[k]["fetch_bytes_per_encoded_sample"] > 0


def test_model_hooks_of_the_sharded_exchange_are_called_where_the_tables_are_read_or_serialised():
    """A data-parallel TrainEngine with the sharded exchange installs two hooks on the model: `_tables_ready` (wait for the
    in-flight all-gather of the fp16 tables) must run before anything reads the tables -- every reader goes through
    _refresh_half -- and `_master_sync` (gather the other ranks' shards of the fp32 masters) before anything serialises the
    model: state_dict() and reference_state_dict() (a checkpoint taken through the model used to store stale shards)."""
    from humanrf_amd.scene_representation import HumanRF
    m = HumanRF(density_scale=100, sorted_frame_numbers=tuple(range(4)), n_features_per_level=2, log2_hashmap_size=8,
                n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
                n_hidden_layers_density=1, n_hidden_layers_color=2, sh_degree=4, segment_sizes=(4,), camera_embedding_dim=0,
                device="cpu")
    calls = []
    m._tables_ready = lambda: calls.append("ready")
    m._master_sync = lambda: calls.append("sync")
    m._refresh_half()
    assert calls == ["ready"]
    sd = m.state_dict()
    ref = m.reference_state_dict()
    assert calls == ["ready", "sync", "sync"]
    assert "table_params" in sd and "sigma_net.params" in ref
    assert not any(k.startswith("_tables_ready") or k.startswith("_master_sync") for k in sd)   # plain attributes, not state


def test_engine_options_are_validated_and_pieces_switch_the_frame_ordering_off():
    from humanrf_amd.trainer import TrainEngine
    eng = TrainEngine.__new__(TrainEngine)
    eng.world_size = 1

    class _Collector:
        sort_batch = True
    eng.collector = _Collector()
    with pytest.warns(UserWarning, match="binned"):                            # (not silently)
        eng.pipeline_pieces = 2
    assert eng.pipeline_pieces == 2 and eng.collector.sort_batch is False      # pieces need draw-order cut points
    eng.pipeline_pieces = 1
    assert eng.collector.sort_batch is True
    assert eng._dp is False
    eng.force_collectives = True
    assert eng._dp is True                                                     # a one-rank group still runs the exchange
    eng.force_collectives = False
    eng.world_size = 2
    assert eng._dp is True


def test_chunk_maps_of_the_binned_scatter_are_bijections_with_whole_lines_per_chunk():
    """csrc/scatter.hip deals the entries of a level table out to 8192-entry chunks: contiguously up to 8 chunks, and above
    that (tables of 65 537 .. 524 288 entries: 2^17 - 2^19 hashed levels and the dense levels res 43 / 55 / 73) by runs of 16
    entries -- one 128-byte line of d_tables -- round-robin. Restated here from the header comment: (queue, local) must be a
    bijection of [0, chunks * 8192), every entry of the table must land below 8192 in its chunk, and a run of 16 entries
    must stay in one chunk (the write-back of a chunk is whole lines)."""
    import numpy as np
    CH = 13

    def qshift(size):
        chunks, s = (size + (1 << CH) - 1) >> CH, 0
        while (1 << s) < chunks:
            s += 1
        return s

    def maps(key, sh):
        if sh <= 3:
            return key >> CH, key & ((1 << CH) - 1)
        return (key >> 4) & ((1 << sh) - 1), (((key >> 4) >> sh) << 4) | (key & 15)

    def entry(q, local, sh):
        return ((q << CH) | local) if sh <= 3 else ((((local >> 4) << sh) | q) << 4) | (local & 15)

    for size in (1, 8192, 8193, 35944, 65536, 65537, 79512, 131072, 166375, 262144, 389017, 524288):
        sh = qshift(size)
        assert sh <= 6
        key = np.arange(size, dtype=np.int64)
        q, loc = maps(key, sh)
        assert int(q.max()) < (1 << sh) and int(loc.max()) < (1 << CH), size
        assert np.array_equal(entry(q, loc, sh), key), size                          # inverse map
        assert len(set(zip(q.tolist(), loc.tolist()))) == size                      # no two entries share an accumulator
        assert np.array_equal(q[: size // 16 * 16].reshape(-1, 16).min(1), q[: size // 16 * 16].reshape(-1, 16).max(1))
        if sh > 3:                                                                   # interleaved: every queue gets its share
            counts = np.bincount(q, minlength=1 << sh)
            assert counts.max() - counts.min() <= 16, (size, counts.min(), counts.max())


def test_bench_main_refers_to_no_undefined_name():
    """bench.py's main() only runs on a GPU box; a name that is used there without ever being bound (a block lost in an edit)
    must fail HERE, not on the driver's lease: every name main() loads is bound in main(), at module level, or a builtin."""
    import ast
    import builtins
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    module_names = set()
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            module_names.add(n.name)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            module_names.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.Assign):
            module_names.update(x.id for t in n.targets for x in ast.walk(t) if isinstance(x, ast.Name))
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    bound, loaded = set(), []
    for x in ast.walk(main):
        if isinstance(x, ast.Name):
            (bound.add(x.id) if isinstance(x.ctx, ast.Store) else loaded.append((x.id, x.lineno)))
        elif isinstance(x, ast.FunctionDef):
            bound.add(x.name)
            bound.update(a.arg for a in x.args.args + x.args.kwonlyargs)
        elif isinstance(x, ast.Lambda):
            bound.update(a.arg for a in x.args.args)
        elif isinstance(x, (ast.Import, ast.ImportFrom)):
            bound.update((a.asname or a.name).split(".")[0] for a in x.names)
        elif isinstance(x, ast.ExceptHandler) and x.name:
            bound.add(x.name)
    undefined = sorted({(n, l) for n, l in loaded if n not in bound and n not in module_names and not hasattr(builtins, n)})
    assert not undefined, undefined
    for needed in ("n_trials", "chosen", "reduce_stat", "build_engine"):
        assert needed in bound or needed in module_names, needed


def test_step_collector_grows_its_packed_sample_buffers_and_keeps_what_is_written():
    """StepCollector._reserve_samples: an iteration of the batch-growing loop may yield more than the 2.1 budgets the buffers start
    with (trainer.py:143-163 takes whatever next(loader) + prune_samples give; merge_input_batches cuts afterwards): the buffers grow,
    the samples already packed stay."""
    from humanrf_amd.fast_path import StepCollector
    c = StepCollector.__new__(StepCollector)
    c.dev = torch.device("cpu")
    c.cap_samples = 100
    c.t = torch.arange(100, dtype=torch.float32)
    c.ray = torch.arange(100, dtype=torch.int64) // 7
    c.t_alt, c.ray_alt = torch.empty_like(c.t), torch.empty_like(c.ray)
    c._reserve_samples(90, 40)                      # fits: nothing happens
    assert c.cap_samples == 100 and c.t.numel() == 100 and c.t_alt is not None
    c._reserve_samples(260, 40)
    assert c.cap_samples >= 260 and c.t.numel() == c.ray.numel() == c.cap_samples
    assert torch.equal(c.t[:40], torch.arange(40, dtype=torch.float32)) and torch.equal(c.ray[:40], torch.arange(40) // 7)
    assert c.t_alt is None and c.ray_alt is None    # the re-sort pair is reallocated at the new size when it is needed
    cap = c.cap_samples
    c._reserve_samples(cap + 1, 0)                  # geometric growth: one sample more buys half as much again
    assert c.cap_samples >= int(cap * 1.5)
    grown = c.t
    c._reserve_samples(cap + 2, 0)
    assert c.t is grown                             # ... so the next request is served without a reallocation


class _FakeDist:
    """torch.distributed for a group of one rank, with switches that break the in-place collectives."""

    class ReduceOp:
        SUM, MAX = "sum", "max"

    def __init__(self, break_rs=False, break_ag=False):
        self.break_rs, self.break_ag, self.calls = break_rs, break_ag, []

    def all_reduce(self, t, op=None, group=None):
        self.calls.append("all_reduce")

    def reduce_scatter_tensor(self, out, inp, op=None, group=None):
        self.calls.append("reduce_scatter_tensor")
        if self.break_rs:
            out.add_(1.0)

    def all_gather_into_tensor(self, out, inp, group=None):
        self.calls.append("all_gather_into_tensor")
        if self.break_ag:
            out[-1] += 1.0


def test_collective_self_check_runs_both_probes_then_raises_by_consensus():
    """TableShardExchange.self_check: a rank never raises between the probes' collectives (the others would wait in the next one): both
    probes run, the verdicts are max-reduced, then every rank raises or none does."""
    from humanrf_amd.trainer import TableShardExchange
    ex = TableShardExchange([(0, 64)], world_size=1, rank=0)
    good = _FakeDist()
    ex.self_check(torch.device("cpu"), numel=1024, _dist=good)
    assert good.calls == ["all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "all_reduce"]
    assert any("self_check" in c for c in ex.collectives_used)
    for fake, word in ((_FakeDist(break_rs=True), "reduce_scatter_tensor"), (_FakeDist(break_ag=True), "all_gather_into_tensor")):
        with pytest.raises(RuntimeError, match=word):
            ex.self_check(torch.device("cpu"), numel=1024, _dist=fake)
        assert fake.calls == ["all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "all_reduce"]   # all four, then the raise


def test_exchange_groups_are_consecutive_cover_everything_and_balance_bytes():
    """TrainEngine._exchange_groups (host logic of the pipelined data-parallel exchange): the segments whose tables are exchanged are
    cut into at most `exchange_groups` groups of CONSECUTIVE ids (hrf_scatter_accumulate takes an id range), every segment exactly
    once and in order, bytes balanced; a gap in the ids always starts a new group."""
    from humanrf_amd.trainer import TrainEngine
    eng = TrainEngine.__new__(TrainEngine)
    sizes = [8, 16, 16, 8, 16, 8, 16, 4, 4, 4]
    eng._table_ranges, off = [], 0
    for n in sizes:
        eng._table_ranges.append((off, off + n))
        off += n
    for groups_wanted in (1, 2, 3, 4, 7, 16):
        eng.exchange_groups = groups_wanted
        for segs in ([0, 1, 2, 3, 4, 5, 6], [2], [1, 2, 5], [0, 2, 4, 6], list(range(10)), [3, 4, 5, 7, 8, 9]):
            groups = eng._exchange_groups(segs)
            assert [s for g in groups for s in g] == segs
            assert all(g == list(range(g[0], g[-1] + 1)) for g in groups)
            runs = 1 + sum(1 for a, b in zip(segs, segs[1:]) if b != a + 1)
            assert len(groups) <= max(groups_wanted, runs)
            if groups_wanted > 1 and segs == [0, 1, 2, 3, 4, 5, 6]:
                tot = [sum(sizes[i] for i in g) for g in groups]
                assert len(groups) == min(groups_wanted, 7) or max(tot) <= 2 * (sum(tot) / len(tot))
    eng.exchange_groups = 1
    assert eng._exchange_groups([0, 1, 2]) == [[0, 1, 2]]


def test_model_with_engine_hooks_can_be_deep_copied_and_pickled():
    """an attached TrainEngine leaves weakref.WeakMethod hooks on the model; copy.deepcopy (EMA copies) and torch.save of
    the whole module must work and give a model WITHOUT the hooks (and with its level metadata intact)."""
    import copy
    import io
    import weakref
    from tests.util import make_model
    m = make_model("cpu", (6, 6), tuple(range(15, 27)), log2_T=12, emb=2)

    class Eng:
        def hook(self):
            pass
    e = Eng()
    m._master_sync = weakref.WeakMethod(e.hook)
    m._tables_ready = weakref.WeakMethod(e.hook)
    c = copy.deepcopy(m)
    assert c._master_sync is None and c._tables_ready is None and m._master_sync is not None
    assert torch.equal(c.table_params, m.table_params) and c.table_params is not m.table_params
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    l = torch.load(buf, weights_only=False)
    assert l._master_sync is None and torch.equal(l.table_params, m.table_params)
    for x in (c, l):
        assert bytes(x._metas_host) == bytes(m._metas_host) and int(x._metas_host[1].levels[3].size) == int(m._metas_host[1].levels[3].size)
