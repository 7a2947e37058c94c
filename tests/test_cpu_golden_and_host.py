"""CPU-side tests (run with -m "not gpu"): the oracle against the committed golden vectors, the host logic
(level tables, frame lookups, merge_input_batches, adaptive partitioning) and the C ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import hrf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "hotpath_seed123.npz")


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(GOLD))


def test_oracle_reproduces_golden_vectors(golden):
    """Re-run the generator's computation and compare with the frozen file: bit-exact for the sampler and the
    visibility mask, tight tolerances elsewhere (CPU BLAS summation order may differ between hosts)."""
    from tests.golden.make_golden import compute
    inp = {k[3:]: v for k, v in golden.items() if k.startswith("in_")}
    out = compute(inp)
    assert np.allclose(out["param_checksum"], golden["param_checksum"], rtol=1e-12), "RNG stream changed"
    for k in ("smp_origins", "smp_dirs", "smp_rgba_s", "smp_frames_s", "smp_cams_s", "smp_minmax", "smp_ray_mask",
              "smp_t", "smp_ray"):
        assert np.array_equal(out[k], golden[k]), k
    flips = int((out["prune_vis"] != golden["prune_vis"]).sum())
    assert flips <= 2, flips
    assert np.allclose(out["prune_sigma"], golden["prune_sigma"], rtol=2e-2, atol=1e-3)
    if flips == 0:
        assert np.allclose(out["color"], golden["color"], atol=1e-4)
        assert np.allclose(out["loss"], golden["loss"], rtol=1e-4)
        for k in ("grad_sigma_w", "grad_color_w", "grad_tables_norm"):
            a, b = out[k].astype(np.float64), golden[k].astype(np.float64)
            assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(b) + 1e-12, k


def test_golden_internal_consistency(golden):
    g = golden
    assert g["smp_ray_mask"].sum() == g["smp_origins"].shape[0] > 10
    assert np.all(np.diff(g["smp_ray"]) >= 0) and g["smp_t"].shape == g["smp_ray"].shape
    assert g["prune_vis"].sum() == g["sigma"].shape[0] == g["rgb"].shape[0]
    assert g["color"].shape == (g["smp_origins"].shape[0], 3) and np.isfinite(g["color"]).all()
    acc = g["acc"].reshape(-1)
    assert acc.min() >= 0 and acc.max() <= 1.0 + 1e-5


def test_level_table_matches_oracle_restatement():
    """Product host code (humanrf_amd.scene_representation.hashgrid) vs the oracle's independent restatement."""
    from humanrf_amd.scene_representation import hashgrid
    pls = hashgrid.per_level_scale(32, 2048, 16)
    for size, log2 in ((100, 19), (50, 19), (25, 19), (12, 19), (6, 19), (6, 14)):
        l2 = hashgrid.segment_log2_hashmap_size(size, log2)
        mine = hashgrid.level_table(16, l2, 32, pls)
        ref = O.hashgrid_levels(16, l2, 32, pls)
        assert [(m[1], m[2], m[3], bool(m[4])) for m in mine] == [(r.res, r.size, r.offset, r.hashed) for r in ref]
        assert all(abs(m[0] - r.scale) == 0 for m, r in zip(mine, ref))
    assert [hashgrid.segment_log2_hashmap_size(s, 19) for s in (100, 50, 25, 12, 6)] == [19, 18, 17, 16, 15]
    t = hashgrid.level_table(16, 19, 32, pls)
    assert t[0][:3] == (31.0, 32, 32768) and not t[3][4] and t[4][4] and t[-1][2] == 1 << 19
    metas, entries, total = hashgrid.build_segment_meta([6, 12], 16, 19, 32, 2048)
    assert total == 4 * sum(entries) and metas[1].table_offset == 4 * entries[0]


def test_frame_tables_follow_reference_semantics():
    from humanrf_amd.scene_representation import hashgrid
    frames = tuple(range(15, 40))
    f2s, f2l = hashgrid.frame_tables(frames, (12, 12, 6))   # last segment is cut at the number of frames
    assert f2s[14] == -1 and f2s[15] == 0 and f2s[26] == 0 and f2s[27] == 1 and f2s[39] == 2
    assert f2l[15] == 0.0 and abs(f2l[26] - 11 / 12) < 1e-7 and f2l[39] == 0.0  # 1 frame in the last segment


def test_merge_input_batches_semantics():
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.input import merge_input_batches

    def batch(n_rays, lens, base):
        ray = torch.repeat_interleave(torch.arange(n_rays), torch.tensor(lens))
        return InputBatch(ray_origins=torch.full((n_rays, 3), float(base)), ray_directions=torch.zeros(n_rays, 3),
                          minmaxes=torch.zeros(n_rays, 2), rgba=torch.zeros(n_rays, 4),
                          ray_masks=torch.ones(n_rays + 2, 1, dtype=torch.bool),
                          frame_numbers=torch.full((n_rays, 1), base, dtype=torch.int32),
                          unique_frame_numbers=torch.tensor([[base]], dtype=torch.int32),
                          camera_numbers=torch.zeros(n_rays, 1, dtype=torch.int32),
                          sample_distances=torch.arange(ray.numel(), dtype=torch.float32).view(-1, 1) + 100 * base,
                          ray_indices=ray, width=4, height=3)
    a, b = batch(3, [2, 0, 3], 1), batch(2, [4, 1], 2)
    m = merge_input_batches([a, b])
    assert m.num_rays == 5 and m.num_samples == 10 and m.ray_indices.tolist() == [0, 0, 2, 2, 2, 3, 3, 3, 3, 4]
    assert sorted(m.unique_frame_numbers.reshape(-1).tolist()) == [1, 2] and m.width == 4
    cut = merge_input_batches([a, b], max_num_samples=6)   # ray_indices[6] == 3 -> keep rays 0..2 only
    assert cut.num_rays == 3 and cut.num_samples == 5 and cut.ray_indices.tolist() == [0, 0, 2, 2, 2]
    assert cut.unique_frame_numbers.reshape(-1).tolist() == [1]
    assert torch.equal(cut.sample_distances, m.sample_distances[:5])


def test_adaptive_partitioning_matches_reference_algorithm():
    from humanrf_amd.adaptive_temporal_partitioning import compute_adaptive_segment_sizes
    G = 8

    def grid_of(volume):
        g = torch.zeros(G * G * G, dtype=torch.uint8)
        g[:volume] = 255
        return g.view(G, G, G)
    # static scene: the cluster only closes at the maximum segment size; 30 frames end up in one final segment
    sizes = compute_adaptive_segment_sizes(lambda f: grid_of(100), list(range(30)), 1.25)
    assert sum(sizes) >= 30 and all(s in (6, 12, 25, 50, 100) for s in sizes)
    # growing occupancy: expansion factor passes 1.25 quickly -> minimum size segments
    sizes = compute_adaptive_segment_sizes(lambda f: grid_of(100 + 10 * f), list(range(24)), 1.25)
    assert sizes[0] == 6 and sum(sizes) >= 24


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from humanrf_amd import _lib
    lib = _lib.lib()
    assert lib.hrf_abi_version() == 10
    header = open(os.path.join(ROOT, "include", "hrf.h")).read()
    declared = set(re.findall(r"\b(hrf_[a-z0-9_]+)\s*\(", header))
    declared -= {"hrf_stream_t"}
    assert len(declared) >= 24
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/hrf.h but not exported by libhrf_hip.so"
    assert set(_lib.exported_symbols()) == declared
    # error channel: argument validation happens before any launch, so this works without a GPU
    rc = lib.hrf_scan_exclusive(None, 0, -1, None, None, None)
    assert rc != 0 and b"hrf_scan_exclusive" in lib.hrf_last_error()
    rc = lib.hrf_occgrid_create(0, 1, ctypes.byref(ctypes.c_void_p()))
    assert rc != 0 and b"grid_resolution" in lib.hrf_last_error()


def test_product_path_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; import humanrf_amd, humanrf_amd.ops, humanrf_amd.trainer, humanrf_amd.volume_rendering, "
            "humanrf_amd.scene_representation, humanrf_amd.dataset.synthetic, humanrf_amd.dataset.ray_sampler_native; "
            "bad=[m for m in sys.modules if m.split('.')[0] in ('oracle','tests')]; assert not bad, bad")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "humanrf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace(
                    "oracle/sampler_oracle.c", "").replace("oracle/)", "").replace("oracle/ ", ""), f


def test_reference_state_dict_round_trip_and_layout():
    """Checkpoint interchange: keys and sizes follow the reference's modules
    (feature_grids.{i}.{xyz,xyt,yzt,xzt}_encoding.params = tcnn's flat level-major table, 2 features per entry;
    sigma_net / color_net.params = tcnn's flat row-major (out, in) matrices), and loading restores every parameter."""
    from humanrf_amd.scene_representation import HumanRF
    kw = dict(density_scale=100, sorted_frame_numbers=tuple(range(15, 33)), n_features_per_level=2, log2_hashmap_size=14,
              n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
              n_hidden_layers_density=1, n_hidden_layers_color=2, sh_degree=4, segment_sizes=(6, 12), camera_embedding_dim=2,
              device="cpu")
    a = HumanRF(seed=1, **kw)
    b = HumanRF(seed=2, **kw)
    assert not torch.equal(a.table_params, b.table_params)
    sd = a.reference_state_dict()
    names = ("xyz", "xyt", "yzt", "xzt")
    total = 0
    for s, entries in enumerate(a.entries_per_segment):
        assert sd[f"feature_grids.{s}.vectors"].shape == (4, 2048, 32)
        for nm in names:
            p = sd[f"feature_grids.{s}.{nm}_encoding.params"]
            assert p.dtype == torch.float32 and p.numel() == entries * 2
            total += p.numel()
        # a segment's table is the concatenation of its 16 levels (sizes from the level table, multiples of 8 entries)
        meta = a._metas_host[s]
        assert sum(int(meta.levels[l].size) for l in range(16)) == entries
        assert int(meta.levels[0].offset) == 0 and all(int(meta.levels[l].size) % 8 == 0 for l in range(16))
    assert total == a.table_params.numel()
    assert sd["sigma_net.params"].numel() == 64 * 32 + 16 * 64
    assert sd["color_net.params"].numel() == 64 * a.color_in_pad + 64 * 64 + 16 * 64
    assert sd["camera_embeddings.weight"].shape == (160, 2) or sd["camera_embeddings.weight"].shape[1] == 2
    assert sd["frame_numbers_to_segment_numbers"][15:33].tolist() == [0] * 6 + [1] * 12
    b.load_reference_state_dict(sd)
    for x, y in ((a.table_params, b.table_params), (a.vectors, b.vectors), (a.sigma_params, b.sigma_params),
                 (a.color_params, b.color_params), (a.camera_embeddings.weight, b.camera_embeddings.weight)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("hidden", [1, 2, 3])
def test_colour_network_depth_knob_layout(hidden):
    """n_hidden_layers_color (model_args.py:31): color_net.params has tcnn's flat layout [w1 (64, in_pad) | hidden - 1 matrices (64, 64) |
    w3 (16, 64)]; split_color hands the kernels first / stacked middle / last; other depths, widths and a second sigma_net layer are
    refused by name."""
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd import ops
    kw = dict(density_scale=100, sorted_frame_numbers=tuple(range(15, 27)), n_features_per_level=2, log2_hashmap_size=14,
              n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
              n_hidden_layers_density=1, sh_degree=4, segment_sizes=(12,), camera_embedding_dim=2, device="cpu")
    m = HumanRF(n_hidden_layers_color=hidden, **kw)
    assert m.color_params.numel() == 64 * 48 + 4096 * (hidden - 1) + 1024 == m.reference_state_dict()["color_net.params"].numel()
    w1, mid, w3 = m.split_color(m.color_params.detach())
    assert (w1.numel(), mid.numel(), w3.numel()) == (64 * 48, 4096 * (hidden - 1), 1024)
    assert torch.equal(torch.cat([w1, mid, w3]), m.color_params.detach())
    assert ops._color_depth(mid) == hidden and ops._color_depth(mid, torch.zeros_like(mid)) == hidden
    with pytest.raises(RuntimeError):
        ops._color_depth(torch.zeros(3 * 4096))
    with pytest.raises(RuntimeError):
        ops._color_depth(mid, torch.zeros(mid.numel() + 4096))
    for bad in (dict(n_hidden_layers_color=0), dict(n_hidden_layers_color=4), dict(n_hidden_layers_color=2, n_neurons=32),
                dict(n_hidden_layers_color=2, n_hidden_layers_density=2)):
        with pytest.raises(NotImplementedError, match="n_hidden_layers_color"):
            HumanRF(**{**kw, **bad})


def test_level_table_reproduces_survey_appendix_b():
    """Entries per encoding and number of dense levels for the five segment sizes (computed there
    from tcnn's published sizing rule): the host-side level table must land on the same totals."""
    from humanrf_amd.scene_representation import hashgrid
    want = {100: (6_984_576, 4), 50: (3_695_768, 3), 25: (1_947_288, 2), 12: (1_015_808, 1), 6: (524_288, 1)}
    for seg, (entries, dense) in want.items():
        metas, per_seg, total = hashgrid.build_segment_meta((seg,), 16, 19, 32, 2048)
        assert per_seg == [entries] and total == 4 * entries
        m = metas[0]
        assert sum(1 for l in range(16) if not m.levels[l].hashed) == dense
        assert int(m.levels[0].res) == 32 and int(m.levels[0].size) == 32768
        hashed_sizes = {int(m.levels[l].size) for l in range(16) if m.levels[l].hashed}
        assert hashed_sizes == {2 ** hashgrid.segment_log2_hashmap_size(seg, 19)}


def test_learning_rate_schedule_equals_torch_lambdalr():
    """run.py:101-104: Adam + LambdaLR(lambda step: lr_decay ** min(step / max_steps, 1)). The engine's step-indexed
    learning rate must reproduce the sequence the reference's scheduler hands to the optimizer."""
    from humanrf_amd.scene_representation import HumanRF
    from humanrf_amd.trainer import TrainEngine
    m = HumanRF(density_scale=100, sorted_frame_numbers=tuple(range(15, 21)), n_features_per_level=2, log2_hashmap_size=14,
                n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
                n_hidden_layers_density=1, n_hidden_layers_color=2, sh_degree=4, segment_sizes=(6,), camera_embedding_dim=0,
                device="cpu")
    eng = TrainEngine(m, loader=None, lr=1e-2, lr_decay=0.3, max_steps=7, fast_collect=False)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: 0.3 ** min(step / 7, 1))
    for _ in range(12):
        assert abs(eng.lr() - opt.param_groups[0]["lr"]) < 1e-12
        p.grad = torch.ones(1); opt.step(); sched.step()
        eng.sched_step += 1
    assert abs(eng.lr() - 1e-2 * 0.3) < 1e-12   # clamped after max_steps


def test_header_is_plain_c_and_library_is_usable_from_c(tmp_path):
    """include/hrf.h compiles as strict C99 and a C program can drive the library through dlopen (version query,
    error channel, struct layouts the ctypes mirror assumes)."""
    import subprocess
    from humanrf_amd import _lib
    exe = tmp_path / "abi_smoke"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-ldl"])
    out = subprocess.check_output([str(exe), _lib.LIB_PATH]).decode().strip()
    assert out == "ok"
    assert ctypes.sizeof(_lib.LevelMeta) == 20 and ctypes.sizeof(_lib.SegmentMeta) == 16 + 20 * _lib.HRF_MAX_LEVELS


def test_ctypes_signatures_have_the_arity_of_the_header():
    """Every prototype in include/hrf.h and its ctypes mirror in humanrf_amd/_lib.py must take the same number of
    arguments (a mismatch would shift every pointer after it)."""
    from humanrf_amd import _lib
    header = open(os.path.join(ROOT, "include", "hrf.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = dict(re.findall(r"\b(hrf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S))
    checked = 0
    for name, argtypes in _lib._SIGNATURES.items():
        params = protos[name].strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(argtypes), f"{name}: header has {n} parameters, ctypes mirror {len(argtypes)}"
        checked += 1
    assert checked >= 30
