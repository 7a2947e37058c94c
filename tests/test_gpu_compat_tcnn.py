"""The tinycudann-shaped modules (humanrf_amd.compat.tinycudann) composed the way the reference composes tcnn's
(decomposition4d.py:79-135: four Encodings + the compose op; humanrf.py:123-208: per-segment dispatch, sigma_net,
truncated_exp, colour network) must reproduce the outputs and gradients the reference's own classes produced
(tests/golden/ref_field_*.npz). The reference's own SOURCE over these modules: tests/test_gpu_reference_dropin.py."""
import numpy as np
import pytest
import torch

from tests import refcases as RC
from tests.test_gpu_ref_fixtures import _load, _rel_cos

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _ComposeFn(torch.autograd.Function):   # decomposition4d.py:8-39
    @staticmethod
    def forward(ctx, a, b, c, d, vectors, xyzt):
        from humanrf_amd.scene_representation import tensor_composition_native as tc
        ctx.save_for_backward(a, b, c, d, vectors, xyzt)
        return tc.compose_tensors_forward(a, b, c, d, vectors, xyzt)

    @staticmethod
    def backward(ctx, g):
        from humanrf_amd.scene_representation import tensor_composition_native as tc
        a, b, c, d, vectors, xyzt = ctx.saved_tensors
        da, db, dc, dd, dv = tc.compose_tensors_backward(a, b, c, d, vectors, xyzt, g.contiguous())
        return da, db, dc, dd, dv, None


class _Decomposition4D(torch.nn.Module):     # decomposition4d.py:42-135 over the compat Encoding
    def __init__(self, log2_T):
        super().__init__()
        import humanrf_amd.compat.tinycudann as tcnn
        cfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2_T,
               "base_resolution": 32, "per_level_scale": RC.PLS}
        self.vectors = torch.nn.Parameter(torch.zeros(4, 2048, 32, device=DEV))
        self.xyz_encoding, self.xyt_encoding = tcnn.Encoding(3, cfg), tcnn.Encoding(3, cfg)
        self.yzt_encoding, self.xzt_encoding = tcnn.Encoding(3, cfg), tcnn.Encoding(3, cfg)

    def forward(self, xyz, times):
        xyzt = torch.cat((xyz, times), axis=-1)
        return _ComposeFn.apply(self.xyz_encoding(xyz), self.xyt_encoding(xyzt[..., [0, 1, 3]].contiguous()),
                                self.yzt_encoding(xyzt[..., [1, 2, 3]].contiguous()),
                                self.xzt_encoding(xyzt[..., [0, 2, 3]].contiguous()), self.vectors, xyzt.contiguous())


class _HumanRFLike(torch.nn.Module):         # humanrf.py:14-208 over the compat modules
    def __init__(self, frames, segs, log2_T, emb):
        super().__init__()
        import humanrf_amd.compat.tinycudann as tcnn
        from humanrf_amd.scene_representation import hashgrid
        f2s, f2l = hashgrid.frame_tables(frames, segs)
        self.register_buffer("f2s", torch.from_numpy(f2s).to(DEV))
        self.register_buffer("f2l", torch.from_numpy(f2l).to(DEV))
        self.emb = emb
        if emb > 0:
            self.camera_embeddings = torch.nn.Embedding(160, emb).to(DEV)
        self.feature_grids = torch.nn.ModuleList([_Decomposition4D(RC.segment_log2(s, log2_T)) for s in segs])
        mlp = {"otype": "FullyFusedMLP", "activation": "ReLU", "n_neurons": 64}
        self.sigma_net = tcnn.Network(32, 16, dict(mlp, output_activation="None", n_hidden_layers=1))
        self.color_net = tcnn.NetworkWithInputEncoding(
            18 + emb, 3, {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4},
                                                            {"otype": "Identity"}]},
            dict(mlp, output_activation="Sigmoid", n_hidden_layers=2))

    def density(self, positions, frame_numbers):
        from humanrf_amd.utils.activation import truncated_exp
        fn = frame_numbers.squeeze(1).long()
        segments = self.f2s[fn]
        features = torch.empty(positions.shape[0], 32, dtype=torch.half, device=DEV)
        for s in range(len(self.feature_grids)):
            m = segments == s
            if bool(m.any()):
                features[m] = self.feature_grids[s](positions[m] + 0.5, self.f2l[fn[m]].unsqueeze(-1))
        h = self.sigma_net(features)
        return truncated_exp(h[..., 0]) * 100.0, h[..., 1:], features

    def forward(self, positions, directions, frame_numbers, camera_numbers):
        sigma, geo, _ = self.density(positions, frame_numbers)
        inp = [(directions + 1) * 0.5, geo]
        if self.emb > 0:
            inp.append(self.camera_embeddings(camera_numbers.squeeze(1).long()))
        return sigma, self.color_net(torch.cat(inp, dim=-1))


@pytest.mark.parametrize("name", ["seg12_T15", "bench7_T19"])
def test_tcnn_shaped_modules_reproduce_the_reference_classes(name):
    fx = _load(f"ref_field_{name}.npz")
    inp = RC.field_inputs(name)
    segs, log2_T, emb = inp["segment_sizes"], inp["log2_T"], inp["emb"]
    sd = RC.seeded_reference_state(segs, log2_T, emb, seed=500 + len(name))
    m = _HumanRFLike(inp["sorted_frames"], segs, log2_T, emb)
    missing = m.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)   # the reference's own key names
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"f2s", "f2l"}, missing
    pos, frames, cams, dirs = (inp[k].to(DEV) for k in ("positions", "frames", "cams", "directions"))
    sigma, rgb = m(pos, dirs, frames, cams)
    with torch.no_grad():
        _, geo, feats = m.density(pos, frames)
    ref = fx["d4_features"].astype(np.float32)
    assert np.abs(feats.float().cpu().numpy() - ref).max() <= 2 ** -8 + 1e-3 * np.abs(ref).max()
    assert np.allclose(sigma.detach().cpu().numpy(), fx["density"], rtol=2e-2, atol=1e-3)
    assert np.allclose(geo.float().cpu().numpy(), fx["geo"].astype(np.float32), rtol=4e-3, atol=4e-3)
    assert np.abs(rgb.detach().float().cpu().numpy() - fx["radiance"].astype(np.float32)).max() <= 4e-3
    loss = (sigma * inp["a"].to(DEV)).sum() + (rgb.float() * inp["b"].to(DEV)).sum()
    loss.backward()
    for got, key in ((m.sigma_net.params.grad, "g_sigma"), (m.color_net.params.grad, "g_color")):
        rel, cos = _rel_cos(got.cpu().numpy(), fx[key])
        assert cos >= 0.999 and rel <= 3e-2, (name, key, rel, cos)
    if emb > 0:
        rel, cos = _rel_cos(m.camera_embeddings.weight.grad.cpu().numpy(), fx["g_emb"])
        assert cos >= 0.999 and rel <= 3e-2, (name, "emb", rel, cos)
    for s in range(len(segs)):
        fg = m.feature_grids[s]
        rel, cos = _rel_cos(fg.vectors.grad[:, ::16, :].cpu().numpy(), fx[f"g_vec{s}"])
        assert cos >= 0.999 and rel <= 3e-2, (name, "vectors", s, rel, cos)
        for e, nm in enumerate(RC.ENC_NAMES):
            g = getattr(fg, f"{nm}_encoding").params.grad.cpu()
            idx = fx[f"g_tab{s}_{e}_idx"]
            if idx.size > 8:
                rel, cos = _rel_cos(g[idx].numpy(), fx[f"g_tab{s}_{e}_val"])
                assert cos >= 0.999 and rel <= 3e-2, (name, s, nm, rel, cos)


def test_decomposition4d_module_and_row_major_gradient_modes():
    """humanrf_amd.scene_representation.Decomposition4D (one segment's fused encoder behind the reference's class name) and
    the fp16 / fp32 row-major forms of hrf_encode4d_bwd it reaches: same gradients as the level-major form the training
    engine uses."""
    from humanrf_amd import ops
    from humanrf_amd.scene_representation import Decomposition4D
    fx = _load("ref_field_seg12_T15.npz")
    inp = RC.field_inputs("seg12_T15")
    sd = RC.seeded_reference_state((12,), 15, 2, seed=500 + len("seg12_T15"))
    d4 = Decomposition4D(ngp_log2_hashmap_size=RC.segment_log2(12, 15), device=DEV)
    with torch.no_grad():
        d4.vectors.copy_(sd["feature_grids.0.vectors"])
        d4.tables.copy_(torch.stack([sd[f"feature_grids.0.{nm}_encoding.params"].view(-1, 2) for nm in RC.ENC_NAMES]))
    pos, frames = inp["positions"].to(DEV), inp["frames"]
    tloc = ((frames.view(-1).float() - 15.0) / 12.0).view(-1, 1).to(DEV)       # humanrf.py:79-98 for one 12-frame segment
    feats = d4(pos + 0.5, tloc)
    ref = fx["d4_features"].astype(np.float32)
    assert np.abs(feats.detach().float().cpu().numpy() - ref).max() <= 2 ** -8 + 1e-3 * np.abs(ref).max()
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(feats.shape, generator=g).to(DEV)
    feats.backward(dy.half())                                                   # fp16 row-major d_features (autograd's form)
    gt_h, gv_h = d4.tables.grad.clone().reshape(-1), d4.vectors.grad.clone()
    xyzt = torch.cat([pos + 0.5, tloc], 1).contiguous()
    seg = torch.zeros(xyzt.shape[0], dtype=torch.int32, device=DEV)
    d4._refresh_half()
    _, enc = ops.encode4d_fwd(xyzt, seg, d4._tables_h, d4.vectors.detach().unsqueeze(0).contiguous(), d4._seg_meta, 1, True)
    res = {}
    for mode in ("row32", "lm"):
        gt, gv = torch.zeros_like(gt_h), torch.zeros_like(gv_h)
        d = dy.half().float().contiguous()
        if mode == "lm":
            d = d.view(-1, 16, 2).permute(1, 0, 2).contiguous()
        ops.encode4d_bwd(xyzt, seg, enc, d4.vectors.detach().unsqueeze(0).contiguous(), d4._seg_meta, 1, d, 1.0, gt, gv.unsqueeze(0),
                         level_major=mode == "lm")
        res[mode] = (gt, gv)
    for mode in res:
        for a, b, what in ((res[mode][0], gt_h, "tables"), (res[mode][1], gv_h, "vectors")):
            rel, cos = _rel_cos(a.cpu().numpy(), b.cpu().numpy())
            assert cos >= 0.9999 and rel <= 5e-3, (mode, what, rel, cos)


def test_density_is_differentiable_like_the_reference():
    """HumanRF.density under grad mode carries gradients to the tables / vectors / sigma_net (the reference's forward()
    differentiates through density(), humanrf.py:188-189); under no_grad it is the plain kernel pair."""
    from humanrf_amd.scene_representation import QueryInput
    from tests.util import make_model
    m = make_model(DEV, (6, 6), tuple(range(15, 27)), log2_T=15, emb=2, table_scale=0.4)
    g = torch.Generator().manual_seed(0)
    pos = (torch.rand(700, 3, generator=g) - 0.5).to(DEV)
    fr = torch.randint(15, 27, (700, 1), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand(700, generator=g).to(DEV) * 1e-2
    q = m.density(QueryInput(is_training=True, positions=pos, frame_numbers=fr))
    assert q.density.requires_grad
    (q.density * w).sum().backward()
    g_dens = [p.grad.clone() for p in (m.table_params, m.vectors, m.sigma_params)]
    for p in (m.table_params, m.vectors, m.sigma_params, m.color_params):
        p.grad = None
    d = torch.nn.functional.normalize(torch.randn(700, 3, generator=g), dim=1).to(DEV)
    qf = m(QueryInput(is_training=True, positions=pos, directions=d, frame_numbers=fr, camera_numbers=torch.zeros_like(fr)))
    (qf.density * w).sum().backward()
    for a, b in zip(g_dens, (m.table_params.grad, m.vectors.grad, m.sigma_params.grad)):
        assert float(a.abs().sum()) > 0
        rel, cos = _rel_cos(a.cpu().numpy(), b.cpu().numpy())
        assert cos >= 0.9999 and rel <= 1e-2, (rel, cos)
    with torch.no_grad():
        q0 = m.density(QueryInput(is_training=False, positions=pos, frame_numbers=fr))
    assert not q0.density.requires_grad and torch.allclose(q0.density, q.density.detach(), rtol=1e-6)
