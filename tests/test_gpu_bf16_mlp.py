"""mlp_precision="bf16" (BASELINE.json configs[4]: fp16 hash tables + bf16 MLPs on the matrix cores) against the oracle's
bf16 mode (oracle/hrf_oracle.py: mlp(..., precision="bf16")). The reference has no bf16 configuration: the fp16 path is the
one the reference fixtures pin; this variant is checked against its definition, with tolerances of its own:
bf16 carries 8 significand bits (2^-8 relative per rounding), fp16 carries 11."""
import numpy as np
import pytest
import torch

from oracle import hrf_oracle as O
from tests.util import make_model, oracle_model_from, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _queries(n, frames, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 3, generator=g) - 0.5
    fn = torch.tensor(frames)[torch.randint(0, len(frames), (n,), generator=g)].reshape(n, 1).int()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    cams = torch.randint(0, 8, (n, 1), generator=g, dtype=torch.int32)
    return pos, fn, d, cams


@pytest.mark.parametrize("emb", [0, 2])
def test_bf16_field_forward(emb):
    from humanrf_amd.scene_representation import QueryInput
    frames = list(range(15, 27))
    m = make_model(DEV, (12,), tuple(frames), log2_T=16, emb=emb, table_scale=0.5, mlp_precision="bf16")
    assert m._sigma_h.dtype == torch.bfloat16 and m._color_h.dtype == torch.bfloat16 and m._tables_h.dtype == torch.float16
    om = oracle_model_from(m)
    assert om.mlp_precision == "bf16"
    n = 2049
    pos, fn, d, cams = _queries(n, frames, 2)
    for training in (True, False):
        with torch.no_grad():
            q = m(QueryInput(is_training=training, positions=pos.to(DEV), directions=d.to(DEV), frame_numbers=fn.to(DEV),
                             camera_numbers=cams.to(DEV)))
            sig_ref, geo_ref, _ = O.model_density(om, pos, fn)
            _, rgb_ref = O.model_forward(om, pos, d, fn, cams, training)
        # 2 bf16 ulps of the magnitude: a rounding that falls on the other side of a tie moves a value by one ulp
        assert torch.allclose(q.geometry_features.float().cpu(), geo_ref, atol=2e-2, rtol=1.6e-2)
        assert torch.allclose(q.density.cpu(), sig_ref, rtol=1e-1, atol=1e-2)
        assert float((q.radiance.cpu() - rgb_ref).abs().max()) <= 1.6e-2
        # ... and most values agree exactly (same roundings in the same places)
        same = (q.geometry_features.float().cpu() == geo_ref).float().mean()
        assert float(same) > 0.9, float(same)
    # the bf16 network is a different function from the fp16 one: the test would not pass against the fp16 oracle mode
    om16 = oracle_model_from(m); om16.mlp_precision = "fp16"
    with torch.no_grad():
        _, geo16, _ = O.model_density(om16, pos, fn)
    assert float((q.geometry_features.float().cpu() == geo16).float().mean()) < 0.5


@pytest.mark.parametrize("emb", [0, 2])
def test_bf16_field_backward(emb):
    from humanrf_amd.scene_representation import QueryInput
    frames = list(range(15, 27))
    m = make_model(DEV, (6, 6), tuple(frames), log2_T=15, emb=emb, table_scale=0.5, mlp_precision="bf16")
    om = oracle_model_from(m, requires_grad=True)
    n = 1500
    pos, fn, d, cams = _queries(n, frames, 4)
    g = torch.Generator().manual_seed(5)
    w_sig = torch.randn(n, generator=g) * 1e-4
    w_rgb = torch.randn(n, 3, generator=g)
    q = m(QueryInput(is_training=True, positions=pos.to(DEV), directions=d.to(DEV), frame_numbers=fn.to(DEV),
                     camera_numbers=cams.to(DEV)))
    ((q.density * w_sig.to(DEV)).sum() + (q.radiance * w_rgb.to(DEV)).sum()).backward()
    sig, rgb = O.model_forward(om, pos, d, fn, cams, True)
    ((sig * w_sig).sum() + (rgb * w_rgb).sum()).backward()

    def close(a, b, name, cos_min=0.998, rel_max=6e-2):
        # the kernel rounds the back-propagated activations' gradients to bf16 as well (the oracle's autograd keeps them
        # in fp32): 2^-8 per rounding, a handful of roundings per path, averaged over the batch in the weight gradients
        a, b = a.double().reshape(-1).cpu(), b.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        assert cos >= cos_min and rel <= rel_max, (name, cos, rel)

    close(m.sigma_params.grad, torch.cat([w.grad.reshape(-1) for w in om.sigma_w]), "sigma_net")
    close(m.color_params.grad, torch.cat([w.grad.reshape(-1) for w in om.color_w]), "color_net")
    close(m.vectors.grad, torch.stack([v.grad for v in om.vectors]), "vectors")
    close(m.table_params.grad, torch.cat([t.grad.reshape(-1) for seg in om.tables for t in seg]), "tables")
    if emb:
        close(m.camera_embeddings.weight.grad, om.camera_embeddings.grad, "camera_embeddings")


def test_bf16_prune_march_equals_unfused_bf16_sequence():
    """The fused prune march in bf16 mode evaluates the same density network as hrf_encode4d_fwd + hrf_density_mlp_fwd in
    bf16 mode: the kept samples are the same set, for training (jitter) and inference."""
    import copy
    import humanrf_amd.volume_rendering as vr
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    scene = small_scene(DEV)
    loader = SyntheticDataLoader(scene, batch_size=900, max_buffer_size=8, max_num_frames_per_batch=3, seed=2)
    m = make_model(DEV, (6, 6), tuple(scene.frame_numbers), log2_T=15, table_scale=0.4, mlp_precision="bf16")
    torch.manual_seed(5)
    base = next(iter(loader))
    for training in (True, False):
        outs = []
        try:
            for fused in (True, False):
                vr.FUSED_PRUNE = fused
                ib = copy.copy(base)
                ib.sample_distances = base.sample_distances.clone(); ib.ray_indices = base.ray_indices.clone()
                torch.manual_seed(77)
                vr.prune_samples(ib, m, training)
                outs.append((ib.sample_distances.clone(), ib.ray_indices.clone()))
        finally:
            vr.FUSED_PRUNE = True
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert 100 < outs[0][0].numel() < base.num_samples


def test_bf16_training_converges_and_keeps_bf16_copies_fresh():
    """TrainEngine with a bf16 model: the fused Adam refreshes the bf16 weight copies (round-to-nearest-even of the fp32
    masters, bit-exact), the loss falls as with fp16."""
    from humanrf_amd.dataset.synthetic import SyntheticDataLoader
    from humanrf_amd.trainer import TrainEngine
    frames = tuple(range(15, 27))
    scene = small_scene(DEV, G=64, W=96, H=96, frames=frames, num_cameras=12)
    losses = {}
    for prec in ("fp16", "bf16"):
        torch.manual_seed(0)
        m = make_model(DEV, (12,), frames, log2_T=15, emb=2, mlp_precision=prec)
        loader = SyntheticDataLoader(scene, batch_size=1024, max_buffer_size=12, max_num_frames_per_batch=4, seed=5)
        iter(loader)
        eng = TrainEngine(m, loader, samples_max_batch_size=60_000, rays_initial_batch_size=1024)
        ls = []
        for _ in range(150):
            st = eng.train_iteration()
            ls.append(float(st.sums[0]) / max(st.num_rays, 1))          # mean Huber term
        losses[prec] = (np.mean(ls[:10]), np.mean(ls[-10:]))
        assert eng.found_inf() == 0
        if prec == "bf16":
            assert torch.equal(m._sigma_h, m.sigma_params.detach().bfloat16())
            assert torch.equal(m._color_h, m.color_params.detach().bfloat16())
            assert torch.equal(m._tables_h[:m.table_params.numel()], m.table_params.detach().half())
    assert losses["bf16"][1] < 0.5 * losses["bf16"][0], losses
    assert losses["bf16"][1] < 1.5 * losses["fp16"][1] + 1e-4, losses
