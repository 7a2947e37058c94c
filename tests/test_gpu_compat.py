"""humanrf_amd.compat.nerfacc against the oracle's restatement of nerfacc 0.3.1: visibility
bit-exact, weights / accumulation and their gradients to fp32 tolerance."""
import pytest
import torch

from oracle import hrf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rays(seed=0, n_rays=300):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, 150, (n_rays,), generator=g)
    counts[::17] = 0                                   # empty rays
    counts[5] = 300                                    # longer than several wavefronts
    ray = torch.repeat_interleave(torch.arange(n_rays), counts)
    n = int(counts.sum())
    t0 = torch.rand(n, generator=g).cumsum(0) * 1e-3
    dt = 4e-4 * (0.5 + torch.rand(n, generator=g))
    sigma = torch.exp(torch.randn(n, generator=g) * 2.0 + 3.0)
    return ray, t0, t0 + dt, sigma, n_rays


def test_render_visibility_matches_oracle_bit_exact():
    import humanrf_amd.compat.nerfacc as nerfacc
    ray, t0, t1, sigma, R = _rays(1)
    alphas = 1.0 - torch.exp(-sigma * (t1 - t0))
    got = nerfacc.render_visibility(alphas.to(DEV), ray_indices=ray.to(DEV), early_stop_eps=1e-4, alpha_thre=1e-4, n_rays=R)
    ref = O.render_visibility(alphas, ray, 1e-4, 1e-4)
    assert got.dtype == torch.bool and torch.equal(got.cpu(), ref)
    assert 0 < int(ref.sum()) < ref.numel()


def test_weights_and_accumulate_values_and_gradients():
    import humanrf_amd.compat.nerfacc as nerfacc
    ray, t0, t1, sigma, R = _rays(2)
    g = torch.Generator().manual_seed(7)
    values = torch.rand(sigma.numel(), 3, generator=g)
    up_c, up_a = torch.randn(R, 3, generator=g), torch.randn(R, 1, generator=g)

    s_ref = sigma.clone().double().requires_grad_(True)
    v_ref = values.clone().double().requires_grad_(True)
    w_ref = O.render_weight_from_density(t0.double(), t1.double(), s_ref, ray)
    c_ref = O.accumulate_along_rays(w_ref, ray, v_ref, R)
    a_ref = O.accumulate_along_rays(w_ref, ray, None, R)
    ((c_ref * up_c.double()).sum() + (a_ref * up_a.double()).sum()).backward()

    s = sigma.clone().to(DEV).requires_grad_(True)
    v = values.clone().to(DEV).requires_grad_(True)
    w = nerfacc.render_weight_from_density(t0.to(DEV), t1.to(DEV), s, ray_indices=ray.to(DEV), n_rays=R)
    c = nerfacc.accumulate_along_rays(w, ray.to(DEV), v, R)
    a = nerfacc.accumulate_along_rays(w, ray.to(DEV), None, R)
    ((c * up_c.to(DEV)).sum() + (a * up_a.to(DEV)).sum()).backward()

    assert w.shape == (sigma.numel(), 1) and c.shape == (R, 3) and a.shape == (R, 1)
    assert float((w.detach().cpu().double().reshape(-1) - w_ref.detach()).abs().max()) <= 2e-6      # fp32 vs fp64
    assert float((c.detach().cpu().double() - c_ref.detach()).abs().max()) <= 1e-5
    assert float((a.detach().cpu().double() - a_ref.detach()).abs().max()) <= 1e-5
    assert bool((a.detach().cpu()[::17] == 0).all())                                                # empty rays
    gs, gs_ref = s.grad.cpu().double(), s_ref.grad
    rel = float((gs - gs_ref).norm() / gs_ref.norm())
    assert rel <= 1e-4, rel
    gv, gv_ref = v.grad.cpu().double(), v_ref.grad
    assert float((gv - gv_ref).abs().max()) <= 1e-5
