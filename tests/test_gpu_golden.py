"""HIP path against the committed golden vectors (tests/golden/hotpath_seed123.npz): sampler bit-exact, field and
rendering within the stated tolerances, gradients by cosine / relative L2."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_path_matches_golden_vectors():
    from humanrf_amd import ops
    from humanrf_amd.dataset import ray_sampler_native as rs
    from humanrf_amd.dataset.input_batch import InputBatch
    from humanrf_amd.dataset.occupancy_grid_native import OccupanyGrid
    from humanrf_amd.volume_rendering import render
    from tests.util import make_model
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "hotpath_seed123.npz")))
    T = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dt)
    G, W, H = int(g["in_G"]), int(g["in_W"]), int(g["in_H"])
    ring = OccupanyGrid(G, 3)
    tex = torch.tensor([ring.add_grid(T(g["in_grids"][i])) for i in range(3)], dtype=torch.int64, device=DEV)
    out = rs.get_samples_occupancy_minmax(
        T(g["in_rgba"]), torch.zeros(g["in_rgba"].shape[0], dtype=torch.bool, device=DEV), T(g["in_frames"]), T(g["in_cams"]),
        tex, torch.ones(3, dtype=torch.bool, device=DEV), T(g["in_idx"]), T(g["in_inverse_krs"]), T(g["in_camera_origins"]),
        T(g["in_aabb"]), G, W, H, 4e-4, False)
    for nm, a in zip(["origins", "dirs", "rgba_s", "frames_s", "cams_s", "minmax", "ray_mask", "t", "ray"], out):
        assert np.array_equal(a.cpu().numpy(), g["smp_" + nm]), f"sampler output {nm} is not bit-identical to the golden file"

    m = make_model(DEV, (6,), tuple(range(15, 21)), log2_T=12, emb=2, seed=1337, table_scale=0.3)
    chk = np.array([float(m.table_params.double().sum()), float(m.vectors.double().sum()), float(m.sigma_params.double().sum()),
                    float(m.color_params.double().sum()), float(m.camera_embeddings.weight.double().sum())])
    assert np.allclose(chk, g["param_checksum"], rtol=1e-9), "parameter RNG stream differs from the generator's"
    o, d, fr, cm, t, ray = out[0], out[1], out[3], out[4], out[7].clone(), out[8].long()
    # prune pass with the stored jitter
    xyzt, seg = ops.query_prep(o, d, fr, ray, t, T(g["jitter"]), m.frame_numbers_to_segment_numbers,
                               m.frame_numbers_to_normalized_local_frame_numbers)
    sigma, _ = m.density_from_xyzt(xyzt, seg)
    assert np.allclose(sigma.cpu().numpy(), g["prune_sigma"], rtol=2e-2, atol=1e-3)
    rs_ = ops.ray_offsets(ray, o.shape[0])
    vis, _ = ops.visibility(1.0 - torch.exp(-sigma * 4e-4), None, rs_, o.shape[0], 1e-4, 1e-4)
    flips = int((vis.cpu().numpy().astype(bool) != g["prune_vis"]).sum())
    assert flips <= max(3, 0.002 * vis.numel()), flips
    # field + rendering on exactly the golden file's visible samples
    gv = torch.from_numpy(g["prune_vis"]).to(DEV)
    ib = InputBatch(ray_origins=o, ray_directions=d, rgba=out[2], frame_numbers=fr.view(-1, 1), camera_numbers=cm.view(-1, 1),
                    sample_distances=t[gv].view(-1, 1), ray_indices=ray[gv], unique_frame_numbers=fr[:1].view(-1, 1))
    xyzt1, seg1 = xyzt[gv].contiguous(), seg[gv].contiguous()
    with torch.no_grad():
        sg, rgb, geo = m.field(xyzt1, seg1, d, ib.ray_indices, cm, True)
        feats, _ = ops.encode4d_fwd(xyzt1, seg1, m._tables_h, m.vectors.detach(), m._seg_meta, 1, False)
    k = g["features"].shape[0]
    assert float(np.abs(feats[:k].float().cpu().numpy() - g["features"].astype(np.float32)).max()) <= 2 ** -9
    assert np.allclose(sg.cpu().numpy(), g["sigma"], rtol=2e-2, atol=1e-3)
    assert np.allclose(geo[:k].float().cpu().numpy(), g["geo"].astype(np.float32), rtol=4e-3, atol=4e-3)
    assert float(np.abs(rgb.cpu().numpy() - g["rgb"].astype(np.float32)).max()) <= 4e-3
    ro = render(ib, m, T(g["background"]), True)
    assert float(np.abs(ro.color.detach().cpu().numpy() - g["color"]).max()) <= 2e-3
    assert float(np.abs(ro.weights_sum.detach().cpu().numpy() - g["acc"]).max()) <= 2e-3
    # loss and gradients through the reference-shaped autograd path
    gt_mask = out[2][:, 3:4]
    bg = T(g["background"])
    gt = out[2][:, :3] * gt_mask + bg * (1 - gt_mask)
    photo = torch.nn.functional.huber_loss(ro.color, gt, reduction="mean", delta=0.01)
    p = torch.clamp(ro.weights_sum, 0, 1)
    loss = photo + (-(gt_mask * torch.log(p + 1e-10) + (1 - gt_mask) * torch.log(1 - p + 1e-10))).mean() * 1e-3
    assert abs(float(loss) - g["loss"][0]) <= 2e-3 * abs(g["loss"][0]) + 1e-6
    # backward with the golden file's upstream gradients (the BCE gradient is ill-conditioned at acc ~ 0 / 1, so
    # both sides must start from identical d_color / d_acc), scaled like GradScaler's initial scale
    torch.autograd.backward([ro.color, ro.weights_sum], [T(g["d_color"]) * 65536.0, T(g["d_acc"]) * 65536.0])

    def close(a, b, name):
        a = a.double().cpu().reshape(-1) / 65536.0
        b = torch.from_numpy(b).double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        # gradient activations travel as fp16 inside the fused backward (as in tcnn): cos >= 0.999, rel-L2 <= 5e-2
        assert cos >= 0.999 and rel <= 5e-2, (name, cos, rel)
    close(m.sigma_params.grad, g["grad_sigma_w"], "sigma_net")
    close(m.color_params.grad, g["grad_color_w"], "color_net")
    close(m.vectors.grad[0][:, ::64, :], g["grad_vectors_rows"], "vectors")
    ent = m.entries_per_segment[0]
    tg = m.table_params.grad.view(4, ent, 2)
    close(tg[:, :512], g["grad_tables_l0"], "tables level 0")
    norms = np.array([float(tg[e].double().norm()) / 65536.0 for e in range(4)])
    assert np.allclose(norms, g["grad_tables_norm"], rtol=3e-2)
    close(m.camera_embeddings.weight.grad[:8], g["grad_emb"], "camera embeddings")
