#!/usr/bin/env python3
"""Generates tests/golden/hotpath_seed123.npz: frozen input/output vectors of the CPU oracle for the whole hot
path on a tiny synthetic scene (seed 123, humanrf/args/run_args.py:125). The reference ships no golden vectors
and cannot be imported or run here, so these vectors pin OUR oracle against drift and give the
GPU tests a fixture that does not depend on re-running the oracle. Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import hrf_oracle as O  # noqa: E402
from tests.util import make_model, oracle_model_from  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_seed123.npz")
FRAMES = tuple(range(15, 21))


def build_inputs():
    from humanrf_amd.dataset.synthetic import SyntheticScene
    torch.manual_seed(123)
    scene = SyntheticScene(FRAMES, num_cameras=4, width=24, height=20, grid_resolution=32, device="cpu")
    slots = [(0, 15), (1, 17), (3, 20)]
    rgba = torch.stack([scene.render_rgba(c, f) for c, f in slots]).reshape(-1, 4).numpy()
    grids = [scene.occupancy_grid(f).numpy() for _, f in slots]
    P = scene.width * scene.height
    g = torch.Generator().manual_seed(123)
    idx = torch.randint(0, len(slots) * P, (90,), generator=g, dtype=torch.int64).numpy()
    cams = np.array([c for c, _ in slots], np.int32)
    frames = np.array([f for _, f in slots], np.int32)
    return dict(rgba=rgba, grids=np.stack(grids), idx=idx, cams=cams, frames=frames,
                inverse_krs=scene.all_inverse_krs[cams].numpy(), camera_origins=scene.all_camera_origins[cams].numpy(),
                aabb=scene.aabb.numpy(), W=scene.width, H=scene.height, G=scene.grid_resolution)


def compute(inp):
    out = {}
    s = O.sampler_get_data(inp["rgba"], None, inp["frames"], inp["cams"], list(inp["grids"]), np.ones(3, bool), inp["idx"],
                           inp["inverse_krs"], inp["camera_origins"], inp["aabb"], int(inp["G"]), int(inp["W"]),
                           int(inp["H"]), 4e-4, False, True, True)
    names = ["origins", "dirs", "rgba_s", "frames_s", "cams_s", "minmax", "ray_mask", "t", "ray"]
    for n, a in zip(names, s):
        out["smp_" + n] = a
    m = make_model("cpu", (6,), FRAMES, log2_T=12, emb=2, seed=1337, table_scale=0.3)
    om = oracle_model_from(m, requires_grad=True)
    out["param_checksum"] = np.array([float(m.table_params.double().sum()), float(m.vectors.double().sum()),
                                      float(m.sigma_params.double().sum()), float(m.color_params.double().sum()),
                                      float(m.camera_embeddings.weight.double().sum())])
    o, d = torch.from_numpy(s[0]), torch.from_numpy(s[1])
    fr, cm = torch.from_numpy(s[3]), torch.from_numpy(s[4])
    t, ray = torch.from_numpy(s[7]), torch.from_numpy(s[8]).long()
    g = torch.Generator().manual_seed(7)
    jitter = torch.rand(t.shape[0], generator=g)
    t_j, _, vis, sigma = O.prune_samples(om, o, d, fr, t, ray, jitter)
    out["jitter"] = jitter.numpy()
    out["prune_sigma"], out["prune_vis"] = sigma.numpy(), vis.numpy()
    t1, r1 = t_j[vis], ray[vis]
    bg = torch.rand(o.shape[0], 3, generator=g)
    pos = o[r1] + t1 * d[r1]
    with torch.no_grad():
        sig1, geo1, feats1 = O.model_density(om, pos, fr[r1])
        _, rgb1 = O.model_forward(om, pos, d[r1], fr[r1], cm[r1], True)
    k = 1500  # keep the fixture small: per-sample tensors are stored for the first k visible samples
    out["features"], out["sigma"], out["geo"], out["rgb"] = feats1[:k].half().numpy(), sig1.numpy(), geo1[:k].half().numpy(), rgb1.half().numpy()
    color, acc = O.render(om, o, d, fr, cm, t1, r1, bg, True)
    color.retain_grad(); acc.retain_grad()
    loss, photo = O.training_loss(color, acc, torch.from_numpy(s[2]), bg)
    loss.backward()
    # the BCE term is ill-conditioned for rays whose acc is ~0 or ~1: the GPU test feeds these exact upstream
    # gradients into its backward instead of differentiating its own (differently rounded) acc
    out["d_color"], out["d_acc"] = color.grad.numpy().copy(), acc.grad.numpy().copy()
    out["background"], out["color"], out["acc"] = bg.numpy(), color.detach().numpy(), acc.detach().numpy()
    out["loss"] = np.array([float(loss), float(photo)])
    out["grad_sigma_w"] = torch.cat([w.grad.reshape(-1) for w in om.sigma_w]).numpy()
    out["grad_color_w"] = torch.cat([w.grad.reshape(-1) for w in om.color_w]).numpy()
    out["grad_vectors_rows"] = om.vectors[0].grad[:, ::64, :].numpy()          # every 64th row of the four vectors
    out["grad_tables_l0"] = torch.stack([om.tables[0][e].grad[:512] for e in range(4)]).numpy()  # head of level 0
    out["grad_tables_norm"] = np.array([float(om.tables[0][e].grad.double().norm()) for e in range(4)])
    out["grad_emb"] = om.camera_embeddings.grad[:8].numpy()
    return out


if __name__ == "__main__":
    inp = build_inputs()
    out = compute(inp)
    np.savez_compressed(OUT, **{"in_" + k: v for k, v in inp.items()}, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out["smp_t"]), "samples,", int(out["prune_vis"].sum()), "visible")
