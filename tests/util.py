"""Helpers shared by the parity tests: build small scenes / models and mirror them into the CPU oracle."""
import numpy as np
import torch

from oracle import hrf_oracle as O

PLS = float(np.exp(np.log(2048 / 32) / 15))


def oracle_model_from(model, requires_grad=False) -> O.OracleModel:
    """CPU OracleModel holding exactly the values the kernels read (fp16-rounded tables and MLP weights)."""
    from humanrf_amd.scene_representation import hashgrid
    model._refresh_half()
    tables_h = model._tables_h.detach().float().cpu()
    levels, tables, vectors = [], [], []
    off = 0
    for s, size in enumerate(model.segment_sizes):
        entries = model.entries_per_segment[s]
        meta = model._metas_host[s]
        lv = [O.Level(float(meta.levels[l].scale), int(meta.levels[l].res), int(meta.levels[l].size),
                      int(meta.levels[l].offset), bool(meta.levels[l].hashed)) for l in range(16)]
        levels.append(lv)
        seg_tables = []
        for e in range(4):
            t = tables_h[off * 2:(off + entries) * 2].reshape(entries, 2).clone()
            seg_tables.append(t.requires_grad_(requires_grad))
            off += entries
        tables.append(seg_tables)
        vectors.append(model.vectors[s].detach().float().cpu().clone().requires_grad_(requires_grad))
    sw = model._sigma_h.detach().float().cpu()
    cw = model._color_h.detach().float().cpu()
    kin = model.color_in_pad
    sigma_w = [sw[:2048].reshape(64, 32).clone().requires_grad_(requires_grad),
               sw[2048:].reshape(16, 64).clone().requires_grad_(requires_grad)]
    n_mid = int(getattr(model, "n_hidden_layers_color", 2)) - 1      # tcnn's flat layout [w1 | hidden-to-hidden ... | w3]
    color_w = [cw[:64 * kin].reshape(64, kin).clone().requires_grad_(requires_grad)]
    color_w += [cw[64 * kin + 4096 * i:64 * kin + 4096 * (i + 1)].reshape(64, 64).clone().requires_grad_(requires_grad)
                for i in range(n_mid)]
    color_w += [cw[64 * kin + 4096 * n_mid:].reshape(16, 64).clone().requires_grad_(requires_grad)]
    emb = None
    if model.camera_embedding_dim > 0:
        emb = model.camera_embeddings.weight.detach().float().cpu().clone().requires_grad_(requires_grad)
    return O.OracleModel(levels=levels, tables=tables, vectors=vectors, sigma_w=sigma_w, color_w=color_w,
                         frame_to_segment=model.frame_numbers_to_segment_numbers.cpu().long(),
                         frame_to_local=model.frame_numbers_to_normalized_local_frame_numbers.cpu(),
                         density_scale=float(model.density_scale), camera_embeddings=emb,
                         mlp_precision=getattr(model, "mlp_precision", "fp16"))


def oracle_levels_check(model):
    """The product's host-side level table must equal the oracle's independent restatement."""
    out = []
    for s, size in enumerate(model.segment_sizes):
        meta = model._metas_host[s]
        out.append([(float(meta.levels[l].scale), int(meta.levels[l].res), int(meta.levels[l].size),
                     int(meta.levels[l].offset), bool(meta.levels[l].hashed)) for l in range(16)])
    return out


def make_model(device="cuda", segment_sizes=(12,), frames=tuple(range(15, 27)), log2_T=15, emb=0, seed=1337,
               table_scale=None, mlp_precision="fp16", n_hidden_color=2):
    from humanrf_amd.scene_representation import HumanRF
    m = HumanRF(density_scale=100, sorted_frame_numbers=frames, n_features_per_level=2, log2_hashmap_size=log2_T,
                n_levels=16, coarsest_resolution=32, finest_resolution=2048, geometry_feature_dim=15, n_neurons=64,
                n_hidden_layers_density=1, n_hidden_layers_color=n_hidden_color, sh_degree=4, segment_sizes=segment_sizes,
                camera_embedding_dim=emb, device=device, seed=seed, mlp_precision=mlp_precision)
    if table_scale is not None:  # larger table values so the outputs are not dominated by the initial 1e-4 range
        with torch.no_grad():
            g = torch.Generator().manual_seed(seed + 1)
            m.table_params.copy_(((torch.rand(m.table_params.numel(), generator=g) * 2 - 1) * table_scale).to(device))
    return m


def small_scene(device="cuda", G=64, W=48, H=40, frames=tuple(range(15, 27)), num_cameras=6):
    from humanrf_amd.dataset.synthetic import SyntheticScene
    return SyntheticScene(frames, num_cameras=num_cameras, width=W, height=H, grid_resolution=G, device=device)


def record_parity(test: str, what: str, rel: float, cos: float, bound: float) -> None:
    """HRF_RECORD_PARITY=<file>: append the measured gradient distance of a parity test (tools/measure.sh gradparity writes
    profiles/r06_gradient_parity_measured.txt from it); the bounds in the tests are set from these numbers."""
    import os
    path = os.environ.get("HRF_RECORD_PARITY")
    if path:
        with open(path, "a") as f:
            f.write("%-70s %-44s rel-L2 %.3e  1-cos %.2e  bound %.1e\n" % (test, what, rel, 1.0 - cos, bound))
