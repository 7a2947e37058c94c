"""Case definitions shared by tests/golden/make_ref_fixtures.py (which runs the REFERENCE on them, here, where
/root/reference exists) and by the tests that replay the same cases through the oracle (CPU) and the HIP path (GPU).

Everything that is large is regenerated from a seed with NumPy's PCG64 (bit-stable across machines and versions)
instead of being stored: model parameters in the reference's state-dict layout, query points, loss coefficients.
Only what the reference computed is frozen in tests/golden/ref_*.npz."""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

from oracle import hrf_oracle as O

PLS = float(np.exp(np.log(2048 / 32) / 15))          # decomposition4d.py:73
ENC_NAMES = ("xyz", "xyt", "yzt", "xzt")             # decomposition4d.py:79-122

# name -> (first frame, number of frames, segment_sizes, model log2_hashmap_size, camera_embedding_dim, samples)
# per-segment table sizes follow humanrf.py:107-109: 6 -> 2^15, 12 -> 2^16, 50 -> 2^18, 100 -> 2^19 at log2_T 19
FIELD_CASES: Dict[str, tuple] = {
    "seg12_T15": (15, 12, (12,), 15, 2, 3000),                    # small tables (2^12), embedding on
    "seg50_T19": (15, 50, (50,), 19, 0, 3000),                    # --partitioning none: one 2^18 table, 3 dense levels
    "seg100_T19": (15, 100, (100,), 19, 0, 3000),                 # 2^19 tables, 4 dense levels (Appendix B)
    "bench7_T19": (15, 50, (6, 6, 6, 12, 6, 6, 12), 19, 2, 4000),  # the bench's adaptive partition of 50 frames
}


def rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def segment_log2(segment_size: int, log2_T: int) -> int:
    return int(np.round(np.log2(segment_size / 100 * (2 ** log2_T))))   # humanrf.py:107-109


def seeded_reference_state(segment_sizes: Sequence[int], log2_T: int, emb: int, seed: int, table_scale: float = 0.4,
                           vec_scale: float = 0.6) -> Dict[str, torch.Tensor]:
    """Parameters in the reference's state-dict keys and layouts (trainer.py:528-581), with values
    large enough that a wrong table index or a wrong weight shows up far above fp16 rounding."""
    g = rng(seed)
    sd = {}
    for s, size in enumerate(segment_sizes):
        levels = O.hashgrid_levels(16, segment_log2(size, log2_T), 32, PLS)
        n = sum(lv.size for lv in levels) * 2
        sd[f"feature_grids.{s}.vectors"] = torch.from_numpy((g.standard_normal((4, 2048, 32), dtype=np.float32) * vec_scale))
        for nm in ENC_NAMES:
            sd[f"feature_grids.{s}.{nm}_encoding.params"] = torch.from_numpy(
                ((g.random(n, dtype=np.float32) * 2.0 - 1.0) * table_scale).astype(np.float32))

    def xavier(o, i):
        b = np.float32(np.sqrt(6.0 / (i + o)))
        return ((g.random((o, i), dtype=np.float32) * 2.0 - 1.0) * b).reshape(-1)

    kin = 16 * ((31 + emb + 15) // 16)
    sd["sigma_net.params"] = torch.from_numpy(np.concatenate([xavier(64, 32), xavier(16, 64)]))
    sd["color_net.params"] = torch.from_numpy(np.concatenate([xavier(64, kin), xavier(64, 64), xavier(16, 64)]))
    if emb > 0:
        sd["camera_embeddings.weight"] = torch.from_numpy(g.standard_normal((160, emb), dtype=np.float32))
    return sd


def field_inputs(name: str):
    """Query points of a field case: short runs of consecutive march samples (step 4e-4 along random directions, like
    the sampler emits them) mixed over all frames -> dict of CPU tensors."""
    f0, nf, segs, log2_T, emb, n = FIELD_CASES[name]
    g = rng(1000 + sum(ord(c) for c in name))
    run = 25
    n_runs = (n + run - 1) // run
    start = (g.random((n_runs, 3), dtype=np.float32) - 0.5) * 0.9
    d = g.standard_normal((n_runs, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = np.arange(run, dtype=np.float32)[None, :, None]
    pos = (start[:, None, :] + d[:, None, :] * (k * np.float32(4e-4) * 40.0)).reshape(-1, 3)[:n]   # 40-step stride: crosses cells
    pos = np.clip(pos, -0.5, 0.5).astype(np.float32)
    frames = np.repeat(g.integers(f0, f0 + nf, n_runs), run)[:n].astype(np.int32)
    cams = np.repeat(g.integers(0, 160, n_runs), run)[:n].astype(np.int32)
    dirs = np.repeat(d, run, axis=0)[:n].astype(np.float32)
    a = (g.random(n, dtype=np.float32) + 0.5) * 1e-2      # loss = sum(sigma*a) + sum(rgb*b): gradients of O(0.1..10),
    b = g.random((n, 3), dtype=np.float32) + 0.5          # inside fp16's normal range without loss scaling
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    return dict(positions=t(pos), frames=t(frames).view(-1, 1), cams=t(cams).view(-1, 1), directions=t(dirs),
                a=t(a), b=t(b), sorted_frames=tuple(range(f0, f0 + nf)), segment_sizes=segs, log2_T=log2_T, emb=emb)


def sample_indices(numel: int, k: int, seed: int) -> np.ndarray:
    """k distinct sorted positions in [0, numel) (the entries of a large gradient that a fixture keeps)."""
    g = rng(seed)
    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    return np.unique(g.integers(0, numel, size=2 * k))[:k].astype(np.int64)


def oracle_model_from_state(sd: Dict[str, torch.Tensor], sorted_frames, segment_sizes, log2_T: int, emb: int,
                            f2s: torch.Tensor, f2l: torch.Tensor, requires_grad: bool = False) -> O.OracleModel:
    """OracleModel straight from a reference-layout state dict (values rounded to fp16 where the kernels / tcnn read
    fp16 copies: tables and MLP weights)."""
    levels, tables, vectors = [], [], []
    rh = lambda x: x.half().float()
    for s, size in enumerate(segment_sizes):
        lv = O.hashgrid_levels(16, segment_log2(size, log2_T), 32, PLS)
        levels.append(lv)
        tables.append([rh(sd[f"feature_grids.{s}.{nm}_encoding.params"]).reshape(-1, 2).clone().requires_grad_(requires_grad)
                       for nm in ENC_NAMES])
        vectors.append(sd[f"feature_grids.{s}.vectors"].clone().requires_grad_(requires_grad))
    kin = 16 * ((31 + emb + 15) // 16)
    sw, cw = rh(sd["sigma_net.params"]), rh(sd["color_net.params"])
    sigma_w = [sw[:2048].reshape(64, 32).clone().requires_grad_(requires_grad),
               sw[2048:].reshape(16, 64).clone().requires_grad_(requires_grad)]
    color_w = [cw[:64 * kin].reshape(64, kin).clone().requires_grad_(requires_grad),
               cw[64 * kin:64 * kin + 4096].reshape(64, 64).clone().requires_grad_(requires_grad),
               cw[64 * kin + 4096:].reshape(16, 64).clone().requires_grad_(requires_grad)]
    e = sd["camera_embeddings.weight"].clone().requires_grad_(requires_grad) if emb > 0 else None
    return O.OracleModel(levels=levels, tables=tables, vectors=vectors, sigma_w=sigma_w, color_w=color_w,
                         frame_to_segment=f2s.long(), frame_to_local=f2l.float(), density_scale=100.0, camera_embeddings=e)
