"""
oracle/ref_harness.py -- runs the REFERENCE's own hot-path Python from /root/reference on the CPU. TEST INFRASTRUCTURE ONLY.

`load()` puts the reference's sources on sys.path, installs stand-ins for the packages that are not in the
repository (tinycudann, nerfacc, the nvcc-built tensor_composition_native and sampler extensions, cv2 ...) and imports the
reference modules unmodified. Two backends: "cpu" = oracle/ref_stubs.py (used by tests/golden/make_ref_fixtures.py, which
freezes reference outputs into tests/golden/ref_*.npz, and by the CPU tests that compare live against the imported
reference); "hip" = humanrf_amd's own drop-in modules, so that the reference's loop bodies run over libhrf_hip.so
(tests/test_gpu_reference_dropin.py). /root/reference does not exist on the GPU box: there the sources are the byte
copies oracle/snapshot_reference.py leaves under oracle/_ref/reference (git-ignored). smoke() and bench.py never call
this module.

Adaptations made to the reference at import time (each is plumbing, none changes arithmetic), all in `load()`:
  * `Decomposition4D.to` is a no-op: HumanRF.density moves active segments to "cuda" and the others to "cpu"
    on every call (humanrf.py:171,179); there is no CUDA device here and the move has no numerical effect.
  * `Trainer` is instantiated without `__init__` (which needs lpips weights, a workspace directory and a checkpoint
    scan, trainer.py:49-103); `make_trainer` sets exactly the attributes `train_step` / `_calculate_losses` read,
    with the same constructors run.py:101-104 and trainer.py:74,89-90 use. The GradScaler is `torch.amp.GradScaler("cpu")`
    (trainer.py:74 builds the CUDA one, which disables itself without a device).
"""
from __future__ import annotations

import os
import sys
import types
import warnings
from types import SimpleNamespace

import torch

# Where the reference's Python is read from: /root/reference in the build container; on the GPU box the byte copies
# oracle/snapshot_reference.py left under oracle/_ref/reference (git-ignored, pushed by gpurun with the built libraries).
_SNAPSHOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference")


def _reference_root() -> str:
    env = os.environ.get("HRF_REFERENCE_ROOT")
    if env:
        return env
    return "/root/reference" if os.path.isdir("/root/reference/humanrf") else _SNAPSHOT


REFERENCE_ROOT = _reference_root()
_NS = {}          # backend -> namespace
_ACTIVE = None    # backend whose stand-ins are in sys.modules right now
_STAND_INS = ("tinycudann", "nerfacc", "humanrf.scene_representation.tensor_composition_native",
              "actorshq.dataset.occupancy_grid_native", "actorshq.dataset.ray_sampler_native")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "humanrf"))


def _purge() -> None:
    """Forget the reference's modules and the stand-ins, so that the next import binds the other backend. Namespaces
    handed out earlier keep working: their classes hold the module globals they were defined with."""
    for name in list(sys.modules):
        if name in _STAND_INS or name in ("humanrf", "actorshq") or name.startswith(("humanrf.", "actorshq.")):
            del sys.modules[name]


def _install_hip() -> None:
    """The product's drop-in modules under the names the reference imports: every kernel the
    reference's source then reaches is libhrf_hip.so's."""
    from . import ref_stubs
    import humanrf_amd.compat.nerfacc as nerfacc
    import humanrf_amd.compat.tinycudann as tcnn
    import humanrf_amd.dataset.occupancy_grid_native as occ
    import humanrf_amd.dataset.ray_sampler_native as sampler
    import humanrf_amd.scene_representation.tensor_composition_native as tc
    sys.modules["tinycudann"], sys.modules["nerfacc"] = tcnn, nerfacc
    sys.modules["humanrf.scene_representation.tensor_composition_native"] = tc
    sys.modules["actorshq.dataset.ray_sampler_native"] = sampler
    sys.modules["actorshq.dataset.occupancy_grid_native"] = occ
    ref_stubs.install()     # setdefault: only the inert ones (cv2, lpips, tensorboardX, skimage, simple_parsing) are added


def load(backend: str = "cpu") -> SimpleNamespace:
    """-> namespace with the reference's callables / classes, imported over
         backend "cpu": oracle/ref_stubs.py (the oracle's restatement of tcnn / nerfacc / the compose op) -- CPU tests, fixtures;
         backend "hip": humanrf_amd's drop-in modules, i.e. libhrf_hip.so -- the GPU drop-in tests.
    Raises RuntimeError when the reference's sources are not on this machine."""
    global _ACTIVE
    if backend not in ("cpu", "hip"):
        raise ValueError(backend)
    if backend in _NS and _ACTIVE == backend:
        return _NS[backend]
    if not available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present: run oracle/snapshot_reference.py where /root/reference exists")
    if _ACTIVE is not None:
        _purge()
    from . import ref_stubs
    if backend == "hip":
        _install_hip()
    else:
        ref_stubs.install()
    _ACTIVE = backend
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import actorshq.dataset.input_batch as r_input_batch
        import humanrf.adaptive_temporal_partitioning as r_atp
        import humanrf.input as r_input
        import humanrf.scene_representation.decomposition4d as r_d4
        import humanrf.scene_representation.humanrf as r_humanrf
        import humanrf.scene_representation.query_io as r_query_io
        import humanrf.trainer as r_trainer
        import humanrf.utils.activation as r_activation
        import humanrf.utils.loss as r_loss
        import humanrf.volume_rendering as r_vr
    for mod in (r_input_batch, r_atp, r_input, r_d4, r_humanrf, r_query_io, r_trainer, r_activation, r_loss, r_vr):
        assert os.path.abspath(mod.__file__).startswith(os.path.abspath(REFERENCE_ROOT)), mod.__file__
    r_d4.Decomposition4D.to = lambda self, *a, **k: self   # see module docstring
    import actorshq.dataset.data_loader as r_loader
    _NS[backend] = SimpleNamespace(
        DataLoader=r_loader.DataLoader, backend=backend,
        InputBatch=r_input_batch.InputBatch, merge_input_batches=r_input.merge_input_batches,
        truncated_exp=r_activation.truncated_exp, bce_loss=r_loss.bce_loss,
        QueryInput=r_query_io.QueryInput, QueryOutput=r_query_io.QueryOutput,
        Decomposition4D=r_d4.Decomposition4D, HumanRF=r_humanrf.HumanRF,
        prune_samples=r_vr.prune_samples, render=r_vr.render, RenderOutput=r_vr.RenderOutput,
        Trainer=r_trainer.Trainer, compute_adaptive_segment_sizes=r_atp.compute_adaptive_segment_sizes,
        get_segment_size=r_atp.get_segment_size, get_final_segment_size=r_atp.get_final_segment_size,
        PREDEFINED_SEGMENT_SIZES=r_atp.PREDEFINED_SEGMENT_SIZES, modules=SimpleNamespace(
            volume_rendering=r_vr, humanrf=r_humanrf, decomposition4d=r_d4, trainer=r_trainer, input=r_input,
            data_loader=r_loader))
    return _NS[backend]


def make_model(ref: SimpleNamespace, frames, segment_sizes, log2_T: int = 19, emb: int = 0, density_scale: float = 100.0):
    """The reference's HumanRF with the example configuration (humanrf/configs/example_humanrf.py, model_args.py:10-47)."""
    return ref.HumanRF(density_scale=density_scale, sorted_frame_numbers=tuple(frames), n_features_per_level=2,
                       log2_hashmap_size=log2_T, n_levels=16, coarsest_resolution=32, finest_resolution=2048,
                       geometry_feature_dim=15, n_neurons=64, n_hidden_layers_density=1, n_hidden_layers_color=2,
                       sh_degree=4, segment_sizes=tuple(segment_sizes), camera_embedding_dim=emb)


def make_trainer(ref: SimpleNamespace, model, lr: float = 1e-2, lr_decay: float = 0.5, max_steps: int = 50_001,
                 bce_loss_weight: float = 1e-3, init_scale: float = 65536.0, growth_interval: int = 2000,
                 device: str = "cpu"):
    """A reference Trainer carrying what train_step reads (trainer.py:205-255), built like run.py:101-104 builds it."""
    tr = ref.Trainer.__new__(ref.Trainer)
    tr.model = model
    tr.config = SimpleNamespace(training=SimpleNamespace(bce_loss_weight=bce_loss_weight, max_steps=max_steps, lr=lr,
                                                         lr_decay=lr_decay))
    tr.optimizer = torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)           # run.py:101
    tr.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(tr.optimizer, lambda step: lr_decay ** min(step / max_steps, 1))
    tr.scaler = torch.amp.GradScaler(device, init_scale=init_scale, growth_interval=growth_interval)  # trainer.py:74
    tr.photometric_loss = torch.nn.HuberLoss(reduction="mean", delta=0.01)                        # trainer.py:89
    tr.mask_loss = ref.bce_loss                                                                   # trainer.py:90
    tr.step = 0
    return tr


class GridDataset:
    """What compute_adaptive_segment_sizes needs of a VolumetricDataset: get_occupancy_grid(frame_number) -> uint8 grid
    (adaptive_temporal_partitioning.py:79; volumetric_dataset.py:151-153 returns a fresh array per call)."""

    def __init__(self, grids_by_frame):
        self.grids = grids_by_frame

    def get_occupancy_grid(self, frame_number: int):
        return self.grids[frame_number].copy()


_ = types  # (kept: harness users build SimpleNamespace / ModuleType objects next to it)
