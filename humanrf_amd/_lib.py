"""ctypes binding of libhrf_hip.so (C ABI in include/hrf.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded, importing any
operator raises. `build()` compiles it in-tree with hipcc for gfx950 (humanrf_amd/csrc/Makefile).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  (must be imported first: libhrf_hip.so binds to the HIP runtime torch already loaded)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libhrf_hip.so")
_lock = threading.Lock()
_lib = None

HRF_MAX_LEVELS = 16


class LevelMeta(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_float), ("res", ctypes.c_uint32), ("size", ctypes.c_uint32),
                ("offset", ctypes.c_uint32), ("hashed", ctypes.c_uint32)]


class SegmentMeta(ctypes.Structure):
    _fields_ = [("table_offset", ctypes.c_uint64), ("entries", ctypes.c_uint32), ("n_levels", ctypes.c_uint32),
                ("levels", LevelMeta * HRF_MAX_LEVELS)]


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("p16", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("group", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class GradScaler(ctypes.Structure):
    """hrf_grad_scaler (include/hrf.h)."""
    _fields_ = [("scale", ctypes.c_float), ("growth_factor", ctypes.c_float), ("backoff_factor", ctypes.c_float),
                ("growth_interval", ctypes.c_int32), ("growth_tracker", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]


def build(force: bool = False) -> str:
    """Compile libhrf_hip.so in-tree (cross-compiles without a GPU)."""
    src_dir = os.path.join(_PKG, "csrc")
    args = ["make", "-C", src_dir, "-j8", "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_VP, _I64, _I32, _F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

_SIGNATURES = {
    "hrf_abi_version": [],
    "hrf_occgrid_create": [ctypes.c_uint64, _I32, ctypes.POINTER(_VP)],
    "hrf_occgrid_add": [_VP, _VP, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _VP, ctypes.POINTER(_I64)],
    "hrf_occgrid_destroy": [_VP],
    "hrf_sampler_rays": [_VP] * 7 + [_I64, _I32, _I32, _I32, _F, _I32] + [_VP] * 4 + [_VP, _VP],
    "hrf_scan_exclusive": [_VP, _I32, _I64, _VP, _VP, _VP],
    "hrf_sampler_compact_rays": [_VP] * 10 + [_I64, _I64] + [_VP] * 10 + [_VP],
    "hrf_pool_replace": [_VP, _I32, _VP, _I64, _I32] + [_VP] * 11 + [_VP],
    "hrf_sampler_samples": [_VP] * 7 + [_I64, _VP, _I64, _I32, _F, _I32] + [_VP] * 3 + [_I64, _VP],
    "hrf_compose_fwd": [_VP] * 6 + [_I64, _I32, _I32, _VP, _VP],
    "hrf_compose_bwd": [_VP] * 7 + [_I64, _I32, _I32] + [_VP] * 5 + [_VP],
    "hrf_query_prep": [_VP] * 6 + [_F, _VP, _VP, _I64, _VP, _VP, _VP],
    "hrf_encode4d_fwd": [_VP] * 5 + [_I32, _I32, _I64, _VP, _VP, _VP],
    "hrf_encode4d_density_fwd": [_VP] * 5 + [_I32, _I32, _I64, _VP, _VP, _VP, _VP, _F, _VP, _VP, _I32, _VP],
    "hrf_encode4d_bwd": [_VP] * 5 + [_I32, _I32, _I64, _VP, _I32, _F, _F, _VP, _VP, _VP, _VP],
    "hrf_encode4d_bwd_tables_binned": [_VP] * 4 + [_I32, _I32, _I64, _VP, _F, _F, _VP, _VP, _I64, _I32, _VP, _VP],
    "hrf_scatter_emit": [_VP] * 4 + [_I32, _I32, _I64, _VP, _F, _F, _VP, _VP, _I64, _I32, _VP],
    "hrf_scatter_accumulate": [_VP, _I32, _VP, _VP, _I64, _I32, _VP, _I32, _I32, _VP],
    "hrf_scatter_accumulate_signalled": [_VP, _I32, _VP, _VP, _I64, _I32, _VP, _VP, _I32, _VP, _VP],
    "hrf_stream_wait_value64": [_VP, _VP, ctypes.c_uint64],
    "hrf_hashgrid_fwd": [_VP, _VP, _VP, _I32, _I64, _VP, _VP],
    "hrf_hashgrid_bwd": [_VP, _VP, _I32, _I64, _VP, _I32, _F, _VP, _VP],
    "hrf_density_mlp_fwd": [_VP, _VP, _VP, _F, _I64, _VP, _VP, _I32, _VP],
    "hrf_color_mlp_fwd": [_VP] * 5 + [_I32, _I32, _VP, _VP, _VP, _I64, _VP, _I32, _I32, _I32, _VP],
    "hrf_mlp_bwd": [_VP] * 5 + [_I32, _I32] + [_VP] * 5 + [_F, _VP, _VP, _I64, _VP, _I32, _F] + [_VP] * 7 + [_I32, _I32, _I32, _VP],
    "hrf_density_mlp_bwd": [_VP, _VP, _VP, _VP, _I64, _VP, _I32, _F, _VP, _VP, _VP, _I32, _VP],
    "hrf_color_mlp_bwd": [_VP] * 5 + [_I32, _I32, _VP, _VP, _VP, _VP, _VP, _F, _I64] + [_VP] * 6 + [_I32, _I32, _I32, _VP],
    "hrf_ray_offsets": [_VP, _I64, _I64, _VP, _VP],
    "hrf_visibility": [_VP, _VP, _VP, _I64, _F, _F, _F, _VP, _VP, _VP],
    "hrf_prune_march": [_VP] * 6 + [_F, _F, _F] + [_VP] * 5 + [_I32, _I32, _VP, _VP, _F, _I64, _VP, _I64] + [_VP] * 5
                       + [_VP, ctypes.c_uint32, _VP, _I32] + [_VP],
    "hrf_ray_segment_order": [_VP, _VP, _I64, _VP, _I32, _VP, _VP, _VP],
    "hrf_ray_segment_order_values": [_VP, _VP, _I64, _VP, _I32, _VP, _VP, _VP, _VP, _VP],
    "hrf_pack_runs_sorted": [_VP] * 5 + [_I64] + [_VP] * 16 + [_VP],
    "hrf_batch_plan": [_VP, _VP] + [_I64] * 7 + [_VP, _VP, _VP],
    "hrf_pack_runs": [_VP] * 4 + [_I64, _VP, _I64, _VP, _VP, _VP],
    "hrf_compact_samples": [_VP] * 4 + [_I64, _VP, _VP, _VP],
    "hrf_composite_fwd": [_VP] * 5 + [_I64, _F, _VP, _VP, _VP],
    "hrf_composite_bwd": [_VP] * 7 + [_I64, _F, _VP, _VP, _VP],
    "hrf_loss_fwd_bwd": [_VP] * 4 + [_I64, _I64, _F, _F, _F, _VP, _VP, _VP] + [_VP] * 4 + [_VP],
    "hrf_render_loss_fused": [_VP] * 6 + [_I64, _I64, _F, _F, _F, _F] + [_VP] * 9 + [_VP],
    "hrf_adam_step": [_VP] * 5 + [_I64] + [_F] * 7 + [_VP, _VP],
    "hrf_adam_multi": [_VP, _I32, _I32, _I64] + [_F] * 5 + [_VP, _VP, _VP, _VP],
    "hrf_uniform_fill": [ctypes.c_uint32, _I64, _VP, _VP],
    "hrf_weights_fwd": [_VP] * 4 + [_I64, _VP, _VP],
    "hrf_weights_bwd": [_VP] * 5 + [_I64, _VP, _VP],
    "hrf_accumulate_fwd": [_VP, _VP, _I32, _VP, _I64, _VP, _VP],
    "hrf_accumulate_bwd": [_VP, _VP, _I32, _VP, _VP, _I64, _VP, _VP, _VP],
    "hrf_occgrid_from_masks": [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _VP],
    "hrf_mask_dilate": [_VP, _I32, _I32, _I32, _I64, _VP, _VP],
}


def lib() -> ctypes.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: the HIP extension is not built (run `python -c 'import "
                    f"__graft_entry__ as g; g.build()'` or `make -C humanrf_amd/csrc`). There is no CPU fallback.")
            l = ctypes.CDLL(LIB_PATH)
            l.hrf_last_error.restype = ctypes.c_char_p
            l.hrf_adam_workspace_bytes.restype = ctypes.c_size_t
            l.hrf_adam_workspace_bytes.argtypes = []
            l.hrf_scatter_workspace_bytes.restype = ctypes.c_size_t
            l.hrf_scatter_workspace_bytes.argtypes = [_I64, _I32]
            l.hrf_scatter_signals_per_segment.restype = ctypes.c_int64
            l.hrf_scatter_signals_per_segment.argtypes = [_I32]
            l.hrf_can_stream_wait_value.restype = ctypes.c_int
            l.hrf_can_stream_wait_value.argtypes = []
            for name, argtypes in _SIGNATURES.items():
                fn = getattr(l, name)  # AttributeError here = the library does not export what hrf.h declares
                fn.argtypes = argtypes
                fn.restype = ctypes.c_int
            if l.hrf_abi_version() != 10:
                raise RuntimeError("libhrf_hip.so ABI version mismatch")
            _lib = l
    return _lib


def exported_symbols():
    """Names hrf.h declares (used by the CPU-side ABI test)."""
    return ["hrf_last_error", "hrf_adam_workspace_bytes", "hrf_scatter_workspace_bytes", "hrf_scatter_signals_per_segment",
            "hrf_can_stream_wait_value"] + list(_SIGNATURES.keys())


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(lib().hrf_last_error().decode())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    return None if t is None else t.data_ptr()
