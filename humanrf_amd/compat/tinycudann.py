"""tinycudann's torch module surface, as far as the reference uses it (decomposition4d.py:79-122, humanrf.py:123-156), on
the gfx950 kernels: `import humanrf_amd.compat.tinycudann as tcnn` lets the reference's own Decomposition4D / HumanRF
classes run unmodified. Three modules, each with ONE flat fp32 `params` Parameter in tcnn's layout (so that state dicts
interchange with `HumanRF.reference_state_dict()`), outputs in torch.half like tcnn:

  Encoding(3, {"otype": "HashGrid", ...})                       -> hrf_hashgrid_fwd / hrf_hashgrid_bwd
  Network(2 n_levels <= 32, 1 + G <= 16, FullyFusedMLP ReLU/None, 64 neurons, 1 hidden) -> hrf_density_mlp_fwd (MFMA, weights in LDS)
  NetworkWithInputEncoding(3+G+E, 3, Composite[SH4, Identity], FullyFusedMLP ReLU/Sigmoid, 64 neurons, 1..3 hidden)
                                                                 -> hrf_color_mlp_fwd (MFMA, weights in LDS)

The training engine and humanrf_amd's own HumanRF never go through these (they use the fused encode / MLP / backward
kernels); this is the compatibility surface SURVEY.md 8(b) lists. The two networks' BACKWARD passes are the MFMA backward
kernel with one network compiled out (hrf_density_mlp_bwd / hrf_color_mlp_bwd), run like tcnn's torch binding runs its
half-precision backward: dL/dy is multiplied by loss_scale = 128 on entry and every gradient divided by it on exit; an
fp16 overflow inside turns the weight gradients non-finite so that a GradScaler skips the step. Gradients with respect to
the three direction inputs of the colour network are not produced (the reference feeds ray directions, which carry no
gradient, humanrf.py:192)."""
from __future__ import annotations

import math
from typing import Dict

import torch

from .. import _lib, ops
from .._lib import LevelMeta, SegmentMeta, check, ptr, stream_ptr
from ..scene_representation import hashgrid


LOSS_SCALE = 128.0   # tcnn's default loss_scale for half-precision modules (bindings/torch/tinycudann/modules.py)


def _device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())


# ------------------------------------------------------------------------------------------------ HashGrid encoding
class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, params):
        module._refresh_half()
        n = x.shape[0]
        out = torch.empty(n, module.n_output_dims, dtype=torch.float16, device=x.device)
        check(_lib.lib().hrf_hashgrid_fwd(ptr(x), ptr(module._params_h), ptr(module._meta), module.n_levels, n, ptr(out),
                                          stream_ptr()))
        ctx.module = module
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        module = ctx.module
        d = d_out.contiguous()
        fp32 = d.dtype == torch.float32
        scale = 1.0
        if not fp32:
            # tcnn's torch binding multiplies dL/dy by its loss_scale (128) before the half-precision backward and divides
            # the parameter gradients by it afterwards: small gradients stay above fp16's subnormals
            scale = LOSS_SCALE
            d = (d.float() * scale).half()
        d_params = torch.zeros_like(module.params)
        check(_lib.lib().hrf_hashgrid_bwd(ptr(x), ptr(module._meta), module.n_levels, x.shape[0], ptr(d), 1 if fp32 else 0,
                                          scale, ptr(d_params), stream_ptr()))
        return None, None, d_params


class Encoding(torch.nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: Dict, seed: int = 1337, dtype=None):
        super().__init__()
        if encoding_config.get("otype") != "HashGrid" or n_input_dims != 3:
            raise NotImplementedError("only the 3-D HashGrid encoding the reference instantiates (decomposition4d.py:79-122)")
        if int(encoding_config["n_features_per_level"]) != 2 or int(encoding_config["n_levels"]) > _lib.HRF_MAX_LEVELS:
            raise NotImplementedError("kernels are specialised for 2 features per level and at most 16 levels")
        self.n_input_dims = n_input_dims
        self.n_levels = int(encoding_config["n_levels"])
        self.n_output_dims = 2 * self.n_levels
        lv = hashgrid.level_table(self.n_levels, int(encoding_config["log2_hashmap_size"]),
                                  int(encoding_config["base_resolution"]), float(encoding_config["per_level_scale"]))
        self.entries = lv[-1][3] + lv[-1][2]
        metas = (SegmentMeta * 1)()
        metas[0].table_offset, metas[0].entries, metas[0].n_levels = 0, self.entries, self.n_levels
        for l, row in enumerate(lv):
            metas[0].levels[l] = LevelMeta(*row)
        dev = _device()
        self.register_buffer("_meta", torch.frombuffer(bytearray(bytes(metas)), dtype=torch.uint8).clone().to(dev), persistent=False)
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(((torch.rand(self.entries * 2, generator=g) * 2.0 - 1.0) * 1e-4).to(dev))  # A.1
        self.register_buffer("_params_h", torch.zeros(self.entries * 2 + 2, dtype=torch.float16, device=dev), persistent=False)
        self._ver = None

    def mark_dirty(self) -> None:
        """Call after writing the parameters through `.data` (which does not bump the version counter)."""
        self._ver = None

    def _refresh_half(self) -> None:   # tcnn gathers from an fp16 copy of the fp32 masters
        ver = (self.params._version, self.params.data_ptr(), self._params_h.data_ptr())
        # a training-mode module re-casts every call, like tcnn does per step: optimizers that write through .data
        # (and fused ones) leave the version counter alone
        if ver != self._ver or (self.training and self.params.requires_grad):
            with torch.no_grad():
                self._params_h[:self.params.numel()].copy_(self.params)
            self._ver = (self.params._version, self.params.data_ptr(), self._params_h.data_ptr())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _HashGridFn.apply(self, x.float().contiguous(), self.params)


# ------------------------------------------------------------------------------------------------ FullyFusedMLP modules
def _xavier(o: int, i: int, g: torch.Generator) -> torch.Tensor:
    return ((torch.rand(o, i, generator=g) * 2.0 - 1.0) * math.sqrt(6.0 / (i + o))).reshape(-1)


def _check_mlp(network_config: Dict, hidden, out_act: str) -> None:
    ok = (network_config.get("otype") == "FullyFusedMLP" and network_config.get("activation") == "ReLU"
          and network_config.get("output_activation") == out_act and int(network_config.get("n_neurons")) == 64
          and int(network_config.get("n_hidden_layers")) in (hidden if isinstance(hidden, tuple) else (hidden,)))
    if not ok:
        raise NotImplementedError("only the FullyFusedMLP shapes the reference instantiates (humanrf.py:123-156)")


class _SigmaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, params):
        n_in, pad = module.n_input_dims, module.in_pad
        xh = x.half().contiguous()
        ph = params.detach().half()
        w1, w2 = ph[:64 * pad], ph[64 * pad:].contiguous()
        if n_in < 32:
            # the kernels' rows are 32 wide: [x | ones up to tcnn's padded width | zeros], the first matrix (64, pad) in 32 columns
            row = torch.zeros(x.shape[0], 32, dtype=torch.float16, device=x.device)
            row[:, :n_in] = xh
            row[:, n_in:pad] = 1.0
            xh = row
            wide = torch.zeros(64, 32, dtype=torch.float16, device=x.device)
            wide[:, :pad] = w1.reshape(64, pad)
            w1 = wide.reshape(-1)
        w1 = w1.contiguous()
        h, _ = ops.density_mlp_fwd(xh, w1, w2, 1.0, want_h=True, want_sigma=False)
        ctx.module = module
        ctx.save_for_backward(xh, w1, w2)
        return h

    @staticmethod
    def backward(ctx, d_h):
        xh, w1, w2 = ctx.saved_tensors
        n_in, pad = ctx.module.n_input_dims, ctx.module.in_pad
        dev = xh.device
        g1 = torch.zeros(64 * 32, dtype=torch.float32, device=dev)
        g2 = torch.zeros(16 * 64, dtype=torch.float32, device=dev)
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        d = (d_h.float() * LOSS_SCALE).contiguous()
        d_x = ops.density_mlp_bwd(xh, w1, w2, d, g1, g2, flags, fp32_out=True)
        inv = 1.0 / LOSS_SCALE
        poison = torch.where(flags[0] != 0, float("inf"), 0.0).to(torch.float32)   # an fp16 overflow inside -> found_inf
        if pad < 32:
            g1 = g1.reshape(64, 32)[:, :pad].reshape(-1)
        return None, (d_x[:, :n_in] * inv).to(d_h.dtype), torch.cat([g1, g2]) * inv + poison


class Network(torch.nn.Module):
    """tcnn.Network(2 n_levels, 1 + geometry_feature_dim, FullyFusedMLP) = sigma_net (humanrf.py:123-133): up to 32 inputs -- padded with
    ones to a multiple of 16, the width of the first matrix in `params` [UPSTREAM-KNOWLEDGE: tcnn's padded input width; INTEGRATION.md
    section 2 says what is and is not verified about it] -- and up to 16 outputs."""

    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: Dict, seed: int = 1337):
        super().__init__()
        _check_mlp(network_config, 1, "None")
        if not 1 <= n_input_dims <= 32 or not 1 <= n_output_dims <= 16:
            raise NotImplementedError("sigma_net shape: at most 32 inputs, at most 16 outputs")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.in_pad = 16 * ((n_input_dims + 15) // 16)
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(torch.cat([_xavier(64, self.in_pad, g), _xavier(16, 64, g)]).to(_device()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _SigmaFn.apply(self, x, self.params)[:, :self.n_output_dims]


class _ColorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, params):
        n, E, G, kin = x.shape[0], module.emb_dim, module.geo_dim, module.in_pad
        xf = x.float()
        dirs = (xf[:, :3] * 2.0 - 1.0).contiguous()                      # the kernel maps [-1,1] back to [0,1] itself
        h = torch.zeros(n, 16, dtype=torch.float16, device=x.device)
        h[:, 1:1 + G] = xf[:, 3:3 + G]
        idx = torch.arange(n, device=x.device)
        emb = xf[:, 3 + G:3 + G + E].contiguous() if E > 0 else None
        ph = params.detach().half()
        mid = 64 * kin + 4096 * (module.n_hidden_layers - 1)
        w1, w2, w3 = ph[:64 * kin].contiguous(), ph[64 * kin:mid].contiguous(), ph[mid:].contiguous()
        cams = idx.int() if E > 0 else None
        rgb = ops.color_mlp_fwd(dirs, idx, h, emb, cams, E, E > 0, w1, w2, w3, geo_dim=G)
        ctx.module = module
        ctx.x_dtype = x.dtype
        ctx.save_for_backward(dirs, idx, h, emb, cams, w1, w2, w3)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        dirs, idx, h, emb, cams, w1, w2, w3 = ctx.saved_tensors
        m = ctx.module
        n, E, G, kin = h.shape[0], m.emb_dim, m.geo_dim, m.in_pad
        dev = h.device
        g1 = torch.zeros(64 * kin, dtype=torch.float32, device=dev)
        g2 = torch.zeros(64 * 64 * (m.n_hidden_layers - 1), dtype=torch.float32, device=dev)
        g3 = torch.zeros(16 * 64, dtype=torch.float32, device=dev)
        g_emb = torch.zeros(n, E, dtype=torch.float32, device=dev) if E > 0 else None   # every sample is its own "camera"
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        d = (d_rgb.float() * LOSS_SCALE).contiguous()
        d_h = ops.color_mlp_bwd(dirs, idx, h, emb, cams, E, E > 0, w1, w2, w3, d, g1, g2, g3, g_emb, flags, geo_dim=G)
        inv = 1.0 / LOSS_SCALE
        d_x = torch.zeros(n, m.n_input_dims, dtype=torch.float32, device=dev)
        d_x[:, 3:3 + G] = d_h[:, 1:1 + G] * inv
        if E > 0:
            d_x[:, 3 + G:3 + G + E] = g_emb * inv
        poison = torch.where(flags[0] != 0, float("inf"), 0.0).to(torch.float32)
        return None, d_x.to(ctx.x_dtype), torch.cat([g1, g2, g3]) * inv + poison


class NetworkWithInputEncoding(torch.nn.Module):
    """tcnn.NetworkWithInputEncoding(3 + G + E, 3, Composite[SphericalHarmonics(3, degree 4), Identity], FullyFusedMLP)
    = color_net (humanrf.py:135-156). The identity part (G geometry features, E embedding dimensions) is one run of columns for the network;
    the kernels take the first min(15, n - 3) of them through their geometry slot and the rest through the embedding slot -- the same
    input row [SH 16 | identity columns | ones] either way."""

    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: Dict, network_config: Dict, seed: int = 1337):
        super().__init__()
        nested = encoding_config.get("nested", [])
        ok = (encoding_config.get("otype") == "Composite" and len(nested) == 2
              and nested[0].get("otype") == "SphericalHarmonics" and nested[0].get("n_dims_to_encode") == 3
              and nested[0].get("degree") == 4 and nested[1].get("otype") == "Identity")
        if not ok or n_output_dims != 3 or not 4 <= n_input_dims <= 35:
            raise NotImplementedError("only the colour network the reference instantiates (humanrf.py:135-156): 3 direction inputs + 1..32 "
                                      "identity inputs")
        _check_mlp(network_config, (1, 2, 3), "Sigmoid")     # n_hidden_layers_color (model_args.py:31)
        self.n_hidden_layers = int(network_config["n_hidden_layers"])
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.geo_dim = min(15, n_input_dims - 3)
        self.emb_dim = n_input_dims - 3 - self.geo_dim
        self.in_pad = 16 * ((16 + self.geo_dim + self.emb_dim + 15) // 16)
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(torch.cat([_xavier(64, self.in_pad, g)] + [_xavier(64, 64, g) for _ in range(self.n_hidden_layers - 1)]
                                                   + [_xavier(16, 64, g)]).to(_device()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _ColorFn.apply(self, x, self.params)
