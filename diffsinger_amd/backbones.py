"""Backbone registry and nn.Module shims over libdsdenoise (drop-in for `modules/backbones`).

`BACKBONES` / `build_backbone` have the reference's names, arguments and behaviour
(modules/backbones/__init__.py:6-18).  `WaveNet` / `LYNXNet` hold their parameters in ordinary torch
modules so that `state_dict()` names and shapes are exactly the reference's (wavenet.py:22-31,56-72;
lynxnet.py:52-62,71-74,104-124) and `load_ckpt(..., strict=True)` (utils/__init__.py:216) works; their
`forward` has the reference signature (wavenet.py:75-81, lynxnet.py:128-133) and runs ONLY on the HIP
library: a CPU tensor, a missing library or a gradient-requiring call raises instead of falling back.
"""
from __future__ import annotations

import ctypes as C
import inspect
import math

import torch
import torch.nn as nn

from . import _lib
from .hparams import hparams


def filter_kwargs(dict_to_filter, kwarg_obj):
    """utils/__init__.py:149-163: keep only the keys the callable accepts (all, if it takes **kwargs)."""
    sig = inspect.signature(kwarg_obj)
    params = sig.parameters.values()
    if any(p.kind == p.VAR_KEYWORD for p in params):
        return dict_to_filter.copy()
    keys = [p.name for p in params if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
    return {k: dict_to_filter[k] for k in keys if k in dict_to_filter}


class _KaimingConv1d(nn.Conv1d):
    """`Conv1d` of wavenet.py:12-15 / lynxnet.py:13-16 (kaiming-normal weight)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        nn.init.kaiming_normal_(self.weight)


class _NativeBackbone(nn.Module):
    """Common part: native handle life cycle, weight upload, cond hoist cache, forward()."""

    def __init__(self):
        super().__init__()
        self._handle = None
        self._handle_device = None
        self._weights_dirty = True
        self._cond_key = None
        self._cond_ref = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # -- life cycle ---------------------------------------------------------------------------
    def _mark_dirty(self):
        self._weights_dirty = True
        self._cond_key = None

    def _apply(self, fn, *a, **k):           # .to() / .cuda() / .float() move the parameters
        r = super()._apply(fn, *a, **k)
        self._mark_dirty()
        return r

    def refresh_native(self):
        """Call after mutating parameters in place (optimizer step, manual edits)."""
        self._mark_dirty()

    def _config(self, device_index):
        raise NotImplementedError

    def _extra_weights(self):
        return {}

    def _native_state(self):
        """name -> tensor of what the native handle loads (default: the whole state_dict)."""
        return dict(self.state_dict())

    def native_handle(self, device):
        if device.type != "cuda":
            raise RuntimeError(
                f"diffsinger_amd.{type(self).__name__} runs only on an MI355X (HIP) device; got a "
                f"{device.type} tensor. There is no CPU path - use the reference module for CPU.")
        lib = _lib.lib()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is not None and self._handle_device != idx:
            self.release_native()
        if self._handle is None:
            cfg = self._config(idx)
            hp = C.c_void_p()
            create = (lib.dsd_encoder_create if isinstance(cfg, _lib.DsdEncoderConfig) else
                      lib.dsd_token_encoder_create if isinstance(cfg, _lib.DsdTokenEncoderConfig) else
                      lib.dsd_vocoder_create if isinstance(cfg, _lib.DsdVocoderConfig) else lib.dsd_create)
            rc = create(C.byref(cfg), C.byref(hp))
            if rc != 0:
                raise _lib.NativeLibraryError(f"dsd_create failed ({rc}): {lib.dsd_last_error(None).decode()}")
            self._handle, self._handle_device = hp, idx
            self._weights_dirty = True
        if self._weights_dirty:
            tensors = self._native_state()
            tensors.update(self._extra_weights())
            for name, t in tensors.items():
                t = t.detach().to(device=device, dtype=torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.check(self._handle, lib.dsd_load_weight(self._handle, name.encode(), C.c_void_p(t.data_ptr()),
                                                            shape, t.dim(), 1), f"dsd_load_weight({name})")
            torch.cuda.synchronize(device)
            _lib.check(self._handle, lib.dsd_finalize_weights(self._handle), "dsd_finalize_weights")
            self._weights_dirty = False
            self._cond_key = None
        return self._handle

    def set_precision(self, mode, device=None):
        """Arithmetic of the residual layers' GEMMs (dsd_set_precision): "f32" (default, the reference's) or "bf16x3" - every
        operand split into two bf16 values, three bf16 MFMAs per fp32 one, fp32 accumulation (opt-in; WaveNet, C = 256,
        batched grids; measured 9.9e-6 off fp32 on one evaluation)."""
        modes = {"f32": 0, "fp32": 0, "bf16x3": 1}
        if mode not in modes:
            raise ValueError(f"unknown precision {mode!r}: one of {sorted(modes)}")
        if device is None:
            device = next(self.parameters()).device
        handle = self.native_handle(device)
        _lib.check(handle, _lib.lib().dsd_set_precision(handle, modes[mode]), "dsd_set_precision")
        self._cond_key = None

    def set_lengths(self, lengths, device):
        """Ragged batch (dsd_set_lengths): item b of the following calls is valid on [0, lengths[b]) and treated as zero
        padding beyond - it comes out as if it were run alone at its own length.  None: dense batches again."""
        handle = self.native_handle(device)
        stream = torch.cuda.current_stream(device).cuda_stream
        if lengths is None:
            _lib.check(handle, _lib.lib().dsd_set_lengths(handle, None, 0, C.c_void_p(stream)), "dsd_set_lengths")
            return
        vals = [int(v) for v in (lengths.tolist() if torch.is_tensor(lengths) else lengths)]
        arr = (C.c_int32 * len(vals))(*vals)
        _lib.check(handle, _lib.lib().dsd_set_lengths(handle, arr, len(vals), C.c_void_p(stream)), "dsd_set_lengths")

    def release_native(self):
        if self._handle is not None:
            _lib.lib().dsd_destroy(self._handle)
            self._handle = None
            self._cond_key = None

    def __del__(self):
        try:
            self.release_native()
        except Exception:
            pass

    # -- cond hoist ---------------------------------------------------------------------------
    def prepare_cond(self, cond, layout="BHT"):
        """Hoist every layer's conditioner_projection for this cond ([B,H,T], or [B,T,H] with layout='BTH')."""
        if cond.dtype != torch.float32:
            cond = cond.float()
        handle = self.native_handle(cond.device)
        if layout == "BHT":
            b, h, t = cond.shape
            sb, sh, st = cond.stride()
        else:
            b, t, h = cond.shape
            sb, st, sh = cond.stride()
        if h != self._hidden:
            raise ValueError(f"cond has {h} channels, expected hidden_size={self._hidden}")
        if st != 1 and sh != 1:
            return self.prepare_cond(cond.contiguous(), layout)
        key = (cond.data_ptr(), cond._version, tuple(cond.shape), tuple(cond.stride()), layout)
        if key == self._cond_key:
            return handle
        stream = torch.cuda.current_stream(cond.device).cuda_stream
        _lib.check(handle, _lib.lib().dsd_prepare_cond(handle, C.c_void_p(cond.data_ptr()), b, t, sb, sh, st,
                                                       C.c_void_p(stream)), "dsd_prepare_cond")
        self._cond_key = key
        self._cond_ref = cond          # keep the storage alive while the key is cached
        return handle

    # -- reference forward signature ----------------------------------------------------------
    def forward(self, spec, diffusion_step, cond):
        """
        :param spec: [B, F, M, T]
        :param diffusion_step: [B] or [1] (int64 or float)
        :param cond: [B, H, T]
        :return: [B, F, M, T]
        """
        if torch.is_grad_enabled() and (spec.requires_grad or cond.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError(
                "diffsinger_amd backbones are inference-only (no backward kernels): call them under "
                "torch.no_grad(); training (p_losses, ddpm.py:212-219) stays on the reference modules.")
        if spec.dim() != 4 or spec.shape[1] != self.n_feats or spec.shape[2] != self.in_dims:
            raise ValueError(f"spec must be [B, {self.n_feats}, {self.in_dims}, T], got {tuple(spec.shape)}")
        b, _, _, t_len = spec.shape
        if cond.shape[0] != b or cond.shape[2] != t_len:
            raise ValueError(f"cond {tuple(cond.shape)} does not match spec {tuple(spec.shape)}")
        if spec.numel() == 0:               # empty batch / zero frames: nothing to launch
            self.native_handle(spec.device)
            return torch.empty_like(spec, dtype=torch.float32)
        handle = self.prepare_cond(cond)
        x = spec.detach().to(torch.float32).contiguous()
        step = diffusion_step.detach().reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
        if step.numel() not in (1, b):
            raise ValueError(f"diffusion_step must have 1 or {b} elements, got {step.numel()}")
        out = torch.empty_like(x)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(handle, _lib.lib().dsd_denoise(handle, C.c_void_p(x.data_ptr()), C.c_void_p(step.data_ptr()),
                                                  step.numel(), C.c_void_p(out.data_ptr()), C.c_void_p(stream)),
                   "dsd_denoise")
        return out

    def kernel_timing(self, enable):
        """dsd_kernel_timing: while enabled, every 7th launch of each layer-kernel class carries events (graph replay is bypassed)."""
        _lib.check(self._handle, _lib.lib().dsd_kernel_timing(self._handle, 1 if enable else 0), "dsd_kernel_timing")

    def kernel_classes(self):
        """dsd_kernel_timing_classes: [{name, mean_ms, launches, evaluations, flops_per_launch, bytes_per_launch}] of the pass so far -
        which instantiations of the layer kernels actually ran (tests assert on the names)."""
        arr = (_lib.DsdKernelTime * 16)()
        n, empty = C.c_int32(), C.c_double()
        _lib.check(self._handle, _lib.lib().dsd_kernel_timing_classes(self._handle, arr, 16, C.byref(n), C.byref(empty)),
                   "dsd_kernel_timing_classes")
        return [{"name": k.name.decode(), "mean_ms": k.mean_ms, "launches": k.launches, "evaluations": k.evaluations,
                 "flops_per_launch": k.flops_per_launch, "bytes_per_launch": k.bytes_per_launch} for k in arr[:n.value]]

    def stats(self):
        st = _lib.DsdStats()
        _lib.check(self._handle, _lib.lib().dsd_get_stats(self._handle, C.byref(st)), "dsd_get_stats")
        return {n: getattr(st, n) for n, _ in st._fields_}


def sinusoidal_freqs(dim):
    """Frequency table of SinusoidalPosEmb (common_layers.py:275-276), evaluated with the same torch ops."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half) * -emb).to(torch.float32)


class _WaveNetResidualBlock(nn.Module):
    """Parameter holder with the names of wavenet.py:18-31 (never called)."""

    def __init__(self, encoder_hidden, residual_channels, dilation):
        super().__init__()
        self.dilated_conv = nn.Conv1d(residual_channels, 2 * residual_channels, kernel_size=3,
                                      padding=dilation, dilation=dilation)
        self.diffusion_projection = nn.Linear(residual_channels, residual_channels)
        self.conditioner_projection = nn.Conv1d(encoder_hidden, 2 * residual_channels, 1)
        self.output_projection = nn.Conv1d(residual_channels, 2 * residual_channels, 1)


class WaveNet(_NativeBackbone):
    """wavenet.py:51-107 on libdsdenoise."""

    def __init__(self, in_dims, n_feats, *, num_layers=20, num_channels=256, dilation_cycle_length=4):
        super().__init__()
        self.in_dims, self.n_feats = in_dims, n_feats
        self.num_layers, self.num_channels = num_layers, num_channels
        self.dilation_cycle_length = dilation_cycle_length
        self._hidden = hparams["hidden_size"]
        self.input_projection = _KaimingConv1d(in_dims * n_feats, num_channels, 1)
        self.mlp = nn.Sequential(nn.Linear(num_channels, num_channels * 4), nn.Mish(),
                                 nn.Linear(num_channels * 4, num_channels))
        self.residual_layers = nn.ModuleList([
            _WaveNetResidualBlock(self._hidden, num_channels, 2 ** (i % dilation_cycle_length))
            for i in range(num_layers)])
        self.skip_projection = _KaimingConv1d(num_channels, num_channels, 1)
        self.output_projection = _KaimingConv1d(num_channels, in_dims * n_feats, 1)
        nn.init.zeros_(self.output_projection.weight)

    def _config(self, device_index):
        return _lib.DsdConfig(C.sizeof(_lib.DsdConfig), 0, self.in_dims, self.n_feats, self.num_layers,
                              self.num_channels, self._hidden, self.dilation_cycle_length, 0, 0, 0, 0, device_index)

    def _extra_weights(self):
        return {"diffusion_embedding.freqs": sinusoidal_freqs(self.num_channels)}


class _LYNXConvModule(nn.Module):
    """Parameter holder with the `net.{0,2,4,5,6}` names of lynxnet.py:52-62 (never called)."""

    def __init__(self, dim, expansion_factor, kernel_size, activation):
        super().__init__()
        inner = dim * expansion_factor
        acts = {"SiLU": nn.SiLU, "ReLU": nn.ReLU, "PReLU": lambda: nn.PReLU(inner)}
        if activation not in acts:
            raise ValueError(f"{activation} is not a valid activation")
        pad = kernel_size // 2
        self.net = nn.Sequential(
            nn.LayerNorm(dim), nn.Identity(), nn.Conv1d(dim, inner * 2, 1), nn.Identity(),
            nn.Conv1d(inner, inner, kernel_size=kernel_size, padding=pad, groups=inner),
            acts[activation](), nn.Conv1d(inner, dim, 1), nn.Identity(), nn.Identity())


class _LYNXResidualLayer(nn.Module):
    def __init__(self, dim_cond, dim, expansion_factor, kernel_size, activation):
        super().__init__()
        self.diffusion_projection = nn.Conv1d(dim, dim, 1)
        self.conditioner_projection = nn.Conv1d(dim_cond, dim, 1)
        self.convmodule = _LYNXConvModule(dim, expansion_factor, kernel_size, activation)


class LYNXNet(_NativeBackbone):
    """lynxnet.py:90-163 on libdsdenoise."""

    def __init__(self, in_dims, n_feats, *, num_layers=6, num_channels=512, expansion_factor=2, kernel_size=31,
                 activation="PReLU", dropout=0.0, strong_cond=False):
        super().__init__()
        if float(dropout) > 0.:
            raise ValueError("dropout > 0 is a training-time feature; the HIP path is inference-only")
        if kernel_size % 2 == 0:
            raise ValueError("even kernel_size gives asymmetric padding (lynxnet.py:31-33); not supported")
        self.in_dims, self.n_feats = in_dims, n_feats
        self.num_layers, self.num_channels = num_layers, num_channels
        self.expansion_factor, self.kernel_size = expansion_factor, kernel_size
        self.activation = activation if activation is not None else "PReLU"
        self.strong_cond = strong_cond
        self._hidden = hparams["hidden_size"]
        self.input_projection = _KaimingConv1d(in_dims * n_feats, num_channels, 1)
        self.diffusion_embedding = nn.Sequential(
            nn.Identity(), nn.Linear(num_channels, num_channels * 4), nn.GELU(),
            nn.Linear(num_channels * 4, num_channels))
        self.residual_layers = nn.ModuleList([
            _LYNXResidualLayer(self._hidden, num_channels, expansion_factor, kernel_size, self.activation)
            for _ in range(num_layers)])
        self.norm = nn.LayerNorm(num_channels)
        self.output_projection = _KaimingConv1d(num_channels, in_dims * n_feats, kernel_size=1)
        nn.init.zeros_(self.output_projection.weight)

    def _config(self, device_index):
        return _lib.DsdConfig(C.sizeof(_lib.DsdConfig), 1, self.in_dims, self.n_feats, self.num_layers,
                              self.num_channels, self._hidden, 0, self.expansion_factor, self.kernel_size,
                              _lib.ACT_IDS[self.activation], int(bool(self.strong_cond)), device_index)

    def _extra_weights(self):
        return {"diffusion_embedding.freqs": sinusoidal_freqs(self.num_channels)}

    @staticmethod
    def max_positions():
        return int(1e5)


BACKBONES = {
    'wavenet': WaveNet,
    'lynxnet': LYNXNet,
}


def build_backbone(out_dims: int, num_feats: int, backbone_type: str, backbone_args: dict) -> torch.nn.Module:
    """modules/backbones/__init__.py:12-18."""
    backbone = BACKBONES[backbone_type]
    kwargs = filter_kwargs(backbone_args, backbone)
    return BACKBONES[backbone_type](out_dims, num_feats, **kwargs)


def register_into_reference():
    """Put the HIP backbones into an importable DiffSinger checkout's own registry:
    after this, `backbone_type: wavenet_hip` builds the HIP WaveNet inside the unchanged reference code."""
    import modules.backbones as ref  # noqa: WPS433
    ref.BACKBONES['wavenet_hip'] = WaveNet
    ref.BACKBONES['lynxnet_hip'] = LYNXNet
    return ref.BACKBONES
