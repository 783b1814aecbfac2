"""Shallow-diffusion aux decoder on libdsdenoise (drop-in for `modules/aux_decoder`).

`AUX_DECODERS` / `build_aux_decoder` / `AuxDecoderAdaptor` keep the reference's names, arguments and behaviour
(modules/aux_decoder/__init__.py:7-71).  `ConvNeXtDecoder` holds its parameters in ordinary torch modules so that
`state_dict()` is the reference's (convnext.py:24-35,63-76: `inconv.*`, `conv.N.{gamma,dwconv,norm,pwconv1,
pwconv2}.*`, `outconv.*`) and runs ONLY on the HIP library (`dsd_aux_decode`): no CPU path, inference only.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .backbones import _NativeBackbone, filter_kwargs


class _ConvNeXtBlock(nn.Module):
    """Parameter holder with the names of convnext.py:24-35 (never called)."""

    def __init__(self, dim, intermediate_dim, layer_scale_init_value):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, intermediate_dim)
        self.pwconv2 = nn.Linear(intermediate_dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim), requires_grad=True)


class ConvNeXtDecoder(_NativeBackbone):
    """convnext.py:58-85 on libdsdenoise."""

    def __init__(self, in_dims, out_dims, /, *, num_channels=512, num_layers=6, kernel_size=7, dropout_rate=0.1):
        super().__init__()
        if kernel_size % 2 == 0:
            raise ValueError("even kernel_size changes the sequence length (convnext.py:65); not supported")
        self.in_dims, self.out_dims = in_dims, out_dims
        self.num_channels, self.num_layers, self.kernel_size = num_channels, num_layers, kernel_size
        self._hidden = in_dims
        pad = (kernel_size - 1) // 2
        self.inconv = nn.Conv1d(in_dims, num_channels, kernel_size, stride=1, padding=pad)
        self.conv = nn.ModuleList(_ConvNeXtBlock(num_channels, num_channels * 4, 1e-6) for _ in range(num_layers))
        self.outconv = nn.Conv1d(num_channels, out_dims, kernel_size, stride=1, padding=pad)

    def _config(self, device_index):
        return _lib.DsdConfig(C.sizeof(_lib.DsdConfig), _lib.AUX_CONVNEXT, self.out_dims, 1, self.num_layers,
                              self.num_channels, self.in_dims, 0, 0, self.kernel_size, 0, 0, device_index)

    def prepare_cond(self, cond, layout="BHT"):
        raise RuntimeError("ConvNeXtDecoder has no hoisted conditioner; call forward(condition)")

    # noinspection PyUnusedLocal
    def forward(self, x, infer=False, *, out_scale=None, out_shift=None, lengths=None):
        """x: [B, T, in_dims] -> [B, T, out_dims] (convnext.py:78-85); `out_scale/out_shift` ([out_dims]) fuse
        AuxDecoderAdaptor.denorm_spec into the output transpose; `lengths` [B]: ragged batch (dsd_set_lengths)."""
        if lengths is not None:
            self.set_lengths(lengths, x.device)
            try:
                return self.forward(x, infer, out_scale=out_scale, out_shift=out_shift)
            finally:
                self.set_lengths(None, x.device)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError(
                "diffsinger_amd.ConvNeXtDecoder is inference-only (no backward kernels): call it under "
                "torch.no_grad(); aux-decoder training (toplevel.py:107-113) stays on the reference module.")
        if x.dim() != 3 or x.shape[2] != self.in_dims:
            raise ValueError(f"condition must be [B, T, {self.in_dims}], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32)
        handle = self.native_handle(x.device)
        b, t, _ = x.shape
        sb, st, sh = x.stride()
        if st != 1 and sh != 1:
            x = x.contiguous()
            sb, st, sh = x.stride()
        out = torch.empty((b, t, self.out_dims), device=x.device, dtype=torch.float32)
        if b == 0 or t == 0:                # empty batch / zero frames: nothing to launch
            return out
        ptr = lambda v: C.c_void_p(0 if v is None else v.data_ptr())  # noqa: E731
        if out_scale is not None:
            out_scale = out_scale.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            out_shift = out_shift.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            if out_scale.numel() != self.out_dims or out_shift.numel() != self.out_dims:
                raise ValueError(f"out_scale/out_shift must have {self.out_dims} elements")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(handle, _lib.lib().dsd_aux_decode(handle, C.c_void_p(x.data_ptr()), b, t, sb, sh, st,
                                                     C.c_void_p(out.data_ptr()), ptr(out_scale), ptr(out_shift),
                                                     C.c_void_p(stream)), "dsd_aux_decode")
        return out


AUX_DECODERS = {
    'convnext': ConvNeXtDecoder
}
AUX_LOSSES = {
    'convnext': nn.L1Loss
}


def build_aux_decoder(in_dims: int, out_dims: int, aux_decoder_arch: str, aux_decoder_args: dict) -> torch.nn.Module:
    """modules/aux_decoder/__init__.py:15-21."""
    decoder_cls = AUX_DECODERS[aux_decoder_arch]
    kwargs = filter_kwargs(aux_decoder_args, decoder_cls)
    return AUX_DECODERS[aux_decoder_arch](in_dims, out_dims, **kwargs)


def build_aux_loss(aux_decoder_arch):
    return AUX_LOSSES[aux_decoder_arch]()


class AuxDecoderAdaptor(nn.Module):
    """modules/aux_decoder/__init__.py:28-71."""

    def __init__(self, in_dims: int, out_dims: int, num_feats: int, spec_min: list, spec_max: list,
                 aux_decoder_arch: str, aux_decoder_args: dict):
        super().__init__()
        self.decoder = build_aux_decoder(in_dims=in_dims, out_dims=out_dims * num_feats,
                                         aux_decoder_arch=aux_decoder_arch, aux_decoder_args=aux_decoder_args)
        self.out_dims = out_dims
        self.n_feats = num_feats
        if spec_min is not None and spec_max is not None:
            # spec: [B, T, M] or [B, F, T, M]; buffers [1, 1, M] or [1, F, 1, M]  (:40-46)
            spec_min = torch.FloatTensor(spec_min)[None, None, :].transpose(-3, -2)
            spec_max = torch.FloatTensor(spec_max)[None, None, :].transpose(-3, -2)
            self.register_buffer('spec_min', spec_min, persistent=False)
            self.register_buffer('spec_max', spec_max, persistent=False)

    def _affine(self):
        """(k, b) of the spec <-> [-1, 1] map: half range and mid point of [spec_min, spec_max]."""
        return (self.spec_max - self.spec_min) / 2., (self.spec_max + self.spec_min) / 2.

    def norm_spec(self, x):
        k, b = self._affine()
        return (x - b) / k

    def denorm_spec(self, x):
        k, b = self._affine()
        return x * k + b

    def forward(self, condition, infer=False, lengths=None):
        fuse = infer and self.n_feats == 1 and self.spec_min.numel() in (1, self.out_dims)
        if fuse:        # x * k + b rides on the decoder's output transpose
            k, b = (v.reshape(-1).expand(self.out_dims) for v in self._affine())
            return self.decoder(condition, infer=True, out_scale=k, out_shift=b, lengths=lengths)
        x = self.decoder(condition, infer=infer, lengths=lengths)  # [B, T, F x C]
        if self.n_feats > 1:
            x = x.reshape(-1, x.shape[1], self.n_feats, self.out_dims)  # [B, T, F, C]
            x = x.transpose(1, 2)  # [B, F, T, C]
        if infer:
            x = self.denorm_spec(x)
        return x  # [B, T, C] or [B, F, T, C]
