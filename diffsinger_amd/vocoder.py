"""NSF-HiFiGAN vocoder on libdsdenoise (drop-in for `modules/nsf_hifigan/models.py:Generator` and the
`modules/vocoders/nsf_hifigan.py:NsfHifiGAN` wrapper's `spec2wav_torch`).

`Generator(h)` takes the checkpoint's config (dict / AttrDict, the fields models.py:207-260 reads) and holds its
parameters under the reference's names in their inference form (after `remove_weight_norm()`); a checkpoint that
still carries `weight_g` / `weight_v` pairs is folded on load (`w = g * v / ||v||`, the norm over all dims but 0,
torch.nn.utils.weight_norm's default).  `forward(x, f0)` = models.py:262-290 with SineGen's two random draws
(`torch.rand` initial phases, `torch.randn_like` noise) made on the caller's device, or passed in for
reproducibility; `mini_nsf: true` generators (fastsinegen source, `source_conv`) have no random draws.
Inference only; no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .backbones import _NativeBackbone


def _get(h, key, default=None):
    v = h.get(key, default) if hasattr(h, "get") else getattr(h, key, default)
    return default if v is None else v


class _ResBlock(nn.Module):
    def __init__(self, kind, ch, k, dils):
        super().__init__()
        mk = lambda d: nn.Conv1d(ch, ch, k, 1, dilation=d, padding=(k * d - d) // 2)   # noqa: E731
        if kind == 1:
            self.convs1 = nn.ModuleList([mk(d) for d in dils])
            self.convs2 = nn.ModuleList([mk(1) for _ in dils])
        else:
            self.convs = nn.ModuleList([mk(d) for d in dils])


class _Source(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.l_linear = nn.Linear(dim, 1)


class Generator(_NativeBackbone):
    def __init__(self, h):
        super().__init__()
        self.mini_nsf = bool(_get(h, "mini_nsf", False))
        self.h = h
        self.num_mels = int(_get(h, "num_mels"))
        self.sampling_rate = int(_get(h, "sampling_rate"))
        self.upsample_rates = [int(v) for v in _get(h, "upsample_rates")]
        self.upsample_kernel_sizes = [int(v) for v in _get(h, "upsample_kernel_sizes")]
        self.upsample_initial_channel = int(_get(h, "upsample_initial_channel"))
        self.resblock = 1 if str(_get(h, "resblock", "1")) == "1" else 2
        self.resblock_kernel_sizes = [int(v) for v in _get(h, "resblock_kernel_sizes")]
        self.resblock_dilation_sizes = [[int(d) for d in ds] for ds in _get(h, "resblock_dilation_sizes")]
        self.noise_sigma = _get(h, "noise_sigma", None)          # models.py:213: > 0 adds sigma * randn after conv_pre
        self.harmonic_num = 8
        self.upp = int(np.prod(self.upsample_rates))
        self._hidden = self.num_mels
        if not self.mini_nsf:
            self.m_source = _Source(self.harmonic_num + 1)
            self.noise_convs = nn.ModuleList()
        self.conv_pre = nn.Conv1d(self.num_mels, self.upsample_initial_channel, 7, 1, padding=3)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        ch = self.upsample_initial_channel
        for i, (u, k) in enumerate(zip(self.upsample_rates, self.upsample_kernel_sizes)):
            ch //= 2
            self.ups.append(nn.ConvTranspose1d(2 * ch, ch, k, u, padding=(k - u) // 2))
            for rk, rd in zip(self.resblock_kernel_sizes, self.resblock_dilation_sizes):
                self.resblocks.append(_ResBlock(self.resblock, ch, rk, rd))
            if self.mini_nsf:
                if i == 1:
                    self.source_conv = nn.Conv1d(1, ch, 1)
            elif i + 1 < len(self.upsample_rates):
                sf = int(np.prod(self.upsample_rates[i + 1:]))
                self.noise_convs.append(nn.Conv1d(1, ch, kernel_size=sf * 2, stride=sf, padding=sf // 2))
            else:
                self.noise_convs.append(nn.Conv1d(1, ch, kernel_size=1))
        self.conv_post = nn.Conv1d(ch, 1, 7, 1, padding=3)
        self._register_load_state_dict_pre_hook(self._fold_weight_norm)

    @staticmethod
    def _fold_weight_norm(state_dict, prefix, *args):
        """Checkpoints saved before remove_weight_norm(): `X.weight_g`, `X.weight_v`  ->  `X.weight`."""
        for key in [k for k in state_dict if k.startswith(prefix) and k.endswith(".weight_g")]:
            base = key[:-len("weight_g")]
            g, v = state_dict.pop(key), state_dict.pop(base + "weight_v")
            norm = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            state_dict[base + "weight"] = v * (g / norm)

    def remove_weight_norm(self):           # the parameters are already stored without weight norm
        pass

    def _config(self, device_index):
        cfg = _lib.DsdVocoderConfig()
        cfg.struct_size = C.sizeof(_lib.DsdVocoderConfig)
        cfg.num_mels, cfg.sampling_rate = self.num_mels, self.sampling_rate
        cfg.upsample_initial_channel = self.upsample_initial_channel
        cfg.n_ups = len(self.upsample_rates)
        for i, (u, k) in enumerate(zip(self.upsample_rates, self.upsample_kernel_sizes)):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.resblock = self.resblock
        cfg.n_kernels = len(self.resblock_kernel_sizes)
        for j, (rk, rd) in enumerate(zip(self.resblock_kernel_sizes, self.resblock_dilation_sizes)):
            cfg.resblock_kernel_sizes[j] = rk
            cfg.n_dilations[j] = len(rd)
            for d, dv in enumerate(rd):
                cfg.resblock_dilation_sizes[j][d] = dv
        cfg.harmonic_num = self.harmonic_num
        cfg.mini_nsf = int(self.mini_nsf)
        cfg.noise_sigma = float(self.noise_sigma) if self.noise_sigma is not None and self.noise_sigma > 0 else 0.0
        cfg.device = device_index
        return cfg

    def prepare_cond(self, cond, layout="BHT"):
        raise RuntimeError("the vocoder has no conditioner; call forward(mel, f0)")

    def forward(self, x, f0, *, rand_ini=None, noise=None, pre_noise=None):
        """x: [B, num_mels, T] natural-log mel, f0: [B, T] -> [B, 1, T * prod(upsample_rates)]."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("diffsinger_amd.vocoder.Generator is inference-only: call it under torch.no_grad()")
        if x.dim() != 3 or x.shape[1] != self.num_mels or tuple(f0.shape) != (x.shape[0], x.shape[2]):
            raise ValueError(f"x [B, {self.num_mels}, T] and f0 [B, T] expected; got {tuple(x.shape)}, {tuple(f0.shape)}")
        dev = x.device
        handle = self.native_handle(dev)
        b, _, t = x.shape
        out = torch.empty((b, 1, t * self.upp), device=dev, dtype=torch.float32)
        if b == 0 or t == 0:
            return out
        x = x.detach().to(torch.float32)
        sb, sm, st_ = x.stride()
        if st_ != 1 and sm != 1:
            x = x.contiguous()
            sb, sm, st_ = x.stride()
        f0 = f0.detach().to(device=dev, dtype=torch.float32).contiguous()
        dim = self.harmonic_num + 1
        want_pre = self.noise_sigma is not None and self.noise_sigma > 0      # models.py:272-273
        # the reference's order of random draws (same generator state in, same state out): SineGen's initial phases
        # (models.py:145), its additive noise (:165), and only then - after conv_pre - the noise_sigma normals (:272-273)
        if not self.mini_nsf:
            if rand_ini is None:
                rand_ini = torch.rand(dim, device=dev)
            if noise is None:
                noise = torch.randn((b, t * self.upp, dim), device=dev)
        pre_ptr = None
        if want_pre:
            c0 = self.upsample_initial_channel
            if pre_noise is None:
                pre_noise = torch.randn((b, c0, t), device=dev)
            pre_noise = pre_noise.detach().to(device=dev, dtype=torch.float32).contiguous()
            if tuple(pre_noise.shape) != (b, c0, t):
                raise ValueError(f"pre_noise [{b}, {c0}, {t}] expected")
            pre_ptr = C.c_void_p(pre_noise.data_ptr())
        if self.mini_nsf:           # deterministic source (models.py:251-260): nothing else to draw
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(handle, _lib.lib().dsd_vocode(handle, C.c_void_p(x.data_ptr()), b, t, sb, sm, st_,
                                                     C.c_void_p(f0.data_ptr()), None, None, pre_ptr,
                                                     C.c_void_p(out.data_ptr()), C.c_void_p(stream)), "dsd_vocode")
            return out
        rand_ini = rand_ini.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        noise = noise.detach().to(device=dev, dtype=torch.float32).contiguous()
        if rand_ini.numel() != dim or tuple(noise.shape) != (b, t * self.upp, dim):
            raise ValueError(f"rand_ini [{dim}] and noise [{b}, {t * self.upp}, {dim}] expected")
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(handle, _lib.lib().dsd_vocode(handle, C.c_void_p(x.data_ptr()), b, t, sb, sm, st_,
                                                 C.c_void_p(f0.data_ptr()), C.c_void_p(rand_ini.data_ptr()),
                                                 C.c_void_p(noise.data_ptr()), pre_ptr, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(stream)), "dsd_vocode")
        return out


class NsfHifiGAN:
    """The `spec2wav_torch` half of modules/vocoders/nsf_hifigan.py:16-70 around a `Generator` (the checkpoint /
    config loading of `load_model`, models.py:18-33, is the caller's: pass the built generator and its config)."""

    def __init__(self, generator: Generator, mel_base="10"):
        self.model, self.h, self.mel_base = generator, generator.h, mel_base

    def spec2wav_torch(self, mel, **kwargs):        # mel: [B, T, bins]
        with torch.no_grad():
            c = mel.transpose(2, 1)
            if self.mel_base != 'e':
                assert self.mel_base in [10, '10'], "mel_base must be 'e', '10' or 10."
                c = 2.30259 * c                      # log10 to log mel (nsf_hifigan.py:62-64)
            f0 = kwargs.get('f0')
            if f0 is None:
                raise ValueError("the NSF generator needs f0 (models.py:262)")
            extra = {k: kwargs[k] for k in ("rand_ini", "noise", "pre_noise") if kwargs.get(k) is not None}
            return self.model(c, f0, **extra).view(-1)
