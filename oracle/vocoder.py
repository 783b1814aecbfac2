"""numpy fp32 restatement of the NSF-HiFiGAN generator (TEST ORACLE - SURVEY.md section 8(f) rank 3, the step after
the denoise loop: mel + f0 -> waveform).

Follows modules/nsf_hifigan/models.py (weights in their inference form, i.e. after `remove_weight_norm()`):
  * SineGen._f02sine / forward            :120-168     (random initial phases and additive noise are INPUTS here)
  * SourceModuleHnNSF.forward             :200-203
  * ResBlock1.forward / ResBlock2.forward :62-69, :92-97
  * Generator.forward (mini_nsf = False)  :262-290
and the wrapper's mel scaling, modules/vocoders/nsf_hifigan.py:59-64 (log10 -> ln).
"""
from __future__ import annotations

import numpy as np

from .backbones import F32

LRELU_SLOPE = 0.1


def lrelu(x, slope):
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def conv1d(x, w, b, dilation=1, padding=0, stride=1):
    """x [B,Ci,T], w [Co,Ci,K] -> [B,Co,To]."""
    bsz, ci, t = x.shape
    co, _, k = w.shape
    xp = np.zeros((bsz, ci, t + 2 * padding), dtype=F32)
    xp[:, :, padding:padding + t] = x
    to = (t + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    y = np.zeros((bsz, co, to), dtype=F32)
    for j in range(k):
        seg = xp[:, :, j * dilation: j * dilation + (to - 1) * stride + 1: stride]
        y += np.matmul(np.ascontiguousarray(w[:, :, j]), seg)
    return (y + b[None, :, None]).astype(F32)


def conv_transpose1d(x, w, b, stride, padding):
    """x [B,Ci,T], w [Ci,Co,K] -> [B,Co,(T-1)*stride - 2*padding + K]."""
    bsz, ci, t = x.shape
    _, co, k = w.shape
    full = np.zeros((bsz, co, (t - 1) * stride + k), dtype=F32)
    for j in range(k):
        full[:, :, j: j + (t - 1) * stride + 1: stride] += np.matmul(np.ascontiguousarray(w[:, :, j]).T, x)
    out = full[:, :, padding: full.shape[2] - padding]
    return (out + b[None, :, None]).astype(F32)


def sine_source(p, f0, upp, sampling_rate, rand_ini, noise, harmonic_num=8, sine_amp=0.1, noise_std=0.003):
    """SourceModuleHnNSF: f0 [B,T] -> har_source [B,1,T*upp].  rand_ini [dim] (rand_ini[0] is forced to 0),
    noise [B, T*upp, dim] standard normals."""
    f0 = np.asarray(f0, dtype=F32)[:, :, None]
    dim = harmonic_num + 1
    n = np.arange(1, upp + 1, dtype=F32)
    rad = (f0 / F32(sampling_rate) * n).astype(F32)                               # [B,T,upp]
    rad2 = (np.fmod(rad[..., -1:] + F32(0.5), F32(1.0)) - F32(0.5)).astype(F32)
    acc = np.zeros_like(rad2)
    run = np.zeros((f0.shape[0], 1), dtype=F32)
    for t in range(f0.shape[1]):                                                  # fp32 running sum, like torch.cumsum
        run = (run + rad2[:, t]).astype(F32)
        acc[:, t] = run
    rad_acc = np.fmod(acc, F32(1.0)).astype(F32)
    rad = rad.copy()
    rad[:, 1:] += rad_acc[:, :-1]
    rad = rad.reshape(f0.shape[0], -1, 1)
    rad = (rad * np.arange(1, dim + 1, dtype=F32).reshape(1, 1, -1)).astype(F32)
    ri = np.asarray(rand_ini, dtype=F32).reshape(1, 1, dim).copy()
    ri[..., 0] = 0
    rad = (rad + ri).astype(F32)
    sines = (np.sin(F32(2 * np.pi) * rad) * F32(sine_amp)).astype(F32)
    uv = (f0 > 0).astype(F32)
    uv = np.repeat(uv, upp, axis=1)
    noise_amp = (uv * F32(noise_std) + (1 - uv) * F32(sine_amp) / 3).astype(F32)
    sine_waves = (sines * uv + noise_amp * np.asarray(noise, dtype=F32)).astype(F32)
    merged = np.tanh(sine_waves @ p["m_source.l_linear.weight"].T + p["m_source.l_linear.bias"]).astype(F32)
    return np.swapaxes(merged, 1, 2)


def fast_sine_source(f0, upp, source_sr):
    """Generator.fastsinegen (models.py:251-260, mini_nsf): f0 [B,T] -> sines [B,1,T*upp]; phase advances linearly
    interpolated between frames, no noise, no voiced/unvoiced switch."""
    f0 = np.asarray(f0, dtype=F32)
    n = np.arange(1, upp + 1, dtype=F32)
    s0 = (f0[:, :, None] / F32(source_sr)).astype(F32)
    ds0 = np.zeros_like(s0)
    ds0[:, :-1] = s0[:, 1:] - s0[:, :-1]
    rad = (s0 * n + F32(0.5) * ds0 * n * (n - 1) / F32(upp)).astype(F32)
    rad2 = (np.fmod(rad[..., -1:] + F32(0.5), F32(1.0)) - F32(0.5)).astype(F32)
    acc = np.zeros_like(rad2)
    run = np.zeros((f0.shape[0], 1), dtype=F32)
    for t in range(f0.shape[1]):
        run = (run + rad2[:, t]).astype(F32)
        acc[:, t] = run
    rad_acc = np.fmod(acc, F32(1.0)).astype(F32)
    rad = rad.copy()
    rad[:, 1:] += rad_acc[:, :-1]
    rad = rad.reshape(f0.shape[0], 1, -1)
    return np.sin(F32(2 * np.pi) * rad).astype(F32)


def resblock1(p, pre, x, kernel_size, dilations):
    for j, d in enumerate(dilations):
        xt = lrelu(x, LRELU_SLOPE)
        xt = conv1d(xt, p[f"{pre}convs1.{j}.weight"], p[f"{pre}convs1.{j}.bias"], dilation=d,
                    padding=(kernel_size * d - d) // 2)
        xt = lrelu(xt, LRELU_SLOPE)
        xt = conv1d(xt, p[f"{pre}convs2.{j}.weight"], p[f"{pre}convs2.{j}.bias"], dilation=1, padding=(kernel_size - 1) // 2)
        x = (xt + x).astype(F32)
    return x


def resblock2(p, pre, x, kernel_size, dilations):
    for j, d in enumerate(dilations):
        xt = lrelu(x, LRELU_SLOPE)
        xt = conv1d(xt, p[f"{pre}convs.{j}.weight"], p[f"{pre}convs.{j}.bias"], dilation=d, padding=(kernel_size * d - d) // 2)
        x = (xt + x).astype(F32)
    return x


def generator_forward(p, h, mel, f0, rand_ini=None, noise=None, pre_noise=None):
    """Generator.forward: mel [B, num_mels, T] (natural-log mel), f0 [B, T] -> wav [B, 1, T*prod(upsample_rates)]."""
    rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    rk, rd = list(h["resblock_kernel_sizes"]), [list(d) for d in h["resblock_dilation_sizes"]]
    mini = bool(h.get("mini_nsf", False))
    if mini:        # models.py:215-217, 264
        upp = int(np.prod(rates[:2]))
        har = fast_sine_source(f0, upp, h["sampling_rate"] / int(np.prod(rates[2:])))
    else:
        upp = int(np.prod(rates))
        har = sine_source(p, f0, upp, h["sampling_rate"], rand_ini, noise)
    x = conv1d(np.asarray(mel, dtype=F32), p["conv_pre.weight"], p["conv_pre.bias"], padding=3)
    sigma = h.get("noise_sigma", None)
    if sigma is not None and sigma > 0:                  # models.py:272-273
        x = (x + F32(sigma) * np.asarray(pre_noise, dtype=F32)).astype(F32)
    rb = resblock1 if str(h.get("resblock", "1")) == "1" else resblock2
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = lrelu(x, LRELU_SLOPE)
        x = conv_transpose1d(x, p[f"ups.{i}.weight"], p[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if mini:
            if i == 1:
                x = (x + conv1d(har, p["source_conv.weight"], p["source_conv.bias"])).astype(F32)
        else:
            if i + 1 < len(rates):
                sf = int(np.prod(rates[i + 1:]))
                xs = conv1d(har, p[f"noise_convs.{i}.weight"], p[f"noise_convs.{i}.bias"], stride=sf, padding=sf // 2)
            else:
                xs = conv1d(har, p[f"noise_convs.{i}.weight"], p[f"noise_convs.{i}.bias"])
            x = (x + xs).astype(F32)
        acc = None
        for j in range(len(rk)):
            y = rb(p, f"resblocks.{i * len(rk) + j}.", x, rk[j], rd[j])
            acc = y if acc is None else (acc + y).astype(F32)
        x = (acc / F32(len(rk))).astype(F32)
    x = lrelu(x, 0.01)                                   # F.leaky_relu default slope (models.py:287)
    x = conv1d(x, p["conv_post.weight"], p["conv_post.bias"], padding=3)
    return np.tanh(x).astype(F32)


def spec2wav(p, h, mel_btm, f0, rand_ini=None, noise=None, mel_base="10", pre_noise=None):
    """NsfHifiGAN.spec2wav_torch (vocoders/nsf_hifigan.py:54-70): mel [B,T,bins] -> wav [B*T*upp]."""
    c = np.swapaxes(np.asarray(mel_btm, dtype=F32), 1, 2)
    if mel_base != "e":
        c = (F32(2.30259) * c).astype(F32)
    return generator_forward(p, h, c, f0, rand_ini, noise, pre_noise).reshape(-1)
