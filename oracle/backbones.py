"""numpy fp32 restatement of the reference denoiser backbones (TEST ORACLE).

Follows, function by function:
  * SinusoidalPosEmb           modules/commons/common_layers.py:268-280
  * SwiGLU                     modules/commons/common_layers.py:107-117
  * ResidualBlock / WaveNet    modules/backbones/wavenet.py:18-48, 51-107
  * LYNXConvModule / LYNXNetResidualLayer / LYNXNet
                               modules/backbones/lynxnet.py:29-65, 68-87, 90-163

Parameters are a flat dict {state_dict name: float32 ndarray} with exactly the
`nn.Module.state_dict()` names of the reference modules.  Activations are
float32 ndarrays laid out as the reference lays them out ([B, C, T]).
Every intermediate stays float32 (numpy 2 weak-scalar promotion).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32


# ----------------------------------------------------------------------------
# elementwise pieces
# ----------------------------------------------------------------------------
def _sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def _mish(x):
    # nn.Mish: x * tanh(softplus(x))                      (wavenet.py:60)
    with np.errstate(over="ignore"):
        sp = np.log1p(np.exp(x))
    return (x * np.tanh(sp)).astype(F32)


def _gelu(x):
    # nn.GELU() default = exact erf form                   (lynxnet.py:106,143)
    return (x * F32(0.5) * (F32(1.0) + _erf(x * F32(1.0 / math.sqrt(2.0))))).astype(F32)


def _silu(x):
    return (x * _sigmoid(x)).astype(F32)


def sinusoidal_pos_emb(t, dim):
    """common_layers.py:273-280.  t: [B] (int or float) -> [B, dim] float32."""
    t = np.asarray(t)
    half = dim // 2
    step = math.log(10000) / (half - 1)
    freq = np.exp(np.arange(half, dtype=F32) * F32(-step)).astype(F32)
    arg = (t.astype(F32)[:, None] * freq[None, :]).astype(F32)
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1).astype(F32)


def _linear(x, w, b):
    # nn.Linear on [B, in] -> [B, out]
    return (x @ w.T + b).astype(F32)


def _conv1x1(x, w, b):
    # Conv1d(k=1) on [B, Cin, T]; w [Cout, Cin, 1]
    return (np.matmul(w[:, :, 0], x) + b[None, :, None]).astype(F32)


def _dilated_conv3(x, w, b, dil):
    # Conv1d(k=3, dilation=dil, padding=dil): cross-correlation, zero padding
    bsz, cin, t = x.shape
    xp = np.zeros((bsz, cin, t + 2 * dil), dtype=F32)
    xp[:, :, dil:dil + t] = x
    y = np.zeros((bsz, w.shape[0], t), dtype=F32)
    for k in range(3):
        # contiguous tap matrix: a stride-3 view falls off numpy's BLAS path (170x slower)
        y += np.matmul(np.ascontiguousarray(w[:, :, k]), xp[:, :, k * dil:k * dil + t])
    return (y + b[None, :, None]).astype(F32)


def _depthwise_conv(x, w, b, pad):
    # Conv1d(C, C, k, padding=pad, groups=C); w [C, 1, k]
    bsz, c, t = x.shape
    k = w.shape[2]
    xp = np.zeros((bsz, c, t + 2 * pad), dtype=F32)
    xp[:, :, pad:pad + t] = x
    y = np.zeros_like(x)
    for j in range(k):
        y += w[None, :, 0, j, None] * xp[:, :, j:j + t]
    return (y + b[None, :, None]).astype(F32)


def _layer_norm_channels(x, g, b, eps=1e-5):
    # nn.LayerNorm(C) applied on the channel axis of [B, C, T]
    mean = x.mean(axis=1, keepdims=True, dtype=F32)
    xc = (x - mean).astype(F32)
    var = (xc * xc).mean(axis=1, keepdims=True, dtype=F32)
    inv = (F32(1.0) / np.sqrt(var + F32(eps))).astype(F32)
    return (xc * inv * g[None, :, None] + b[None, :, None]).astype(F32)


def _flatten_spec(spec):
    # wavenet.py:82-85 / lynxnet.py:136-139:  [B, F, M, T] -> [B, F*M, T]
    b, f, m, t = spec.shape
    return np.ascontiguousarray(spec.reshape(b, f * m, t), dtype=F32)


def _broadcast_step(t, bsz):
    t = np.asarray(t).reshape(-1)
    if t.shape[0] == 1 and bsz > 1:       # reflow passes a [1] tensor (reflow.py:135)
        t = np.repeat(t, bsz)
    return t


# ----------------------------------------------------------------------------
# WaveNet
# ----------------------------------------------------------------------------
def wavenet_config(params):
    """Recover (C, L, in_dims*n_feats, H) from parameter shapes."""
    c, m, _ = params["input_projection.weight"].shape
    n_layers = 0
    while f"residual_layers.{n_layers}.dilated_conv.weight" in params:
        n_layers += 1
    h = params["residual_layers.0.conditioner_projection.weight"].shape[1]
    return c, n_layers, m, h


def wavenet_forward(params, spec, diffusion_step, cond, dilation_cycle_length=4,
                    return_intermediates=False):
    """WaveNet.forward (wavenet.py:75-107).

    spec [B, F, M, T], diffusion_step [B] or [1], cond [B, H, T] -> [B, F, M, T].
    """
    p = params
    spec = np.asarray(spec, dtype=F32)
    cond = np.asarray(cond, dtype=F32)
    bsz, n_feats, in_dims, t_len = spec.shape
    c, n_layers, _, _ = wavenet_config(p)

    x = _flatten_spec(spec)
    x = _conv1x1(x, p["input_projection.weight"], p["input_projection.bias"])
    x = np.maximum(x, F32(0))
    step = _broadcast_step(diffusion_step, bsz)
    e = sinusoidal_pos_emb(step, c)
    e = _linear(e, p["mlp.0.weight"], p["mlp.0.bias"])
    e = _mish(e)
    e = _linear(e, p["mlp.2.weight"], p["mlp.2.bias"])

    inter = {}
    skip_sum = np.zeros((bsz, c, t_len), dtype=F32)
    skips = []
    inv_sqrt2 = F32(1.0 / math.sqrt(2.0))
    for l in range(n_layers):
        pre = f"residual_layers.{l}."
        dil = 2 ** (l % dilation_cycle_length)
        d = _linear(e, p[pre + "diffusion_projection.weight"], p[pre + "diffusion_projection.bias"])
        cproj = _conv1x1(cond, p[pre + "conditioner_projection.weight"],
                         p[pre + "conditioner_projection.bias"])
        y = (x + d[:, :, None]).astype(F32)
        y = _dilated_conv3(y, p[pre + "dilated_conv.weight"], p[pre + "dilated_conv.bias"], dil)
        y = (y + cproj).astype(F32)
        gate, filt = y[:, :c], y[:, c:]
        y = (_sigmoid(gate) * np.tanh(filt)).astype(F32)
        y = _conv1x1(y, p[pre + "output_projection.weight"], p[pre + "output_projection.bias"])
        residual, skip = y[:, :c], y[:, c:]
        # reference divides by math.sqrt(2.0); torch turns that into a multiply by the
        # reciprocal only in some builds - keep the division, it is what is written.
        x = ((x + residual) / F32(math.sqrt(2.0))).astype(F32)
        skips.append(skip)
        if return_intermediates:
            inter[f"x_after_{l}"] = x.copy()
            inter[f"skip_{l}"] = skip.copy()
    # torch.sum(torch.stack(skip), dim=0): sequential-ish reduction over L
    for s in skips:
        skip_sum += s
    x = (skip_sum / F32(math.sqrt(n_layers))).astype(F32)
    x = _conv1x1(x, p["skip_projection.weight"], p["skip_projection.bias"])
    x = np.maximum(x, F32(0))
    x = _conv1x1(x, p["output_projection.weight"], p["output_projection.bias"])
    out = x.reshape(bsz, n_feats, in_dims, t_len)
    del inv_sqrt2
    if return_intermediates:
        return out, inter
    return out


# ----------------------------------------------------------------------------
# LYNXNet
# ----------------------------------------------------------------------------
def lynxnet_config(params):
    c, m, _ = params["input_projection.weight"].shape
    n_layers = 0
    while f"residual_layers.{n_layers}.conditioner_projection.weight" in params:
        n_layers += 1
    inner = params["residual_layers.0.convmodule.net.4.weight"].shape[0]
    ksz = params["residual_layers.0.convmodule.net.4.weight"].shape[2]
    h = params["residual_layers.0.conditioner_projection.weight"].shape[1]
    return c, n_layers, m, h, inner, ksz


def _lynx_activation(x, params, pre, activation):
    if activation == "PReLU":
        w = params[pre + "convmodule.net.5.weight"]
        return np.where(x >= 0, x, x * w[None, :, None]).astype(F32)
    if activation == "SiLU":
        return _silu(x)
    if activation == "ReLU":
        return np.maximum(x, F32(0))
    raise ValueError(f"{activation} is not a valid activation")      # lynxnet.py:44-45


def lynxnet_forward(params, spec, diffusion_step, cond, activation="PReLU", strong_cond=False):
    """LYNXNet.forward (lynxnet.py:128-163), residual layer :76-87, conv module :52-62."""
    p = params
    spec = np.asarray(spec, dtype=F32)
    cond = np.asarray(cond, dtype=F32)
    bsz, n_feats, in_dims, t_len = spec.shape
    c, n_layers, _, _, inner, ksz = lynxnet_config(p)
    pad = ksz // 2

    x = _flatten_spec(spec)
    x = _conv1x1(x, p["input_projection.weight"], p["input_projection.bias"])
    if not strong_cond:
        x = _gelu(x)
    step = _broadcast_step(diffusion_step, bsz)
    e = sinusoidal_pos_emb(step, c)
    e = _linear(e, p["diffusion_embedding.1.weight"], p["diffusion_embedding.1.bias"])
    e = _gelu(e)
    e = _linear(e, p["diffusion_embedding.3.weight"], p["diffusion_embedding.3.bias"])

    for l in range(n_layers):
        pre = f"residual_layers.{l}."
        cproj = _conv1x1(cond, p[pre + "conditioner_projection.weight"],
                         p[pre + "conditioner_projection.bias"])
        if strong_cond:
            x = (x + cproj).astype(F32)
            res = x
        else:
            res = x
            x = (x + cproj).astype(F32)
        # Conv1d(C, C, 1) on the [B, C, 1] step embedding == a Linear
        dproj = _linear(e, p[pre + "diffusion_projection.weight"][:, :, 0],
                        p[pre + "diffusion_projection.bias"])
        x = (x + dproj[:, :, None]).astype(F32)
        y = _layer_norm_channels(x, p[pre + "convmodule.net.0.weight"], p[pre + "convmodule.net.0.bias"])
        y = _conv1x1(y, p[pre + "convmodule.net.2.weight"], p[pre + "convmodule.net.2.bias"])
        out_half, gate_half = y[:, :inner], y[:, inner:]          # SwiGLU: out * silu(gate)
        y = (out_half * _silu(gate_half)).astype(F32)
        y = _depthwise_conv(y, p[pre + "convmodule.net.4.weight"], p[pre + "convmodule.net.4.bias"], pad)
        y = _lynx_activation(y, p, pre, activation)
        y = _conv1x1(y, p[pre + "convmodule.net.6.weight"], p[pre + "convmodule.net.6.bias"])
        x = (y + res).astype(F32)

    x = _layer_norm_channels(x, p["norm.weight"], p["norm.bias"])
    x = _conv1x1(x, p["output_projection.weight"], p["output_projection.bias"])
    return x.reshape(bsz, n_feats, in_dims, t_len)


def backbone_forward(kind, params, spec, diffusion_step, cond, **kw):
    if kind == "wavenet":
        return wavenet_forward(params, spec, diffusion_step, cond,
                               dilation_cycle_length=kw.get("dilation_cycle_length", 4))
    if kind == "lynxnet":
        return lynxnet_forward(params, spec, diffusion_step, cond,
                               activation=kw.get("activation", "PReLU"),
                               strong_cond=kw.get("strong_cond", False))
    raise KeyError(kind)
