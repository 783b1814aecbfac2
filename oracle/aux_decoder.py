"""numpy fp32 restatement of the shallow-diffusion aux decoder and the acoustic glue around the loop
(TEST ORACLE - SURVEY.md section 8(f) rank 1, the step immediately before the denoise loop).

Follows:
  * ConvNeXtBlock / ConvNeXtDecoder          modules/aux_decoder/convnext.py:7-85
  * AuxDecoderAdaptor (norm/denorm, forward) modules/aux_decoder/__init__.py:28-71
  * DiffSingerAcoustic.forward, infer branch modules/toplevel.py:84-105 (after `condition` is available:
    aux decoder -> padding mask -> diffusion(src_spec=aux mel) -> padding mask)
"""
from __future__ import annotations

import numpy as np

from .backbones import F32, _depthwise_conv, _gelu


def _conv1d_same(x, w, b):
    """Conv1d(Cin, Cout, k, padding=(k-1)//2), stride 1: x [B,Cin,T], w [Cout,Cin,k]."""
    bsz, cin, t = x.shape
    k = w.shape[2]
    pad = (k - 1) // 2
    xp = np.zeros((bsz, cin, t + 2 * pad), dtype=F32)
    xp[:, :, pad:pad + t] = x
    y = np.zeros((bsz, w.shape[0], t), dtype=F32)
    for j in range(k):
        y += np.matmul(np.ascontiguousarray(w[:, :, j]), xp[:, :, j:j + t])
    return (y + b[None, :, None]).astype(F32)


def _layer_norm_last(x, g, b, eps):
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = (x - mean).astype(F32)
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def convnext_num_layers(params):
    n = 0
    while f"conv.{n}.dwconv.weight" in params:
        n += 1
    return n


def convnext_decoder_forward(params, condition):
    """ConvNeXtDecoder.forward (convnext.py:78-85): condition [B,T,H] -> [B,T,out_dims]."""
    p = params
    x = np.ascontiguousarray(np.swapaxes(np.asarray(condition, dtype=F32), 1, 2))
    x = _conv1d_same(x, p["inconv.weight"], p["inconv.bias"])
    for l in range(convnext_num_layers(p)):
        pre = f"conv.{l}."
        res = x
        y = _depthwise_conv(x, p[pre + "dwconv.weight"], p[pre + "dwconv.bias"], 3)
        y = np.swapaxes(y, 1, 2)                                          # [B,T,C]
        y = _layer_norm_last(y, p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-6)
        y = (y @ p[pre + "pwconv1.weight"].T + p[pre + "pwconv1.bias"]).astype(F32)
        y = _gelu(y)
        y = (y @ p[pre + "pwconv2.weight"].T + p[pre + "pwconv2.bias"]).astype(F32)
        if (pre + "gamma") in p:
            y = (p[pre + "gamma"] * y).astype(F32)
        x = (res + np.swapaxes(y, 1, 2)).astype(F32)
    x = _conv1d_same(x, p["outconv.weight"], p["outconv.bias"])
    return np.ascontiguousarray(np.swapaxes(x, 1, 2))


def aux_adaptor_forward(params, condition, out_dims, n_feats, spec_min, spec_max, infer=True):
    """AuxDecoderAdaptor.forward (aux_decoder/__init__.py:58-71); params keys carry the `decoder.` prefix."""
    dec = {k[len("decoder."):]: v for k, v in params.items() if k.startswith("decoder.")}
    x = convnext_decoder_forward(dec, condition)                               # [B,T,F*C]
    if n_feats > 1:
        x = x.reshape(x.shape[0], x.shape[1], n_feats, out_dims)
        x = np.swapaxes(x, 1, 2)                                               # [B,F,T,C]
    if infer:
        smin = np.asarray(spec_min, dtype=F32)[None, None, ...]
        smax = np.asarray(spec_max, dtype=F32)[None, None, ...]
        smin, smax = np.swapaxes(smin, -3, -2), np.swapaxes(smax, -3, -2)
        k = ((smax - smin) / F32(2.0)).astype(F32)
        b = ((smax + smin) / F32(2.0)).astype(F32)
        x = (x * k + b).astype(F32)
    return x


def acoustic_infer(aux_params, diffusion, condition, mel2ph, noise, spec_min, spec_max, out_dims, **sampler_kw):
    """toplevel.py:84-105 after the encoder: returns (aux_mel, mel)."""
    mask = (np.asarray(mel2ph) > 0).astype(F32)[:, :, None]
    aux = aux_adaptor_forward(aux_params, condition, out_dims, 1, spec_min, spec_max, infer=True)
    aux = (aux * mask).astype(F32)
    mel = diffusion.forward(condition, noise, src_spec=aux, **sampler_kw)
    return aux, (mel * mask).astype(F32)
