"""numpy fp32 restatement of the FastSpeech2 acoustic encoder that produces the denoiser's `condition`
(TEST ORACLE - SURVEY.md section 8(f) rank 2, the producer of `cond`).

Follows (rotary-embedding configuration of the reference fork, `configs/acoustic.yaml:66` use_rope: true):
  * FastSpeech2Acoustic.forward                  modules/fastspeech/acoustic_encoder.py:82-118
  * mel2ph_to_dur                                modules/fastspeech/tts_modules.py:344-350
  * FastSpeech2Encoder.forward(_embedding)       modules/fastspeech/tts_modules.py:385-428
  * EncSALayer.forward                           modules/commons/common_layers.py:236-268
  * MultiheadSelfAttentionWithRoPE.forward       modules/commons/common_layers.py:171-213
  * TransformerFFNLayer.forward                  modules/commons/common_layers.py:142-151
  * RotaryEmbedding.rotate_queries_or_keys       modules/commons/rotary_embedding_torch.py:35-75,174-188,290-323
"""
from __future__ import annotations

import numpy as np

from .backbones import F32, _gelu

VARIANCE_ORDER = ("energy", "breathiness", "voicing", "tension")     # acoustic_encoder.py:36-43


def _ln(x, g, b, eps=1e-5):
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = (x - mean).astype(F32)
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def mel2ph_to_dur(mel2ph, t_txt):
    bsz = mel2ph.shape[0]
    dur = np.zeros((bsz, t_txt + 1), dtype=np.int64)
    for b in range(bsz):
        np.add.at(dur[b], mel2ph[b], 1)
    return dur[:, 1:]


def rope(t, freqs):
    """t: [B, heads, L, D]; rotate interleaved pairs (2i, 2i+1) by n * freqs[i]."""
    seq = np.arange(t.shape[2], dtype=F32)
    ang = (seq[:, None] * freqs[None, :].astype(F32)).astype(F32)        # [L, D/2]
    ang = np.repeat(ang, 2, axis=-1)                                      # '... n -> ... (n r)', r = 2
    x1, x2 = t[..., 0::2], t[..., 1::2]
    rot = np.empty_like(t)
    rot[..., 0::2] = -x2
    rot[..., 1::2] = x1
    return (t * np.cos(ang).astype(F32) + rot * np.sin(ang).astype(F32)).astype(F32)


def rel_positional_encoding(seq, dim, max_len=5000):
    """RelPositionalEncoding's table as the encoder uses it (espnet_positional_embedding.py:26-47,98-113): built once
    for max_len with REVERSED positions (max_len-1 ... 0), sin/cos interleaved, then sliced from the front."""
    pos = np.arange(max_len - 1, -1, -1.0, dtype=F32)[:seq, None]
    div = np.exp(np.arange(0, dim, 2, dtype=F32) * F32(-(np.log(10000.0) / dim))).astype(F32)
    ang = (pos * div).astype(F32)
    return np.stack([np.sin(ang), np.cos(ang)], axis=2).reshape(seq, dim).astype(F32)


def sinusoidal_positions(pad_mask, dim):
    """SinusoidalPositionalEmbedding(~padding_mask): positions count the non-padding tokens from 1 (utils/__init__.py:118-128),
    row p = [sin(p f_0..), cos(p f_0..)], f_i = exp(-i ln(10000) / (dim/2 - 1)); padding -> zeros."""
    nonpad = ~np.asarray(pad_mask, dtype=bool)
    pos = np.cumsum(nonpad, axis=1) * nonpad
    half = dim // 2
    freqs = np.exp(np.arange(half, dtype=F32) * F32(-(np.log(10000.0) / (half - 1)))).astype(F32)
    ang = (pos[..., None].astype(F32) * freqs).astype(F32)
    emb = np.concatenate([np.sin(ang), np.cos(ang)], axis=-1).astype(F32)
    return (emb * nonpad[..., None]).astype(F32)


def self_attention_rope(x, p, pre, num_heads, pad_mask, freqs=None):
    """MultiheadSelfAttentionWithRoPE (common_layers.py:171-213); without a rotary table - `in_proj_weight` present -
    torch.nn.MultiheadAttention(bias=False) as EncSALayer calls it (common_layers.py:222-226,247-254): the same
    arithmetic minus the rotation."""
    bsz, seq, dim = x.shape
    hd = dim // num_heads
    plain = pre + "in_proj_weight" in p
    qkv = (x @ p[pre + ("in_proj_weight" if plain else "in_proj.weight")].T).astype(F32)
    q, k, v = (qkv[..., i * dim:(i + 1) * dim].reshape(bsz, seq, num_heads, hd).transpose(0, 2, 1, 3) for i in range(3))
    if not plain:
        if freqs is None:
            freqs = p[pre + "rotary_embed.freqs"]
        q, k = rope(q, freqs), rope(k, freqs)
    scores = (np.matmul(q, k.transpose(0, 1, 3, 2)) / F32(np.sqrt(hd))).astype(F32)
    scores = np.where(pad_mask[:, None, None, :], F32(-np.inf), scores)
    scores = scores - scores.max(axis=-1, keepdims=True)
    w = np.exp(scores).astype(F32)
    w = (w / w.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    out = np.matmul(w, v).astype(F32).transpose(0, 2, 1, 3).reshape(bsz, seq, dim)
    return (out @ p[pre + "out_proj.weight"].T).astype(F32)


def ffn(x, p, pre, act="gelu"):
    """TransformerFFNLayer.forward (common_layers.py:141-151); act: 'gelu' | 'relu' | 'swish' | 'swiglu' (:126-136)."""
    w1, b1 = p[pre + "ffn_1.weight"], p[pre + "ffn_1.bias"]          # [4H, H, k]; SwiGLU: [8H, H, k]
    ks = w1.shape[2]
    pad = ks // 2
    bsz, seq, dim = x.shape
    xp = np.zeros((bsz, seq + 2 * pad, dim), dtype=F32)
    xp[:, pad:pad + seq] = x
    y = np.zeros((bsz, seq, w1.shape[0]), dtype=F32)
    for j in range(ks):
        y += xp[:, j:j + seq] @ np.ascontiguousarray(w1[:, :, j]).T
    y = ((y + b1) * F32(ks ** -0.5)).astype(F32)
    silu = lambda v: (v / (F32(1) + np.exp(-v, dtype=F32))).astype(F32)      # noqa: E731
    if act == "gelu":
        y = _gelu(y)
    elif act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "swish":
        y = silu(y)
    elif act == "swiglu":                    # out, gate = split in halves; out * silu(gate)  (common_layers.py:107-117)
        half = y.shape[-1] // 2
        y = (y[..., :half] * silu(y[..., half:])).astype(F32)
    else:
        raise ValueError(f"{act} is not a valid activation")
    return (y @ p[pre + "ffn_2.weight"].T + p[pre + "ffn_2.bias"]).astype(F32)


def fs2_encoder(p, main_embed, extra_embed, pad_mask, num_heads, prefix="encoder.", pos="rope", ffn_act="gelu"):
    """pos: 'rope' (rotation inside the attention), 'rel' (use_rope false, rel_pos true: x * sqrt(H) + table), 'sin'
    (rel_pos false: x + sinusoidal table of the non-padding positions), 'none'."""
    hidden = main_embed.shape[-1]
    nonpad = (1.0 - pad_mask.astype(F32))[:, :, None]
    x = (F32(np.sqrt(hidden)) * main_embed + extra_embed).astype(F32)
    if pos == "rel":                                     # tts_modules.py:390-392
        x = (x * F32(np.sqrt(hidden)) + rel_positional_encoding(x.shape[1], hidden)[None]).astype(F32)
    elif pos == "sin":                                   # tts_modules.py:393-395, common_layers.py:61-99
        x = (x + sinusoidal_positions(pad_mask, hidden)).astype(F32)
    x = (x * nonpad).astype(F32)
    l = 0
    while f"{prefix}layers.{l}.op.layer_norm1.weight" in p:
        pre = f"{prefix}layers.{l}.op."
        res = x
        y = _ln(x, p[pre + "layer_norm1.weight"], p[pre + "layer_norm1.bias"])
        # one RotaryEmbedding module is shared by all layers (tts_modules.py:366-373): a parameter list names it once
        freqs = p.get(pre + "self_attn.rotary_embed.freqs", p.get(f"{prefix}layers.0.op.self_attn.rotary_embed.freqs"))
        y = self_attention_rope(y, p, pre + "self_attn.", num_heads, pad_mask, freqs)
        x = ((res + y) * nonpad).astype(F32)
        res = x
        y = _ln(x, p[pre + "layer_norm2.weight"], p[pre + "layer_norm2.bias"])
        y = ffn(y, p, pre + "ffn.", ffn_act)
        x = ((res + y) * nonpad).astype(F32)
        l += 1
    return (_ln(x, p[prefix + "layer_norm.weight"], p[prefix + "layer_norm.bias"]) * nonpad).astype(F32)


def _lin1(v, p, name):
    """Linear(1, H) on a [B, T] feature."""
    return (v[:, :, None].astype(F32) * p[name + ".weight"][:, 0] + p[name + ".bias"]).astype(F32)


def fs2_acoustic_forward(p, txt_tokens, mel2ph, f0, num_heads=2, key_shift=None, speed=None, spk_embed_id=None,
                         languages=None, spk_mix_embed=None, pos="rope", ffn_act="gelu", **variances):
    """-> condition [B, T, H]."""
    txt_tokens, mel2ph = np.asarray(txt_tokens), np.asarray(mel2ph)
    txt_embed = p["txt_embed.weight"][txt_tokens]
    dur = mel2ph_to_dur(mel2ph, txt_tokens.shape[1]).astype(F32)
    extra = _lin1(dur, p, "dur_embed")
    if "lang_embed.weight" in p:
        extra = (extra + p["lang_embed.weight"][np.asarray(languages)]).astype(F32)
    enc = fs2_encoder(p, txt_embed, extra, txt_tokens == 0, num_heads, pos=pos, ffn_act=ffn_act)
    enc = np.concatenate([np.zeros_like(enc[:, :1]), enc], axis=1)
    cond = np.take_along_axis(enc, mel2ph[:, :, None].repeat(enc.shape[-1], axis=2), axis=1).astype(F32)
    if "spk_embed.weight" in p:
        if spk_mix_embed is not None:
            cond = (cond + np.asarray(spk_mix_embed, dtype=F32)).astype(F32)
        else:
            cond = (cond + p["spk_embed.weight"][np.asarray(spk_embed_id)][:, None, :]).astype(F32)
    f0_mel = np.log(F32(1) + np.asarray(f0, dtype=F32) / F32(700)).astype(F32)
    cond = (cond + _lin1(f0_mel, p, "pitch_embed")).astype(F32)
    names = [n for n in VARIANCE_ORDER if f"variance_embeds.{n}.weight" in p]
    if names:
        # torch.stack([...], dim=-1).sum(-1): summed in list order
        ve = np.zeros_like(cond)
        for n in names:
            ve = (ve + _lin1(np.asarray(variances[n], dtype=F32), p, f"variance_embeds.{n}")).astype(F32)
        cond = (cond + ve).astype(F32)
    if "key_shift_embed.weight" in p:
        cond = (cond + _lin1(np.asarray(key_shift, dtype=F32), p, "key_shift_embed")).astype(F32)
    if "speed_embed.weight" in p:
        cond = (cond + _lin1(np.asarray(speed, dtype=F32), p, "speed_embed")).astype(F32)
    return cond
