"""FastSpeech2 acoustic encoder on libdsdenoise (drop-in for `modules/fastspeech/acoustic_encoder.py`).

`FastSpeech2Acoustic(vocab_size)` reads the same hparams as the reference (acoustic_encoder.py:15-63), holds its
parameters in ordinary torch modules under the reference's names - `txt_embed.weight`, `dur_embed.*`,
`encoder.layers.N.op.{layer_norm1, self_attn.{in_proj,out_proj}.weight, self_attn.rotary_embed.freqs, layer_norm2,
ffn.ffn_1, ffn.ffn_2}.*`, `encoder.layer_norm.*`, `pitch_embed.*`, ... - so the `fs2.` slice of an acoustic
checkpoint loads with strict=True, and runs `forward(txt_tokens, mel2ph, f0, ...) -> condition [B, T, H]` ONLY on
the HIP library (`dsd_encode`).  Supported: the reference fork's configuration (`use_pos_embed: true`,
`use_rope: true`, `ffn_act: gelu`) and the two pre-rotary layouts - `use_rope: false` with `rel_pos: true`
(RelPositionalEncoding + torch.nn.MultiheadAttention parameter names), `rel_pos: false` (SinusoidalPositionalEmbedding
over the non-padding positions) or `use_pos_embed: false`.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from .backbones import _NativeBackbone
from .hparams import hparams

PAD_INDEX = 0       # utils/phoneme_utils.py:7
VARIANCE_ORDER = ("energy", "breathiness", "voicing", "tension")


class _Rotary(nn.Module):
    """Holder of RotaryEmbedding's `freqs` parameter (rotary_embedding_torch.py:119,131)."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1. / (theta ** (torch.arange(0, dim, 2)[:(dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)


class _SelfAttn(nn.Module):
    def __init__(self, h, rotary):
        super().__init__()
        self.in_proj = nn.Linear(h, 3 * h, bias=False)
        self.out_proj = nn.Linear(h, h, bias=False)
        self.rotary_embed = rotary


class _SelfAttnPlain(nn.Module):
    """Parameter names of torch.nn.MultiheadAttention(h, heads, bias=False) (common_layers.py:222-226)."""

    def __init__(self, h):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * h, h))
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.out_proj = nn.Linear(h, h, bias=False)


def pos_mode_of(get):
    """DSD_POS_* from `use_pos_embed` / `use_rope` / `rel_pos` as FastSpeech2Encoder.__init__ reads them
    (tts_modules.py:362-364,378-384)."""
    use_pos = get('use_pos_embed') if get('use_pos_embed') is not None else True
    if use_pos and get('use_rope'):
        return _lib.POS_ROPE
    if not use_pos:
        return _lib.POS_NONE
    return _lib.POS_REL if get('rel_pos') else _lib.POS_SIN


def ffn_act_of(get):
    """DSD_FFN_* from `ffn_act`; the reference's error for anything else (common_layers.py:135-136)."""
    act = get('ffn_act') or 'gelu'
    if act not in _lib.FFN_ACTS:
        raise ValueError(f'{act} is not a valid activation')
    return _lib.FFN_ACTS[act]


def rel_pos_div_term(h):
    """RelPositionalEncoding's frequency table, with the reference's own torch ops (espnet_positional_embedding.py:38-41)."""
    return torch.exp(torch.arange(0, h, 2, dtype=torch.float32) * -(math.log(10000.0) / h))


def sin_pos_freqs(h):
    """SinusoidalPositionalEmbedding.get_embedding's frequency table, with its torch ops (common_layers.py:69-71)."""
    half = h // 2
    return torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))


def positional_extra_weights(pos_mode, h):
    if pos_mode == _lib.POS_REL:
        return {"encoder.embed_positions.div_term": rel_pos_div_term(h)}
    if pos_mode == _lib.POS_SIN:
        return {"encoder.embed_positions.freqs": sin_pos_freqs(h)}
    return {}


class _SinPos(nn.Module):
    """SinusoidalPositionalEmbedding's only state-dict entry: the `_float_tensor` marker buffer (common_layers.py:59)."""

    def __init__(self):
        super().__init__()
        self.register_buffer('_float_tensor', torch.zeros(1))


class _FFN(nn.Module):
    def __init__(self, h, ks, ffn_act=0):
        super().__init__()
        # SwiGLU: ffn_1 produces out and gate, 2 * filter_size channels (common_layers.py:132-134)
        self.ffn_1 = nn.Conv1d(h, (8 if ffn_act == _lib.FFN_ACTS["swiglu"] else 4) * h, ks, padding=ks // 2)
        self.ffn_2 = nn.Linear(4 * h, h)


class _EncSALayer(nn.Module):
    def __init__(self, h, ks, rotary, ffn_act=0):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(h)
        self.self_attn = _SelfAttn(h, rotary) if rotary is not None else _SelfAttnPlain(h)
        self.layer_norm2 = nn.LayerNorm(h)
        self.ffn = _FFN(h, ks, ffn_act)


class _Layer(nn.Module):
    def __init__(self, h, ks, rotary, ffn_act=0):
        super().__init__()
        self.op = _EncSALayer(h, ks, rotary, ffn_act)


class _Encoder(nn.Module):
    """Parameter holder with the names of FastSpeech2Encoder (tts_modules.py:353-383); never called."""

    def __init__(self, h, layers, heads, ks, pos_mode=0, ffn_act=0):
        super().__init__()
        rotary = _Rotary(h // heads) if pos_mode == _lib.POS_ROPE else None
        self.layers = nn.ModuleList([_Layer(h, ks, rotary, ffn_act) for _ in range(layers)])
        self.layer_norm = nn.LayerNorm(h)
        if pos_mode == _lib.POS_SIN:
            self.embed_positions = _SinPos()


class FastSpeech2Acoustic(_NativeBackbone):
    def __init__(self, vocab_size):
        super().__init__()
        hp = hparams
        self.pos_mode = pos_mode_of(hp.get)
        self.ffn_act = ffn_act_of(hp.get)
        h = self._hidden = hp['hidden_size']
        self.vocab_size = vocab_size
        self.enc_layers, self.num_heads = hp['enc_layers'], hp['num_heads']
        self.ffn_kernel_size = hp['enc_ffn_kernel_size']
        self.txt_embed = nn.Embedding(vocab_size, h, PAD_INDEX)
        self.use_lang_id = hp.get('use_lang_id', False)
        if self.use_lang_id:
            self.lang_embed = nn.Embedding(hp['num_lang'] + 1, h, padding_idx=0)
        self.dur_embed = nn.Linear(1, h)
        self.encoder = _Encoder(h, self.enc_layers, self.num_heads, self.ffn_kernel_size, self.pos_mode, self.ffn_act)
        self.pitch_embed = nn.Linear(1, h)
        self.variance_embed_list = [v for v in VARIANCE_ORDER if hp.get(f'use_{v}_embed', False)]
        self.use_variance_embeds = len(self.variance_embed_list) > 0
        if self.use_variance_embeds:
            self.variance_embeds = nn.ModuleDict({v: nn.Linear(1, h) for v in self.variance_embed_list})
        self.use_key_shift_embed = hp.get('use_key_shift_embed', False)
        if self.use_key_shift_embed:
            self.key_shift_embed = nn.Linear(1, h)
        self.use_speed_embed = hp.get('use_speed_embed', False)
        if self.use_speed_embed:
            self.speed_embed = nn.Linear(1, h)
        self.use_spk_id = hp['use_spk_id']
        if self.use_spk_id:
            self.spk_embed = nn.Embedding(hp['num_spk'], h)
        self.num_spk = hp['num_spk'] if self.use_spk_id else 0
        self.num_lang = hp['num_lang'] if self.use_lang_id else 0

    def _config(self, device_index):
        flags = sum(_lib.EMBED_FLAGS[v] for v in self.variance_embed_list)
        flags |= _lib.EMBED_FLAGS["key_shift"] if self.use_key_shift_embed else 0
        flags |= _lib.EMBED_FLAGS["speed"] if self.use_speed_embed else 0
        return _lib.DsdEncoderConfig(C.sizeof(_lib.DsdEncoderConfig), self.vocab_size, self._hidden, self.enc_layers,
                                     self.num_heads, self.ffn_kernel_size, self.num_spk, self.num_lang, flags,
                                     self.pos_mode, device_index, self.ffn_act)

    def _extra_weights(self):
        return positional_extra_weights(self.pos_mode, self._hidden)

    def prepare_cond(self, cond, layout="BHT"):
        raise RuntimeError("FastSpeech2Acoustic produces the condition; call forward(txt_tokens, mel2ph, f0, ...)")

    def forward(self, txt_tokens, mel2ph, f0, key_shift=None, speed=None, spk_embed_id=None, languages=None, **kwargs):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError(
                "diffsinger_amd.FastSpeech2Acoustic is inference-only (no backward kernels): call it under "
                "torch.no_grad(); training stays on the reference module.")
        if txt_tokens.dim() != 2 or mel2ph.dim() != 2 or tuple(f0.shape) != tuple(mel2ph.shape):
            raise ValueError(f"txt_tokens [B, T_txt], mel2ph [B, T], f0 [B, T] expected; got {tuple(txt_tokens.shape)}, "
                             f"{tuple(mel2ph.shape)}, {tuple(f0.shape)}")
        dev = f0.device
        handle = self.native_handle(dev)
        b, t_txt = txt_tokens.shape
        t = mel2ph.shape[1]
        out = torch.empty((b, t, self._hidden), device=dev, dtype=torch.float32)
        if b == 0 or t == 0 or t_txt == 0:
            return out.zero_()
        keep = []

        def i64(v):
            v = v.detach().to(device=dev, dtype=torch.int64).contiguous()
            keep.append(v)
            return v.data_ptr()

        def f32(v, shape):
            v = v.detach().to(device=dev, dtype=torch.float32).expand(shape).contiguous()
            keep.append(v)
            return v.data_ptr()

        ex = _lib.DsdEncodeExtras()
        if self.use_lang_id:
            if languages is None:
                raise ValueError("use_lang_id model: `languages` [B, T_txt] is required")
            ex.languages = i64(languages)
        if self.use_spk_id:
            mix = kwargs.get('spk_mix_embed')
            if mix is not None:
                mix = mix.detach().to(device=dev, dtype=torch.float32)
                if mix.dim() != 3 or mix.shape[-1] != self._hidden:
                    raise ValueError(f"spk_mix_embed must be [B, T or 1, {self._hidden}], got {tuple(mix.shape)}")
                mix = mix.expand(b, mix.shape[1], self._hidden).contiguous()
                keep.append(mix)
                ex.spk_mix_embed = mix.data_ptr()
                ex.spk_mix_bstride = mix.stride(0)
                ex.spk_mix_tstride = mix.stride(1) if mix.shape[1] > 1 else 0
            elif spk_embed_id is not None:
                ex.spk_embed_id = i64(spk_embed_id.reshape(-1))
            else:
                raise ValueError("use_spk_id model: `spk_embed_id` [B] or `spk_mix_embed` is required")
        for name in self.variance_embed_list:
            if kwargs.get(name) is None:
                raise KeyError(name)           # the reference indexes variances[v_name] (acoustic_encoder.py:69)
            setattr(ex, name, f32(kwargs[name], (b, t)))
        if self.use_key_shift_embed:
            ex.key_shift = f32(key_shift, (b, t))
        if self.use_speed_embed:
            ex.speed = f32(speed, (b, t))
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(handle, _lib.lib().dsd_encode(handle, C.c_void_p(i64(txt_tokens)), C.c_void_p(i64(mel2ph)),
                                                 C.c_void_p(f32(f0, (b, t))), b, t_txt, t, C.byref(ex),
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(stream)), "dsd_encode")
        return out
