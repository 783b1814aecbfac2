"""Variance model on libdsdenoise (drop-in for `modules/toplevel.py` DiffSingerVariance, inference branch).

The classes carry the reference's names and parameter layout - `fs2.{txt_embed, onset_embed, word_dur_embed |
ph_dur_embed, lang_embed, encoder.*, midi_embed, dur_predictor.*}`, `melody_encoder.*`, `base_pitch_embed |
delta_pitch_embed`, `pitch_retake_embed`, `pitch_predictor.*`, `pitch_embed`, `variance_embeds.*`,
`variance_predictor.*`, `spk_embed` - so a variance checkpoint loads with strict=True.  Everything dense runs on the HIP
library: the FastSpeech2Encoder stacks and the DurationPredictor (`dsd_token_encode`, `dsd_predict_dur`), every
embedding sum (`dsd_cond_assemble`), and the pitch / multi-variance denoisers (diffusion.py).  What stays in torch is
integer bookkeeping on [B, T_ph]-sized tensors: word onsets, the rhythm and length regulators.
Encoders: the rotary configuration (`use_rope: true`, configs/variance.yaml:38) and the pre-rotary ones (`use_rope:
false` with `rel_pos: true` or `false`, or `use_pos_embed: false`); inference only; no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .backbones import _NativeBackbone
from .diffusion import (MultiVarianceDiffusion, MultiVarianceRectifiedFlow, PitchDiffusion, PitchRectifiedFlow)
from .encoder import PAD_INDEX, _Encoder, ffn_act_of, pos_mode_of, positional_extra_weights
from .harness import length_regulator
from .hparams import hparams
from .toplevel import get_backbone_args

VARIANCE_CHECKLIST = ['energy', 'breathiness', 'voicing', 'tension']      # param_adaptor.py:10


# ------------------------------------------------------------------------------------------------ dsd_cond_assemble
def lin1(layer: nn.Linear, x):
    """Linear(1, H)(x[:, :, None]) as two assemble terms: x * weight[:, 0] and 1 * bias."""
    return [(x, layer.weight.reshape(-1)), (None, layer.bias)]


def assemble(bsz, t_len, hidden, gathers, terms, device):
    """out[b, t, :] = sum_g scale * row_scale[b, t] * table[b][idx[b, t] + offset, :] + sum_k s_k[b, t] * v_k[:]

    gathers: (table [rows, H] or [B, rows, H], idx [B, T] int64, offset, scale[, row_scale [B, T]]);
    terms: (s [B, T] or None, v [H])."""
    if device.type != "cuda":
        raise RuntimeError("diffsinger_amd.variance runs only on an MI355X (HIP) device; there is no CPU path")
    a = _lib.DsdAssembleArgs()
    a.struct_size = C.sizeof(_lib.DsdAssembleArgs)
    a.device = device.index if device.index is not None else torch.cuda.current_device()
    a.B, a.T, a.H = bsz, t_len, hidden
    keep = []

    def f32(v, shape):
        v = v.detach().to(device=device, dtype=torch.float32).contiguous()
        if tuple(v.shape) != shape:
            raise ValueError(f"expected shape {shape}, got {tuple(v.shape)}")
        keep.append(v)
        return v.data_ptr()

    a.n_gather = len(gathers)
    for i, g in enumerate(gathers):
        table, idx, off, scale = g[:4]
        table = table.detach().to(device=device, dtype=torch.float32).contiguous()
        if table.shape[-1] != hidden or (table.dim() == 3 and table.shape[0] != bsz):
            raise ValueError(f"gather {i}: table {tuple(table.shape)} does not match [B={bsz}, rows, H={hidden}]")
        idx = idx.detach().to(device=device, dtype=torch.int64).contiguous()
        if tuple(idx.shape) != (bsz, t_len):
            raise ValueError(f"gather {i}: index {tuple(idx.shape)} is not [B={bsz}, T={t_len}]")
        keep += [table, idx]
        a.gather[i].table = table.data_ptr()
        a.gather[i].batch_stride = table.shape[1] * hidden if table.dim() == 3 else 0
        a.gather[i].rows = table.shape[-2]
        a.gather[i].idx = idx.data_ptr()
        a.gather[i].idx_offset = off
        a.gather[i].scale = scale
        a.gather[i].row_scale = f32(g[4], (bsz, t_len)) if len(g) > 4 and g[4] is not None else None
    a.n_terms = len(terms)
    for i, (s, v) in enumerate(terms):
        a.term[i].s = None if s is None else f32(s, (bsz, t_len))
        a.term[i].v = f32(v, (hidden,))
    out = torch.empty((bsz, t_len, hidden), device=device, dtype=torch.float32)
    stream = torch.cuda.current_stream(device).cuda_stream
    rc = _lib.lib().dsd_cond_assemble(C.byref(a), C.c_void_p(out.data_ptr()), C.c_void_p(stream))
    if rc != 0:
        raise _lib.NativeLibraryError(f"dsd_cond_assemble failed ({rc}): {_lib.lib().dsd_last_error(None).decode()}")
    return out


def _arange_idx(bsz, t_len, device):
    return torch.arange(t_len, device=device, dtype=torch.int64)[None].expand(bsz, t_len).contiguous()


def _check_infer(module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise RuntimeError(f"diffsinger_amd.{type(module).__name__} is inference-only (no backward kernels): call it "
                           "under torch.no_grad(); training stays on the reference module.")


def _check_encoder_hparams(get):
    """-> (DSD_POS_*, DSD_FFN_*) as FastSpeech2Encoder.__init__ reads them (tts_modules.py:353-384)."""
    return pos_mode_of(get), ffn_act_of(get)


class _TokenEncoderBase(_NativeBackbone):
    """FastSpeech2Encoder (+ heads) on `dsd_token_encode` / `dsd_predict_dur`."""
    _native_prefixes = ("encoder.",)

    def _native_state(self):
        return {k: v for k, v in self.state_dict().items() if k.startswith(self._native_prefixes)}

    def _extra_weights(self):
        return positional_extra_weights(self.pos_mode, self._hidden)

    def prepare_cond(self, cond, layout="BHT"):
        raise RuntimeError(f"{type(self).__name__} is an encoder; call forward(...)")

    def _encode(self, embed, padding_mask, out_dims):
        dev = embed.device
        handle = self.native_handle(dev)
        bsz, n_tok, _ = embed.shape
        mask = padding_mask.to(device=dev, dtype=torch.uint8).contiguous()
        out = torch.empty((bsz, n_tok, out_dims), device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(handle, _lib.lib().dsd_token_encode(handle, C.c_void_p(embed.data_ptr()), C.c_void_p(mask.data_ptr()), bsz,
                                                       n_tok, C.c_void_p(out.data_ptr()), C.c_void_p(stream)),
                   "dsd_token_encode")
        return out


class DurationPredictor(nn.Module):
    """Parameter holder with the names of tts_modules.py:53-100 (`conv.N.1` Conv1d, `conv.N.3` LayerNorm over the
    channels, `linear`); runs inside `FastSpeech2Variance` on `dsd_predict_dur`."""

    def __init__(self, in_dims, n_layers=2, n_chans=384, kernel_size=3, dropout_rate=0.1, offset=1.0, dur_loss_type='mse'):
        super().__init__()
        if dur_loss_type not in ('mse', 'huber'):
            raise NotImplementedError(dur_loss_type)
        self.offset, self.kernel_size, self.n_layers, self.n_chans = offset, kernel_size, n_layers, n_chans
        self.conv = nn.ModuleList()
        for idx in range(n_layers):
            cin = in_dims if idx == 0 else n_chans
            self.conv.append(nn.Sequential(nn.Identity(), nn.Conv1d(cin, n_chans, kernel_size, padding=kernel_size // 2),
                                           nn.ReLU(), nn.LayerNorm(n_chans, eps=1e-12), nn.Dropout(dropout_rate)))
        self.linear = nn.Linear(n_chans, 1)


class FastSpeech2Variance(_TokenEncoderBase):
    """modules/fastspeech/variance_encoder.py:14-99."""
    _native_prefixes = ("encoder.", "dur_predictor.")

    def __init__(self, vocab_size):
        super().__init__()
        hp = hparams
        self.pos_mode, self.ffn_act = _check_encoder_hparams(hp.get)
        h = self._hidden = hp['hidden_size']
        self.predict_dur = hp['predict_dur']
        self.linguistic_mode = 'word' if self.predict_dur else 'phoneme'
        self.use_lang_id = hp.get('use_lang_id', False)
        self.enc_layers, self.num_heads, self.ffn_kernel_size = hp['enc_layers'], hp['num_heads'], hp['enc_ffn_kernel_size']
        self.txt_embed = nn.Embedding(vocab_size, h, PAD_INDEX)
        if self.use_lang_id:
            self.lang_embed = nn.Embedding(hp['num_lang'] + 1, h, padding_idx=0)
        if self.predict_dur:
            self.onset_embed = nn.Embedding(2, h)
            self.word_dur_embed = nn.Linear(1, h)
        else:
            self.ph_dur_embed = nn.Linear(1, h)
        self.encoder = _Encoder(h, self.enc_layers, self.num_heads, self.ffn_kernel_size, self.pos_mode, self.ffn_act)
        if self.predict_dur:
            d = hp['dur_prediction_args']
            self.midi_embed = nn.Embedding(128, h)
            self.dur_predictor = DurationPredictor(in_dims=h, n_chans=d['hidden_size'], n_layers=d['num_layers'],
                                                   dropout_rate=d['dropout'], kernel_size=d['kernel_size'],
                                                   offset=d['log_offset'], dur_loss_type=d['loss_type'])

    def _config(self, device_index):
        d = self.dur_predictor if self.predict_dur else None
        return _lib.DsdTokenEncoderConfig(C.sizeof(_lib.DsdTokenEncoderConfig), self._hidden, self.enc_layers, self.num_heads,
                                          self.ffn_kernel_size, 0, d.n_layers if d else 0, d.n_chans if d else 0,
                                          d.kernel_size if d else 0, float(d.offset) if d else 0.0, self.pos_mode, device_index,
                                          self.ffn_act)

    def forward(self, txt_tokens, midi, ph2word, ph_dur=None, word_dur=None, spk_embed=None, languages=None, infer=True):
        """-> encoder_out [B, T_ph, H], ph_dur_pred [B, T_ph] or None (variance_encoder.py:52-99)."""
        _check_infer(self)
        if not infer:
            raise NotImplementedError("training (infer=False) stays on the reference module")
        dev = txt_tokens.device
        bsz, n_ph = txt_tokens.shape
        h = self._hidden
        gathers = [(self.txt_embed.weight, txt_tokens, 0, math.sqrt(h))]       # embed_scale * main_embed (tts_modules.py:387)
        if self.linguistic_mode == 'word':
            onset = torch.diff(ph2word, dim=1, prepend=ph2word.new_zeros(bsz, 1)) > 0
            gathers.append((self.onset_embed.weight, onset.long(), 0, 1.0))
            if word_dur is None:
                word_dur = ph_dur.new_zeros(bsz, int(ph2word.max()) + 1).scatter_add(1, ph2word, ph_dur)[:, 1:]
            wd_ph = torch.gather(F.pad(word_dur, [1, 0], value=0), 1, ph2word)          # [B, T_w] => [B, T_ph]
            terms = lin1(self.word_dur_embed, wd_ph.float())
        else:
            terms = lin1(self.ph_dur_embed, ph_dur.float())
        if self.use_lang_id:
            gathers.append((self.lang_embed.weight, languages, 0, 1.0))
        embed = assemble(bsz, n_ph, h, gathers, terms, dev)
        pad = txt_tokens == PAD_INDEX
        enc = self._encode(embed, pad, h)
        if not self.predict_dur:
            return enc, None
        gathers = [(enc, _arange_idx(bsz, n_ph, dev), 0, 1.0), (self.midi_embed.weight, midi, 0, 1.0)]
        if spk_embed is not None:                                  # [B, 1, H] or [B, T_ph, H]
            idx = _arange_idx(bsz, n_ph, dev) if spk_embed.shape[1] == n_ph and n_ph > 1 else \
                torch.zeros((bsz, n_ph), dtype=torch.int64, device=dev)
            gathers.append((spk_embed.expand(bsz, -1, -1), idx, 0, 1.0))
        dur_cond = assemble(bsz, n_ph, h, gathers, [], dev)
        handle = self.native_handle(dev)
        mask = pad.to(torch.uint8).contiguous()
        dur = torch.empty((bsz, n_ph), device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(handle, _lib.lib().dsd_predict_dur(handle, C.c_void_p(dur_cond.data_ptr()), C.c_void_p(mask.data_ptr()), bsz,
                                                      n_ph, C.c_void_p(dur.data_ptr()), C.c_void_p(stream)), "dsd_predict_dur")
        return enc, dur


class MelodyEncoder(_TokenEncoderBase):
    """modules/fastspeech/variance_encoder.py:102-148."""
    _native_prefixes = ("encoder.", "out_proj.")

    def __init__(self, enc_hparams: dict):
        super().__init__()

        def get(key):
            return enc_hparams.get(key, hparams.get(key))

        self.pos_mode, self.ffn_act = _check_encoder_hparams(get)
        h = self._hidden = get('hidden_size')
        self.enc_layers, self.num_heads, self.ffn_kernel_size = get('enc_layers'), get('num_heads'), get('enc_ffn_kernel_size')
        self.note_midi_embed = nn.Linear(1, h)
        self.note_dur_embed = nn.Linear(1, h)
        self.use_glide_embed = hparams['use_glide_embed']
        self.glide_embed_scale = hparams['glide_embed_scale']
        if self.use_glide_embed:
            self.note_glide_embed = nn.Embedding(len(hparams['glide_types']) + 1, h, padding_idx=0)      # 0: none, 1: up, 2: down
        self.encoder = _Encoder(h, self.enc_layers, self.num_heads, self.ffn_kernel_size, self.pos_mode, self.ffn_act)
        self.out_dims = hparams['hidden_size']
        self.out_proj = nn.Linear(h, self.out_dims)

    def _config(self, device_index):
        return _lib.DsdTokenEncoderConfig(C.sizeof(_lib.DsdTokenEncoderConfig), self._hidden, self.enc_layers, self.num_heads,
                                          self.ffn_kernel_size, self.out_dims, 0, 0, 0, 0.0, self.pos_mode, device_index, self.ffn_act)

    def forward(self, note_midi, note_rest, note_dur, glide=None):
        """note_midi float [B, T_n] (-1: padding), note_rest bool, note_dur int64, glide int64 -> [B, T_n, H]."""
        _check_infer(self)
        dev = note_midi.device
        bsz, n_note = note_midi.shape
        h = self._hidden
        keep = (~note_rest).float()
        s = math.sqrt(h)                                           # embed_scale applies to the main (MIDI) embedding only
        terms = [(note_midi.float() * keep * s, self.note_midi_embed.weight.reshape(-1)), (keep * s, self.note_midi_embed.bias)]
        terms += lin1(self.note_dur_embed, note_dur.float())
        gathers = []
        if self.use_glide_embed:
            gathers.append((self.note_glide_embed.weight, glide, 0, float(self.glide_embed_scale)))
        embed = assemble(bsz, n_note, h, gathers, terms, dev)
        return self._encode(embed, note_midi < 0, self.out_dims)


class RhythmRegulator(nn.Module):
    """tts_modules.py:250-275: rescale the phoneme durations of every word to the word's given duration."""

    def __init__(self, eps=1e-5):
        super().__init__()
        self.eps = eps

    def forward(self, ph_dur, ph2word, word_dur):
        ph_dur = ph_dur.float() * (ph2word > 0)
        per_word = ph_dur.new_zeros(ph_dur.shape[0], int(ph2word.max()) + 1).scatter_add(1, ph2word, ph_dur)[:, 1:]
        ratio = word_dur.float() / per_word.clamp(min=self.eps)
        return (ph_dur * torch.gather(F.pad(ratio, [1, 0]), 1, ph2word)).round().long()


class LengthRegulator(nn.Module):
    def forward(self, dur, dur_padding=None, alpha=None):
        return length_regulator(dur, dur_padding, alpha)


class ParameterAdaptorModule(nn.Module):
    """modules/fastspeech/param_adaptor.py:13-100."""

    def __init__(self):
        super().__init__()
        self.variance_prediction_list = [v for v in VARIANCE_CHECKLIST if hparams.get(f'predict_{v}', False)]
        self.predict_variances = len(self.variance_prediction_list) > 0

    def build_adaptor(self, cls=MultiVarianceDiffusion):
        ranges, clamps = [], []
        for v in self.variance_prediction_list:
            if v == 'tension':
                ranges.append((hparams['tension_logit_min'], hparams['tension_logit_max']))
                clamps.append((hparams['tension_logit_min'], hparams['tension_logit_max']))
            else:
                ranges.append((hparams[f'{v}_db_min'], hparams[f'{v}_db_max']))
                clamps.append((hparams[f'{v}_db_min'], 0.))
        vh = hparams['variances_prediction_args']
        total = vh['total_repeat_bins']
        assert total % len(self.variance_prediction_list) == 0, \
            f'Total number of repeat bins must be divisible by number of variance parameters ({len(self.variance_prediction_list)}).'
        backbone_type = vh.get('backbone_type', hparams.get('backbone_type', hparams.get('diff_decoder_type', 'wavenet')))
        kwargs = dict(ranges=ranges, clamps=clamps, repeat_bins=total // len(self.variance_prediction_list),
                      backbone_type=backbone_type, backbone_args=get_backbone_args(vh, backbone_type))
        if cls is MultiVarianceDiffusion:
            kwargs['timesteps'] = hparams.get('timesteps')
        else:
            kwargs['time_scale_factor'] = hparams.get('time_scale_factor')
        return cls(**kwargs)

    def collect_variance_inputs(self, **kwargs) -> list:
        return [kwargs.get(name) for name in self.variance_prediction_list]

    def collect_variance_outputs(self, variances) -> dict:
        return dict(zip(self.variance_prediction_list, variances))


class DiffSingerVariance(ParameterAdaptorModule):
    """modules/toplevel.py:125-309, `infer=True`."""

    category = 'variance'

    def __init__(self, vocab_size):
        super().__init__()
        hp = hparams
        h = hp['hidden_size']
        self.predict_dur, self.predict_pitch = hp['predict_dur'], hp['predict_pitch']
        self.use_spk_id = hp['use_spk_id']
        if self.use_spk_id:
            self.spk_embed = nn.Embedding(hp['num_spk'], h)
        self.fs2 = FastSpeech2Variance(vocab_size=vocab_size)
        self.rr = RhythmRegulator()
        self.lr = LengthRegulator()
        self.diffusion_type = hp.get('diffusion_type', 'ddpm')
        if self.diffusion_type not in ('ddpm', 'reflow'):
            raise ValueError(f"Invalid diffusion type: {self.diffusion_type}")
        if self.predict_pitch:
            self.use_melody_encoder = hp.get('use_melody_encoder', False)
            if self.use_melody_encoder:
                self.melody_encoder = MelodyEncoder(enc_hparams=hp['melody_encoder_args'])
                self.delta_pitch_embed = nn.Linear(1, h)
            else:
                self.base_pitch_embed = nn.Linear(1, h)
            self.pitch_retake_embed = nn.Embedding(2, h)
            ph = hp['pitch_prediction_args']
            btype = ph.get('backbone_type', hp.get('backbone_type', hp.get('diff_decoder_type', 'wavenet')))
            common = dict(vmin=ph['pitd_norm_min'], vmax=ph['pitd_norm_max'], cmin=ph['pitd_clip_min'],
                          cmax=ph['pitd_clip_max'], repeat_bins=ph['repeat_bins'], backbone_type=btype,
                          backbone_args=get_backbone_args(ph, btype))
            if self.diffusion_type == 'ddpm':
                self.pitch_predictor = PitchDiffusion(timesteps=hp['timesteps'], k_step=hp['K_step'], **common)
            else:
                self.pitch_predictor = PitchRectifiedFlow(time_scale_factor=hp['time_scale_factor'], **common)
        if self.predict_variances:
            self.pitch_embed = nn.Linear(1, h)
            self.variance_embeds = nn.ModuleDict({v: nn.Linear(1, h) for v in self.variance_prediction_list})
            self.variance_predictor = self.build_adaptor(
                cls=MultiVarianceDiffusion if self.diffusion_type == 'ddpm' else MultiVarianceRectifiedFlow)

    def forward(self, txt_tokens, midi, ph2word, ph_dur=None, word_dur=None, mel2ph=None,
                note_midi=None, note_rest=None, note_dur=None, note_glide=None, mel2note=None,
                base_pitch=None, pitch=None, pitch_expr=None, pitch_retake=None,
                variance_retake: Dict[str, torch.Tensor] = None, spk_id=None, languages=None, infer=True, lengths=None,
                **kwargs):
        """-> dur_pred [B, T_ph] | None, pitch_pred [B, T] | None, {name: [B, T]}.  Extra keywords: the current variance
        curves by name (`energy=...`), `ph_spk_mix_embed` / `spk_mix_embed`, and - for reproducible runs - the x_T of the
        two denoisers, `pitch_noise` [B, 1, R, T] and `variance_noise` [B, F, R, T]; `lengths` [B] runs a zero-padded batch
        of segments as a ragged batch (each comes out as if alone at its own frame count, dsd_set_lengths)."""
        _check_infer(self)
        if not infer:
            raise NotImplementedError("training (infer=False) stays on the reference DiffSingerVariance")
        dev = txt_tokens.device
        h = hparams['hidden_size']
        if self.use_spk_id:
            ph_mix, mix = kwargs.get('ph_spk_mix_embed'), kwargs.get('spk_mix_embed')
            if ph_mix is not None and mix is not None:
                ph_spk_embed, spk_embed = ph_mix, mix
            else:
                ph_spk_embed = spk_embed = self.spk_embed.weight[spk_id][:, None, :]             # [B] => [B, 1, H]
        else:
            ph_spk_embed = spk_embed = None
        encoder_out, dur_pred_out = self.fs2(txt_tokens, midi=midi, ph2word=ph2word, ph_dur=ph_dur, word_dur=word_dur,
                                             spk_embed=ph_spk_embed, languages=languages, infer=True)
        if not self.predict_pitch and not self.predict_variances:
            return dur_pred_out, None, {}
        if mel2ph is None and word_dur is not None:                 # inference from file
            mel2ph = self.lr(self.rr(dur_pred_out, ph2word, word_dur))
            mel2ph = F.pad(mel2ph, [0, base_pitch.shape[1] - mel2ph.shape[1]])
        bsz, t_len = mel2ph.shape
        # condition = gather(pad(encoder_out), mel2ph) (+ spk_embed)  (toplevel.py:233-238)
        cond_g = [(encoder_out, mel2ph, -1, 1.0)]
        if spk_embed is not None:
            idx = _arange_idx(bsz, t_len, dev) if spk_embed.shape[1] == t_len and t_len > 1 else \
                torch.zeros((bsz, t_len), dtype=torch.int64, device=dev)
            cond_g.append((spk_embed.expand(bsz, -1, -1), idx, 0, 1.0))

        pitch_pred_out = None
        if self.predict_pitch:
            gathers, terms = list(cond_g), []
            if self.use_melody_encoder:
                melody_out = self.melody_encoder(note_midi, note_rest, note_dur, glide=note_glide)
                gathers.append((melody_out, mel2note, -1, 1.0))
            retake_unset = pitch_retake is None
            if retake_unset:
                pitch_retake = torch.ones_like(mel2ph, dtype=torch.bool)
            # pitch_retake_embed (toplevel.py:253-264): e * embed[1] + (1 - e) * embed[0], e = retake (* pitch_expr)
            e = pitch_retake.float() if pitch_expr is None else (pitch_expr * pitch_retake).float()
            terms += [(e, self.pitch_retake_embed.weight[1]), (1.0 - e, self.pitch_retake_embed.weight[0])]
            if self.use_melody_encoder:
                delta_in = torch.zeros_like(base_pitch) if retake_unset else (pitch - base_pitch) * ~pitch_retake
                terms += lin1(self.delta_pitch_embed, delta_in)
            else:
                if not retake_unset:
                    base_pitch = base_pitch * pitch_retake + pitch * ~pitch_retake
                terms += lin1(self.base_pitch_embed, base_pitch)
            pitch_cond = assemble(bsz, t_len, h, gathers, terms, dev)
            pitch_pred_out = self.pitch_predictor(pitch_cond, infer=True, noise=kwargs.get('pitch_noise'), lengths=lengths)
        if not self.predict_variances:
            return dur_pred_out, pitch_pred_out, {}

        if pitch is None:
            pitch = base_pitch + pitch_pred_out
        terms = lin1(self.pitch_embed, pitch)
        variance_inputs = self.collect_variance_inputs(**kwargs)
        if variance_retake is not None:
            for name, v_in in zip(self.variance_prediction_list, variance_inputs):
                keep = (~variance_retake[name]).float()
                layer = self.variance_embeds[name]
                terms += [(v_in * keep, layer.weight.reshape(-1)), (keep, layer.bias)]
        var_cond = assemble(bsz, t_len, h, cond_g, terms, dev)
        outs = self.variance_predictor(var_cond, infer=True, noise=kwargs.get('variance_noise'), lengths=lengths)
        return dur_pred_out, pitch_pred_out, self.collect_variance_outputs(outs)
