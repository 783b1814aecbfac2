"""numpy fp32 restatement of the variance model around the pitch / multi-variance denoisers
(TEST ORACLE - BASELINE config 5's callers; the denoisers themselves are oracle/backbones.py + oracle/diffusion.py).

Follows (rotary-embedding configuration, `configs/variance.yaml:38` use_rope: true), inference branch only:
  * DiffSingerVariance.forward                   modules/toplevel.py:198-309
  * FastSpeech2Variance.forward                  modules/fastspeech/variance_encoder.py:52-99
  * MelodyEncoder.forward                        modules/fastspeech/variance_encoder.py:128-148
  * DurationPredictor.forward / out2dur          modules/fastspeech/tts_modules.py:102-134
  * RhythmRegulator.forward                      modules/fastspeech/tts_modules.py:255-275
  * LengthRegulator.forward                      modules/fastspeech/tts_modules.py:280-311
  * ParameterAdaptorModule.build_adaptor         modules/fastspeech/param_adaptor.py:32-89
"""
from __future__ import annotations

import numpy as np

from . import diffusion as od
from .backbones import F32
from .encoder import _lin1, _ln, fs2_encoder

VARIANCE_CHECKLIST = ("energy", "breathiness", "voicing", "tension")     # param_adaptor.py:10


def pos_mode(get):
    """'rope' | 'rel' | 'sin' | 'none' as FastSpeech2Encoder.__init__ decides (tts_modules.py:362-364,378-384)."""
    use_pos = get("use_pos_embed") if get("use_pos_embed") is not None else True
    if use_pos and get("use_rope"):
        return "rope"
    if not use_pos:
        return "none"
    return "rel" if get("rel_pos") else "sin"


def sub(p, prefix):
    """The entries of a flat state dict below `prefix`, with the prefix removed."""
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def duration_predictor(p, xs, pad_mask, offset=1.0, prefix="dur_predictor."):
    """xs [B, L, H] -> durations in the linear domain, clamped at 0 (infer=True)."""
    mask = (1.0 - pad_mask.astype(F32))[:, :, None]
    x = np.asarray(xs, dtype=F32)
    l = 0
    while f"{prefix}conv.{l}.1.weight" in p:
        w, b = p[f"{prefix}conv.{l}.1.weight"], p[f"{prefix}conv.{l}.1.bias"]        # [C, Cin, k]
        ks = w.shape[2]
        pad = ks // 2
        bsz, seq, _ = x.shape
        xp = np.zeros((bsz, seq + 2 * pad, x.shape[2]), dtype=F32)
        xp[:, pad:pad + seq] = x
        y = np.zeros((bsz, seq, w.shape[0]), dtype=F32)
        for j in range(ks):
            y += xp[:, j:j + seq] @ np.ascontiguousarray(w[:, :, j]).T
        y = np.maximum((y + b).astype(F32), F32(0))                                  # Conv1d -> ReLU
        y = _ln(y, p[f"{prefix}conv.{l}.3.weight"], p[f"{prefix}conv.{l}.3.bias"], eps=1e-12)   # tts_modules.py:30-41: eps 1e-12
        x = (y * mask).astype(F32)
        l += 1
    out = (x @ p[prefix + "linear.weight"].T + p[prefix + "linear.bias"]).astype(F32)
    out = (out * mask)[:, :, 0]
    dur = (np.exp(out).astype(F32) - F32(offset)).astype(F32)
    return np.maximum(dur, F32(0))


def fs2_variance_forward(p, hp, txt_tokens, midi, ph2word, ph_dur=None, word_dur=None, spk_embed=None, languages=None):
    """-> encoder_out [B, L, H], ph_dur_pred [B, L] or None.  `p`: the `fs2.` sub-dict."""
    txt_tokens, ph2word = np.asarray(txt_tokens), np.asarray(ph2word)
    bsz = txt_tokens.shape[0]
    txt_embed = p["txt_embed.weight"][txt_tokens]
    if hp["predict_dur"]:                                   # linguistic_mode == 'word'
        onset = np.diff(ph2word, axis=1, prepend=np.zeros((bsz, 1), dtype=ph2word.dtype)) > 0
        extra = p["onset_embed.weight"][onset.astype(np.int64)]
        if word_dur is None:
            wd = np.zeros((bsz, int(ph2word.max()) + 1), dtype=np.asarray(ph_dur).dtype)
            for b in range(bsz):
                np.add.at(wd[b], ph2word[b], np.asarray(ph_dur)[b])
            word_dur = wd[:, 1:]
        wdp = np.concatenate([np.zeros((bsz, 1), dtype=np.asarray(word_dur).dtype), np.asarray(word_dur)], axis=1)
        wd_ph = np.take_along_axis(wdp, ph2word, axis=1)
        extra = (extra + _lin1(wd_ph.astype(F32), p, "word_dur_embed")).astype(F32)
    else:
        extra = _lin1(np.asarray(ph_dur).astype(F32), p, "ph_dur_embed")
    if hp.get("use_lang_id"):
        extra = (extra + p["lang_embed.weight"][np.asarray(languages)]).astype(F32)
    enc = fs2_encoder(p, txt_embed, extra, txt_tokens == 0, hp["num_heads"], pos=pos_mode(hp.get),
                      ffn_act=hp.get("ffn_act") or "gelu")
    if not hp["predict_dur"]:
        return enc, None
    dur_cond = (enc + p["midi_embed.weight"][np.asarray(midi)]).astype(F32)
    if spk_embed is not None:
        dur_cond = (dur_cond + np.asarray(spk_embed, dtype=F32)).astype(F32)
    dur = duration_predictor(p, dur_cond, txt_tokens == 0, offset=hp["dur_prediction_args"]["log_offset"])
    return enc, dur


def melody_encoder(p, hp, note_midi, note_rest, note_dur, glide=None):
    """`p`: the `melody_encoder.` sub-dict -> [B, T_n, H]."""
    args = hp["melody_encoder_args"]
    note_midi = np.asarray(note_midi, dtype=F32)
    keep = (~np.asarray(note_rest, dtype=bool)).astype(F32)[:, :, None]
    midi_embed = (_lin1(note_midi, p, "note_midi_embed") * keep).astype(F32)
    extra = _lin1(np.asarray(note_dur).astype(F32), p, "note_dur_embed")
    if hp.get("use_glide_embed"):
        extra = (extra + p["note_glide_embed.weight"][np.asarray(glide)] * F32(hp["glide_embed_scale"])).astype(F32)
    enc = fs2_encoder(p, midi_embed, extra, note_midi < 0, args.get("num_heads", hp["num_heads"]),
                      pos=pos_mode(lambda k: args.get(k, hp.get(k))), ffn_act=args.get("ffn_act", hp.get("ffn_act")) or "gelu")
    return (enc @ p["out_proj.weight"].T + p["out_proj.bias"]).astype(F32)


def rhythm_regulator(ph_dur, ph2word, word_dur, eps=1e-5):
    ph2word = np.asarray(ph2word)
    ph_dur = (np.asarray(ph_dur).astype(F32) * (ph2word > 0)).astype(F32)
    word_dur = np.asarray(word_dur).astype(F32)
    bsz = ph_dur.shape[0]
    win = np.zeros((bsz, int(ph2word.max()) + 1), dtype=F32)
    for b in range(bsz):
        for i in range(ph_dur.shape[1]):                    # scatter_add in index order, fp32
            win[b, ph2word[b, i]] = F32(win[b, ph2word[b, i]] + ph_dur[b, i])
    alpha_w = (word_dur / np.maximum(win[:, 1:], F32(eps))).astype(F32)
    alpha_ph = np.take_along_axis(np.concatenate([np.zeros((bsz, 1), F32), alpha_w], axis=1), ph2word, axis=1)
    return np.round((ph_dur * alpha_ph).astype(F32)).astype(np.int64)        # torch.round: half to even, like numpy


def length_regulator(dur):
    dur = np.asarray(dur, dtype=np.int64)
    t_max = int(dur.sum(-1).max())
    out = np.zeros((dur.shape[0], t_max), dtype=np.int64)
    for b in range(dur.shape[0]):
        pos = 0
        for i, d in enumerate(dur[b]):
            out[b, pos:pos + d] = i + 1
            pos += int(d)
    return out


def _gather_frames(enc, idx):
    enc = np.concatenate([np.zeros_like(enc[:, :1]), enc], axis=1)          # F.pad(encoder_out, [0, 0, 1, 0])
    return np.take_along_axis(enc, np.asarray(idx)[:, :, None].repeat(enc.shape[-1], axis=2), axis=1).astype(F32)


def build_adaptor_ranges(hp):
    """param_adaptor.py:32-66 -> names, ranges, clamps."""
    names, ranges, clamps = [], [], []
    for n in VARIANCE_CHECKLIST:
        if not hp.get("predict_" + n):
            continue
        names.append(n)
        if n == "tension":
            ranges.append((hp["tension_logit_min"], hp["tension_logit_max"]))
            clamps.append((hp["tension_logit_min"], hp["tension_logit_max"]))
        else:
            ranges.append((hp[n + "_db_min"], hp[n + "_db_max"]))
            clamps.append((hp[n + "_db_min"], 0.0))
    return names, ranges, clamps


def variance_model_forward(p, hp, make_fn, txt_tokens, midi, ph2word, ph_dur=None, word_dur=None, mel2ph=None,
                           note_midi=None, note_rest=None, note_dur=None, note_glide=None, mel2note=None,
                           base_pitch=None, pitch=None, pitch_expr=None, pitch_retake=None, variance_retake=None,
                           spk_id=None, languages=None, noise_pitch=None, noise_var=None, variances=None):
    """DiffSingerVariance.forward(infer=True).  `make_fn(prefix, backbone_args)` returns the numpy backbone
    `fn(x, t, cond)` for the denoiser whose weights sit below `prefix`; x_T is passed in (`noise_*`).
    -> dur_pred [B, L] or None, pitch_pred [B, T] or None, {name: [B, T]}"""
    hidden = hp["hidden_size"]
    ph_spk = spk = None
    if hp.get("use_spk_id"):
        ph_spk = spk = p["spk_embed.weight"][np.asarray(spk_id)][:, None, :]
    enc, dur_pred = fs2_variance_forward(sub(p, "fs2."), hp, txt_tokens, midi, ph2word, ph_dur=ph_dur, word_dur=word_dur,
                                         spk_embed=ph_spk, languages=languages)
    names, ranges, clamps = build_adaptor_ranges(hp)
    if not hp["predict_pitch"] and not names:
        return dur_pred, None, {}
    if mel2ph is None and word_dur is not None:
        mel2ph = length_regulator(rhythm_regulator(dur_pred, ph2word, word_dur))
        t_len = np.asarray(base_pitch).shape[1]
        mel2ph = np.pad(mel2ph, [(0, 0), (0, t_len - mel2ph.shape[1])])
    mel2ph = np.asarray(mel2ph)
    condition = _gather_frames(enc, mel2ph)
    if spk is not None:
        condition = (condition + spk).astype(F32)
    reflow = hp.get("diffusion_type", "ddpm") == "reflow"

    def sample(prefix, args, out_dims, nf, smin, smax, cond, noise):
        fn = make_fn(prefix + ("velocity_fn." if reflow else "denoise_fn."), args)
        cond_t = np.ascontiguousarray(np.swapaxes(cond, 1, 2))
        if reflow:
            d = od.RectifiedFlow(fn, out_dims, nf, time_scale_factor=hp["time_scale_factor"], spec_min=smin, spec_max=smax)
            return d, d.inference(cond_t, noise, sampling_algorithm=hp["sampling_algorithm"],
                                  sampling_steps=hp["sampling_steps"])
        d = od.GaussianDiffusion(fn, out_dims, nf, timesteps=hp["timesteps"], k_step=hp["K_step"], spec_min=smin,
                                 spec_max=smax)
        return d, d.inference(cond_t, noise, diff_speedup=hp["diff_speedup"], diff_accelerator=hp["diff_accelerator"],
                              K_step_infer=hp["K_step"])

    pitch_pred = None
    if hp["predict_pitch"]:
        base_pitch = np.asarray(base_pitch, dtype=F32)
        if hp.get("use_melody_encoder"):
            mel_out = melody_encoder(sub(p, "melody_encoder."), hp, note_midi, note_rest, note_dur, glide=note_glide)
            pitch_cond = (condition + _gather_frames(mel_out, mel2note)).astype(F32)
        else:
            pitch_cond = condition.copy()
        retake_unset = pitch_retake is None
        if retake_unset:
            pitch_retake = np.ones(mel2ph.shape, dtype=bool)
        pitch_retake = np.asarray(pitch_retake, dtype=bool)
        table = p["pitch_retake_embed.weight"]
        if pitch_expr is None:
            retake_embed = table[pitch_retake.astype(np.int64)]
        else:
            e = (np.asarray(pitch_expr, dtype=F32) * pitch_retake)[:, :, None].astype(F32)
            retake_embed = (e * table[1] + (F32(1.0) - e) * table[0]).astype(F32)
        pitch_cond = (pitch_cond + retake_embed).astype(F32)
        if hp.get("use_melody_encoder"):
            if retake_unset:
                delta_in = np.zeros_like(base_pitch)
            else:
                delta_in = ((np.asarray(pitch, dtype=F32) - base_pitch) * ~pitch_retake).astype(F32)
            pitch_cond = (pitch_cond + _lin1(delta_in, p, "delta_pitch_embed")).astype(F32)
        else:
            if not retake_unset:
                base_pitch = (base_pitch * pitch_retake + np.asarray(pitch, dtype=F32) * ~pitch_retake).astype(F32)
            pitch_cond = (pitch_cond + _lin1(base_pitch, p, "base_pitch_embed")).astype(F32)
        ph = hp["pitch_prediction_args"]
        nf, smin, smax = od.repetitive_spec_ranges(ph["pitd_norm_min"], ph["pitd_norm_max"])
        d, x = sample("pitch_predictor.", ph["backbone_args"], ph["repeat_bins"], nf, smin, smax, pitch_cond, noise_pitch)
        pitch_pred = od.pitch_denorm(d, x, ph["pitd_clip_min"], ph["pitd_clip_max"])
    if not names:
        return dur_pred, pitch_pred, {}
    if pitch is None:
        pitch = (np.asarray(base_pitch, dtype=F32) + pitch_pred).astype(F32)
    var_cond = (condition + _lin1(np.asarray(pitch, dtype=F32), p, "pitch_embed")).astype(F32)
    if variance_retake is not None:
        acc = np.zeros_like(var_cond)
        for n in names:                                     # torch.stack([...], dim=-1).sum(-1): list order
            keep = (~np.asarray(variance_retake[n], dtype=bool))[:, :, None]
            acc = (acc + _lin1(np.asarray(variances[n], dtype=F32), p, f"variance_embeds.{n}") * keep).astype(F32)
        var_cond = (var_cond + acc).astype(F32)
    vh = hp["variances_prediction_args"]
    repeat_bins = vh["total_repeat_bins"] // len(names)
    nf, smin, smax = od.repetitive_spec_ranges([r[0] for r in ranges], [r[1] for r in ranges]) if len(names) > 1 else \
        od.repetitive_spec_ranges(ranges[0][0], ranges[0][1])
    d, x = sample("variance_predictor.", vh["backbone_args"], repeat_bins, nf, smin, smax, var_cond, noise_var)
    outs = od.multivar_denorm(d, x, clamps)
    return dur_pred, pitch_pred, dict(zip(names, outs))
