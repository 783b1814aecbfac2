"""`.ds` segments in, completed `.ds` segments out: the variance inference harness around `variance.DiffSingerVariance`.
Behavioural twin of the reference's `inference/ds_variance.py` (score -> model inputs, the auto-completion rules of which
predictors run for which segment, writing predictions back into the project) with what it takes from outside the
reference restated here: note names -> MIDI numbers and Hz <-> MIDI (librosa's `note_to_midi(round_midi=False)`,
`hz_to_midi`, `midi_to_hz`: absent from this image, so these follow their definitions and are pinned by known answers
only - A4 = 69 = 440 Hz - see tests/test_variance_harness.py), and `utils/pitch_utils.py:13-20` (`interp_f0`).

Host-side glue only (wire format, integer frame bookkeeping, smoothing of the step-shaped MIDI curve); the model call is
`diffsinger_amd.variance.DiffSingerVariance` - or any module with the reference's forward signature.
"""
from __future__ import annotations

import copy
import json
import pathlib
import re
from typing import Dict, List, Optional, Set

import numpy as np
import torch
import torch.nn.functional as F

from .harness import AcousticHarness, VARIANCE_NAMES, length_regulator, load_speaker_mix
from .hparams import hparams
from .variance import RhythmRegulator

_NOTE = re.compile(r'^(?P<letter>[A-Ga-g])(?P<acc>[#♯𝄪b!♭𝄫♮n]*)(?P<octave>[+-]?\d+)?(?P<cents>[+-]\d+)?$')
_SEMITONE = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}
_ACCIDENTAL = {'#': 1, '♯': 1, '𝄪': 2, 'b': -1, '!': -1, '♭': -1, '𝄫': -2, '♮': 0, 'n': 0}


def note_to_midi(note: str) -> float:
    """'C4' -> 60.0, 'A#3' -> 58.0, 'Db5-20' -> 72.8: letter, accidentals, octave (default 0), cents; C-1 = 0."""
    m = _NOTE.match(note)
    if not m:
        raise ValueError(f'Improper note format: {note}')
    octave = int(m.group('octave')) if m.group('octave') else 0
    cents = int(m.group('cents')) * 1e-2 if m.group('cents') else 0.0
    return 12 * (octave + 1) + _SEMITONE[m.group('letter').upper()] + sum(_ACCIDENTAL[a] for a in m.group('acc')) + cents


def hz_to_midi(freq):
    return 12 * (np.log2(np.asanyarray(freq)) - np.log2(440.0)) + 69


def midi_to_hz(midi):
    return 440.0 * (2.0 ** ((np.asanyarray(midi) - 69.0) / 12.0))


def interp_f0(f0: np.ndarray):
    """Fill the unvoiced (zero) frames by linear interpolation in the log2 domain -> (f0, uv)."""
    uv = f0 == 0
    logf = np.log2(f0 + uv)
    logf[uv] = -np.inf
    if uv.any() and not uv.all():
        logf[uv] = np.interp(np.where(uv)[0], np.where(~uv)[0], logf[~uv])
    return 2 ** logf, uv


def mel2ph_to_dur(mel2ph: torch.Tensor, n_tokens: int) -> torch.Tensor:
    """frames per token: dur[b, k] = #{t : mel2ph[b, t] == k + 1}."""
    return mel2ph.new_zeros(mel2ph.shape[0], n_tokens + 1).scatter_add(1, mel2ph, torch.ones_like(mel2ph))[:, 1:]


def _seconds_to_frames(seconds: torch.Tensor, timestep: float) -> torch.Tensor:
    """[1, N] durations in seconds -> integer frame counts, by rounding the CUMULATIVE boundaries."""
    bounds = torch.round(torch.cumsum(seconds, dim=1) / timestep + 0.5).long()
    return torch.diff(bounds, dim=1, prepend=bounds.new_zeros(1, 1))


class VarianceHarness(AcousticHarness):
    """`model` = diffsinger_amd.variance.DiffSingerVariance; `predictions` = the set of curves asked for ('dur', 'pitch',
    'energy', ...); empty = auto-completion (predict whatever a segment does not already carry)."""

    def __init__(self, model, phoneme_dictionary, predictions: Optional[Set[str]] = None, spk_map: Optional[Dict[str, int]] = None,
                 lang_map: Optional[Dict[str, int]] = None, device=None):
        super().__init__(model, None, phoneme_dictionary, spk_map=spk_map, lang_map=lang_map, device=device)
        predictions = set(predictions or ())
        self.rr = RhythmRegulator()
        width = round(hparams['midi_smooth_width'] / self.timestep)
        window = np.sin(np.linspace(0, 1, width).astype(np.float32) * np.pi)
        self.smooth_kernel = torch.from_numpy(window).to(self.device)
        self.smooth_kernel = self.smooth_kernel / self.smooth_kernel.sum()
        glide_types = hparams.get('glide_types', [])
        assert 'none' not in glide_types, 'Type name \'none\' is reserved and should not appear in glide_types.'
        self.glide_map = {'none': 0, **{name: i + 1 for i, name in enumerate(glide_types)}}
        self.auto_completion_mode = len(predictions) == 0
        self.global_predict_dur = 'dur' in predictions and hparams['predict_dur']
        self.global_predict_pitch = 'pitch' in predictions and hparams['predict_pitch']
        self.variance_prediction_set = predictions.intersection(VARIANCE_NAMES)
        self.global_predict_variances = len(self.variance_prediction_set) > 0

    def smooth(self, curve: torch.Tensor) -> torch.Tensor:
        """[1, T] -> [1, T]: 'same'-sized correlation with the normalised half-sine window, edges replicated."""
        k = self.smooth_kernel.numel()
        left = (k - 1) // 2
        padded = F.pad(curve[:, None, :], [left, k - 1 - left], mode='replicate')
        return F.conv1d(padded, self.smooth_kernel[None, None])[:, 0]

    # -- one segment -> model inputs --------------------------------------------------------------------------
    def _notes(self, param, batch):
        midi = np.array([note_to_midi(n) if n != 'rest' else -1 for n in param['note_seq'].split()], dtype=np.float32)
        rest = midi < 0
        if rest.all():
            midi = np.full_like(midi, fill_value=60.)
        else:                                       # a rest takes the pitch of the nearest sounding note
            from scipy import interpolate
            nearest = interpolate.interp1d(np.where(~rest)[0], midi[~rest], kind='nearest', fill_value='extrapolate')
            midi[rest] = nearest(np.where(rest)[0])
        batch['note_midi'] = torch.from_numpy(midi).to(self.device)[None]
        batch['note_rest'] = torch.from_numpy(rest).to(self.device)[None]
        seconds = torch.from_numpy(np.array([param['note_dur'].split()], np.float32)).to(self.device)
        batch['note_dur'] = _seconds_to_frames(seconds, self.timestep)
        batch['mel2note'] = length_regulator(batch['note_dur'])
        n_notes = midi.shape[0]
        if hparams.get('use_glide_embed', False) and param.get('note_glide') is not None:
            batch['note_glide'] = torch.LongTensor([[self.glide_map.get(x, 0) for x in param['note_glide'].split()]]).to(self.device)
        else:
            batch['note_glide'] = torch.zeros(1, n_notes, dtype=torch.long, device=self.device)
        return n_notes, batch['mel2note'].shape[1]

    @staticmethod
    def _per_group_mean(curve, frame2group, group_frames, n_groups):
        """mean of a frame-level curve over each group (token or word): sum of curve / group size."""
        size = torch.gather(F.pad(group_frames, [1, 0], value=1), 1, frame2group)
        return curve.new_zeros(1, n_groups + 1).scatter_add(1, frame2group, curve / size)[:, 1:]

    def preprocess_input(self, param: dict, idx: int = 0, load_dur: bool = False, load_pitch: bool = False,
                         verbose: bool = False) -> Dict[str, torch.Tensor]:
        batch, summary = {}, {}
        self._tokens(param, batch)
        tokens = batch['tokens']
        n_ph = tokens.shape[1]
        ph_num = torch.from_numpy(np.array([param['ph_num'].split()], np.int64)).to(self.device)
        ph2word = batch['ph2word'] = length_regulator(ph_num)
        n_word = int(ph2word.max())
        n_notes, n_frames = self._notes(param, batch)
        note_dur = batch['note_dur']
        summary.update(words=n_word, notes=n_notes, tokens=n_ph, frames=n_frames, seconds='%.2f' % (n_frames * self.timestep))
        if hparams['use_spk_id']:
            batch['ph_spk_mix_id'], batch['ph_spk_mix_value'] = load_speaker_mix(
                param, summary, self.spk_map, self.timestep, self.device, mix_mode='token', mix_length=n_ph)
            batch['spk_mix_id'], batch['spk_mix_value'] = load_speaker_mix(
                param, summary, self.spk_map, self.timestep, self.device, mix_mode='frame', mix_length=n_frames)
        if load_dur:                                # phoneme durations come with the segment
            seconds = torch.from_numpy(np.array([param['ph_dur'].split()], np.float32)).to(self.device)
            ph_dur = _seconds_to_frames(seconds, self.timestep)
            mel2ph = length_regulator(ph_dur, tokens == 0)
            if mel2ph.shape[1] != n_frames:         # align the phones with the notes: the last phone takes the rest
                mel2ph = F.pad(mel2ph, [0, n_frames - mel2ph.shape[1]], value=mel2ph[0, -1])
                ph_dur = mel2ph_to_dur(mel2ph, n_ph)
            word_dur = note_dur.new_zeros(1, n_word + 1).scatter_add(1, ph2word, ph_dur)[:, 1:]
        else:                                       # word durations from the notes: a slur continues the word
            ph_dur = mel2ph = None
            is_slur = torch.BoolTensor([[int(s) for s in param['note_slur'].split()]]).to(self.device)
            note2word = torch.cumsum(~is_slur, dim=1)
            word_dur = note_dur.new_zeros(1, n_word + 1).scatter_add(1, note2word, note_dur)[:, 1:]
        batch['ph_dur'], batch['mel2ph'] = ph_dur, mel2ph
        mel2word = length_regulator(word_dur)
        if mel2word.shape[1] != n_frames:
            mel2word = F.pad(mel2word, [0, n_frames - mel2word.shape[1]], value=mel2word[0, -1])
            word_dur = mel2ph_to_dur(mel2word, n_word)
        batch['word_dur'] = word_dur
        # the step-shaped frame-level MIDI curve, smoothed, is the base pitch; its per-phone (or per-word) mean the `midi` input
        frame_midi = torch.gather(F.pad(batch['note_midi'], [1, 0]), 1, batch['mel2note'])
        batch['base_pitch'] = self.smooth(frame_midi)
        if ph_dur is not None:
            ph_midi = self._per_group_mean(frame_midi, mel2ph, ph_dur, n_ph)
        else:
            ph_midi = torch.gather(F.pad(self._per_group_mean(frame_midi, mel2word, word_dur, n_word), [1, 0]), 1, ph2word)
        batch['midi'] = ph_midi.round().long()
        if load_pitch:
            f0 = self._curve(param, 'f0_seq', 'f0_timestep', n_frames)
            batch['pitch'] = torch.from_numpy(hz_to_midi(interp_f0(f0)[0]).astype(np.float32)).to(self.device)[None]
        if self.model.predict_dur:
            summary['ph_dur'] = 'manual' if load_dur else 'auto' if self.auto_completion_mode or self.global_predict_dur else 'ignored'
        if self.model.predict_pitch:
            if load_pitch:
                summary['pitch'] = 'manual'
            elif self.auto_completion_mode or self.global_predict_pitch:
                summary['pitch'] = 'auto'
                expr = param.get('expr', 1.)        # expressiveness: a constant or a curve
                if isinstance(expr, (int, float, bool)):
                    summary['expr'] = f'static({expr:.3f})'
                    batch['expr'] = torch.FloatTensor([expr]).to(self.device)[:, None]
                else:
                    summary['expr'] = 'dynamic'
                    curve = self._curve(param, 'expr', 'expr_timestep', n_frames)
                    batch['expr'] = torch.from_numpy(curve.astype(np.float32)).to(self.device)[None]
            else:
                summary['pitch'] = 'ignored'
        if self.model.predict_variances:
            for name in self.model.variance_prediction_list:
                wanted = (self.auto_completion_mode and param.get(name) is None) or name in self.variance_prediction_set
                summary[name] = 'auto' if wanted else 'ignored'
        if verbose:
            print(f'[{idx}]\t' + ', '.join(f'{k}: {v}' for k, v in summary.items()))
        return batch

    # -- model ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_model(self, sample):
        """-> (durations aligned to the words [1, T_ph] | None, pitch in MIDI [1, T] | None, {name: [1, T]})"""
        kwargs = dict(ph_spk_mix_embed=None, spk_mix_embed=None)
        if hparams['use_spk_id']:
            for key in ('ph_spk_mix', 'spk_mix'):
                table = self.model.spk_embed(sample[key + '_id'])                       # [1, 1, N, H]
                kwargs[key + '_embed'] = torch.sum(table * sample[key + '_value'].unsqueeze(3), dim=2, keepdim=False)
        names = ('ph_dur', 'mel2ph', 'word_dur', 'note_midi', 'note_rest', 'note_dur', 'note_glide', 'mel2note', 'base_pitch')
        dur, pitch, variances = self.model(
            sample['tokens'], languages=sample.get('languages'), midi=sample['midi'], ph2word=sample['ph2word'],
            pitch=sample.get('pitch'), pitch_expr=sample.get('expr'), infer=True, **{k: sample[k] for k in names}, **kwargs)
        if dur is not None:
            dur = self.rr(dur, sample['ph2word'], sample['word_dur'])
        if pitch is not None:
            pitch = sample['base_pitch'] + pitch
        return dur, pitch, variances

    # -- several segments in one launch (ragged batch) --------------------------------------------------------------
    def _draw_noises(self, param, seed, flag, t_len):
        """x_T of the two loops for one segment, drawn as the model draws them when the segment runs alone."""
        if 'seed' in param:
            self._seed(param['seed'])
        elif seed >= 0:
            self._seed(seed)
        out = {}
        if flag[1]:
            d = self.model.pitch_predictor
            out['pitch_noise'] = torch.randn(1, d.num_feats, d.out_dims, t_len, device=self.device)
        if flag[2]:
            d = self.model.variance_predictor
            out['variance_noise'] = torch.randn(1, d.num_feats, d.out_dims, t_len, device=self.device)
        return out

    @staticmethod
    def _signature(flag, batch):
        """Segments can share a launch when the same predictors run on the same kinds of inputs."""
        return flag, tuple(sorted(k for k, v in batch.items() if v is not None))

    @torch.no_grad()
    def forward_model_batch(self, samples, noises):
        """`forward_model` for several segments with one signature: inputs padded to the longest (notes with the padding
        marker -1), frame counts handed on as `lengths` (dsd_set_lengths) -> per segment what `forward_model` returns."""
        pad = torch.nn.functional.pad
        lens = [int(s['base_pitch'].size(1)) for s in samples]
        t_max = max(lens)

        def cat(key, value=0, t_axis=False):
            if samples[0].get(key) is None:
                return None
            rows = []
            for s, t in zip(samples, lens):
                v = s[key]
                if t_axis and v.size(1) == 1 and t > 1:            # a constant given as [1, 1]
                    v = v.expand(-1, t, *v.shape[2:])
                rows.append(v)
            width = max(int(r.size(1)) for r in rows)
            return torch.cat([pad(r, [0, 0] * (r.dim() - 2) + [0, width - r.size(1)], value=value) for r in rows])

        kwargs = dict(ph_spk_mix_embed=None, spk_mix_embed=None)
        if hparams['use_spk_id']:
            for key, n_of in (('ph_spk_mix', lambda s: int(s['tokens'].size(1))), ('spk_mix', lambda s: int(s['base_pitch'].size(1)))):
                rows = []
                for s in samples:
                    table = self.model.spk_embed(s[key + '_id'])
                    e = torch.sum(table * s[key + '_value'].unsqueeze(3), dim=2, keepdim=False)
                    rows.append(e.expand(-1, n_of(s), -1) if e.size(1) == 1 else e)
                width = max(int(r.size(1)) for r in rows)
                kwargs[key + '_embed'] = torch.cat([pad(r, [0, 0, 0, width - r.size(1)]) for r in rows])
        noise = {k: torch.cat([pad(z[k], [0, t_max - z[k].size(-1)]) for z in noises]) for k in noises[0]}
        dur, pitch, variances = self.model(
            cat('tokens'), languages=cat('languages'), midi=cat('midi'), ph2word=cat('ph2word'), ph_dur=cat('ph_dur'),
            mel2ph=cat('mel2ph'), word_dur=cat('word_dur'), note_midi=cat('note_midi', value=-1.), note_rest=cat('note_rest'),
            note_dur=cat('note_dur'), note_glide=cat('note_glide'), mel2note=cat('mel2note'), base_pitch=cat('base_pitch'),
            pitch=cat('pitch'), pitch_expr=cat('expr', t_axis=True), infer=True, lengths=lens, **kwargs, **noise)
        out = []
        for i, (s, t) in enumerate(zip(samples, lens)):
            n_ph = int(s['tokens'].size(1))
            d = None if dur is None else self.rr(dur[i:i + 1, :n_ph], s['ph2word'], s['word_dur'])
            p = None if pitch is None else s['base_pitch'] + pitch[i:i + 1, :t]
            out.append((d, p, {k: v[i:i + 1, :t] for k, v in variances.items()}))
        return out

    def infer_once(self, param):
        dur, pitch, variances = self.forward_model(self.preprocess_input(param))
        dur = None if dur is None else dur[0].cpu().numpy()
        f0 = None if pitch is None else midi_to_hz(pitch[0].cpu().numpy())
        return dur, f0, {k: v[0].cpu().numpy() for k, v in variances.items()}

    # -- a whole project --------------------------------------------------------------------------------------
    def _flags(self, param):
        """(durations, pitch, variances): which predictors run for this segment."""
        m = self.model
        if self.auto_completion_mode:
            return (m.fs2.predict_dur and param.get('ph_dur') is None,
                    m.predict_pitch and param.get('f0_seq') is None,
                    m.predict_variances and any(param.get(v) is None for v in m.variance_prediction_list))
        variances = m.predict_variances and self.global_predict_variances
        pitch = m.predict_pitch and (self.global_predict_pitch or (param.get('f0_seq') is None and variances))
        dur = m.predict_dur and (self.global_predict_dur or (param.get('ph_dur') is None and (pitch or variances)))
        return dur, pitch, variances

    def _with_flags(self, flag, fn):
        m = self.model
        saved = (m.fs2.predict_dur, m.predict_pitch, m.predict_variances)
        m.fs2.predict_dur, m.predict_pitch, m.predict_variances = flag
        try:
            return fn()
        finally:
            m.fs2.predict_dur, m.predict_pitch, m.predict_variances = saved

    def run_inference(self, params: List[dict], out_dir=None, title: str = None, num_runs: int = 1, seed: int = -1,
                      batch_size: int = 1):
        """-> the completed projects (one list of segments per run); written to `out_dir/title[-NNN].ds` when given.
        `batch_size` > 1: segments on which the same predictors run on the same kinds of inputs share a launch as a ragged
        batch (same predictions as one by one, which is the reference's order and the default)."""
        flags = [self._flags(p) for p in params]
        batches = [self.preprocess_input(p, idx=i, load_dur=not f[0] and (f[1] or f[2]), load_pitch=not f[1] and f[2])
                   for i, (p, f) in enumerate(zip(params, flags))]
        runs = []
        for run in range(num_runs):
            ready = {}
            if batch_size > 1 and hasattr(self.model, 'fs2') and hasattr(self.model.fs2, 'native_handle'):
                groups = {}
                for i, (f, b) in enumerate(zip(flags, batches)):
                    groups.setdefault(self._signature(f, b), []).append(i)
                for (flag, _), members in groups.items():
                    members.sort(key=lambda i: -int(batches[i]['base_pitch'].size(1)))       # similar lengths together
                    for k in range(0, len(members), batch_size):
                        part = members[k:k + batch_size]
                        if len(part) < 2:
                            continue
                        noises = [self._draw_noises(params[i], seed, flag, int(batches[i]['base_pitch'].size(1))) for i in part]
                        got = self._with_flags(flag, lambda: self.forward_model_batch([batches[i] for i in part], noises))
                        ready.update(zip(part, got))
            results = []
            for i, (param, flag, batch) in enumerate(zip(params, flags, batches)):
                done = copy.deepcopy(param)
                if i in ready:
                    dur, pitch, variances = ready[i]
                else:
                    if 'seed' in param:
                        self._seed(param['seed'])
                    elif seed >= 0:
                        self._seed(seed)
                    dur, pitch, variances = self._with_flags(flag, lambda: self.forward_model(batch))
                if dur is not None and (self.auto_completion_mode or self.global_predict_dur):
                    seconds = (dur[0].cpu().numpy() * self.timestep).tolist()
                    done['ph_dur'] = ' '.join(str(round(d, 6)) for d in seconds)
                if pitch is not None and (self.auto_completion_mode or self.global_predict_pitch):
                    f0 = midi_to_hz(pitch[0].cpu().numpy())
                    done['f0_seq'] = ' '.join(str(round(freq, 1)) for freq in f0.tolist())
                    done['f0_timestep'] = str(self.timestep)
                for name, curve in variances.items():
                    if (self.auto_completion_mode and param.get(name) is None) or name in self.variance_prediction_set:
                        done[name] = ' '.join(str(round(v, 4)) for v in curve[0].cpu().numpy().tolist())
                        done[f'{name}_timestep'] = str(self.timestep)
                # speaker mixes that load_speaker_mix rewrote go back to what the segment carried
                if 'ph_spk_mix' in done and 'spk_mix' in done:
                    for key in ('ph_spk_mix', 'spk_mix'):
                        if key + '_backup' in done:
                            if done[key + '_backup'] is None:
                                del done[key]
                            else:
                                done[key] = done[key + '_backup']
                            del done[key + '_backup']
                results.append(done)
            runs.append(results)
            if out_dir is not None:
                out_dir = pathlib.Path(out_dir)
                out_dir.mkdir(parents=True, exist_ok=True)
                name = f'{title}-{str(run).zfill(3)}.ds' if num_runs > 1 else f'{title}.ds'
                with open(out_dir / name, 'w', encoding='utf8') as f:
                    json.dump(results, f, ensure_ascii=False, indent=2)
        return runs
