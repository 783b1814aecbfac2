"""`.ds` segments in, waveform out: the acoustic inference harness around the HIP modules
(SURVEY.md section 8(f) rank 4).  Behavioural twin of the reference's `inference/ds_acoustic.py`,
`basics/base_svs_infer.py:38-131` (speaker mix), `utils/infer_utils.py:41-56,99-118` (curve resampling, cross-fade,
wav writing), `modules/fastspeech/tts_modules.py:278-311` (length regulator) and the `--depth/--steps` handling of
`scripts/infer.py:168-198` (= the `depth` / `steps` runtime inputs of the ONNX export,
`deployment/modules/diffusion.py:105-131`).

Host-side glue only: wire format, resampling, length regulation, speaker mixing, per-segment seeding, cross-fade
and wav writing, so that a caller does not need the reference checkout at run time.  The phoneme dictionary is
data handling outside this package: any object with `encode(ph_seq, lang=None) -> list[int]`, `__len__()` and
`is_cross_lingual(phone) -> bool` works (the reference's `utils.phoneme_utils.PhonemeDictionary` is one;
`SimplePhonemeTable` covers single-dictionary models).  Segments are run one by one, as the reference does:
utterance-level parallelism is across GPUs (sharding.py).
"""
from __future__ import annotations

import json
import pathlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .hparams import hparams

VARIANCE_NAMES = ('energy', 'breathiness', 'voicing', 'tension')     # param_adaptor.py: VARIANCE_CHECKLIST order


# --------------------------------------------------------------------------------------------------- depth / steps
def apply_depth_steps(hp: dict, depth: Optional[float] = None, steps: Optional[int] = None) -> dict:
    """Fill the reflow-era keys from the DDPM-era ones (and back), then apply the two runtime knobs:
    `depth` in (0, 1] = how much of the diffusion is run from the aux mel, `steps` = sampling steps."""
    shallow = bool(hp['use_shallow_diffusion'])
    if 'diff_speedup' not in hp and 'pndm_speedup' in hp:
        hp['diff_speedup'] = hp['pndm_speedup']
    for new_key, old_key in (('T_start', 'K_step'), ('T_start_infer', 'K_step_infer')):
        if new_key not in hp:
            hp[new_key] = 1 - hp[old_key] / hp['timesteps']
    if 'sampling_steps' not in hp:
        hp['sampling_steps'] = (hp['K_step_infer'] if shallow else hp['timesteps']) // hp['diff_speedup']
    if 'time_scale_factor' not in hp:
        hp['time_scale_factor'] = hp['timesteps']
    if depth is not None:
        limit = 1 - hp['T_start']
        assert depth <= limit, f"Depth should not be larger than 1 - T_start ({limit})"
        hp['K_step_infer'], hp['T_start_infer'] = round(hp['timesteps'] * depth), 1 - depth
    if steps is not None:
        if shallow and 'K_step_infer' in hp:
            hp['diff_speedup'] = round((1 - hp['T_start_infer']) / steps * hp['K_step_infer'])
        elif not shallow and 'timesteps' in hp:
            hp['diff_speedup'] = round(hp['timesteps'] / steps)
        hp['sampling_steps'] = steps
    return hp


# ------------------------------------------------------------------------------------------------ signal helpers
def resample_align_curve(points: np.ndarray, original_timestep: float, target_timestep: float, align_length: int):
    """Linear resampling of a control curve to the model's frame rate, then cut / hold-last to `align_length`."""
    src_t = original_timestep * np.arange(len(points))
    dst_t = np.arange(0, (len(points) - 1) * original_timestep, target_timestep)
    curve = np.interp(dst_t, src_t, points).astype(points.dtype)
    if len(curve) >= align_length:
        return curve[:align_length]
    return np.concatenate((curve, np.full(align_length - len(curve), fill_value=curve[-1])), axis=0)


def cross_fade(a: np.ndarray, b: np.ndarray, idx: int):
    """`b` starts at sample `idx` of `a`; the overlap is blended with a linear ramp (end points included)."""
    overlap = a.shape[0] - idx
    ramp = np.linspace(0, 1.0, num=overlap, endpoint=True)
    out = np.zeros(idx + b.shape[0])
    out[:idx] = a[:idx]
    out[idx:a.shape[0]] = (1 - ramp) * a[idx:] + ramp * b[:overlap]
    out[a.shape[0]:] = b[overlap:]
    return out


def save_wav(wav, path, sr, norm=False):
    from scipy.io import wavfile
    scaled = (wav / np.abs(wav).max() if norm else wav) * 32767
    wavfile.write(path, sr, scaled.astype(np.int16))


def length_regulator(dur: torch.Tensor, dur_padding: Optional[torch.Tensor] = None, alpha: Optional[float] = None):
    """durations [B, T_txt] -> mel2ph [B, T]: frame t carries the 1-based index of the token whose cumulative
    duration first exceeds t, 0 past the end of its utterance."""
    assert alpha is None or alpha > 0
    if alpha is not None:
        dur = torch.round(dur.float() * alpha).long()
    if dur_padding is not None:
        dur = dur * (1 - dur_padding.long())
    ends = torch.cumsum(dur, 1).contiguous()
    frames = torch.arange(int(ends[:, -1].max()), device=dur.device)[None, :].expand(dur.shape[0], -1).contiguous()
    owner = torch.searchsorted(ends, frames, right=True) + 1
    return torch.where(frames < ends[:, -1:], owner, torch.zeros_like(owner))


# --------------------------------------------------------------------------------------------------- checkpoints
def load_ckpt(model, ckpt_base_dir, ckpt_steps: Optional[int] = None, prefix_in_ckpt: Optional[str] = 'model',
              ignored_prefixes: Optional[Sequence[str]] = None, key_in_ckpt: Optional[str] = 'state_dict', strict: bool = True,
              device='cpu') -> pathlib.Path:
    """The reference's checkpoint convention (`utils/__init__.py:165-222`): a `.ckpt` file, or a work directory whose
    `model_ckpt_steps_<N>.ckpt` with the highest N (or N = `ckpt_steps`) is taken; the weights sit under `state_dict`
    with a `model.` prefix.  Read with `torch.load(weights_only=True)`: nothing from the file is executed.  -> the path."""
    import re
    from collections import OrderedDict
    base = pathlib.Path(ckpt_base_dir)
    ignored = ['model.fs2.encoder.embed_tokens'] if ignored_prefixes is None else list(ignored_prefixes)   # old duplicates
    if base.is_file():
        path = base
    elif ckpt_steps is not None:
        path = base / f'model_ckpt_steps_{int(ckpt_steps)}.ckpt'
    else:
        found = sorted((f for f in base.iterdir() if f.is_file() and re.fullmatch(r'model_ckpt_steps_\d+\.ckpt', f.name)),
                       key=lambda f: int(re.search(r'\d+', f.name).group(0)))
        assert len(found) > 0, f'| ckpt not found in {base}.'
        path = found[-1]
    loaded = torch.load(path, map_location=device, weights_only=True)
    category = getattr(model, 'category', None)
    if category is not None and isinstance(loaded, dict) and loaded.get('category') not in (None, category):
        raise RuntimeError(f"Category mismatches! The checkpoint is of the category '{loaded.get('category')}', "
                           f"but a checkpoint of category '{category}' is required.")
    state = loaded if key_in_ckpt is None else loaded[key_in_ckpt]
    if prefix_in_ckpt is not None:
        state = OrderedDict((k[len(prefix_in_ckpt) + 1:], v) for k, v in state.items()
                            if k.startswith(f'{prefix_in_ckpt}.') and not any(k.startswith(p) for p in ignored))
    if not strict:
        own = model.state_dict()
        for key in [k for k, v in state.items() if k in own and own[k].shape != v.shape]:
            del state[key]
    model.load_state_dict(state, strict=strict)
    return path


def load_vocoder(model_path, device='cuda'):
    """`modules/nsf_hifigan/models.py:18-33` + `modules/vocoders/nsf_hifigan.py:18-37`: `config.json` next to the generator
    checkpoint, weights under 'generator' (weight norm folded on load) -> vocoder.NsfHifiGAN."""
    from .vocoder import Generator, NsfHifiGAN
    model_path = pathlib.Path(model_path)
    with open(model_path.with_name('config.json')) as f:
        h = json.load(f)
    gen = Generator(h)
    gen.load_state_dict(torch.load(model_path, map_location='cpu', weights_only=True)['generator'], strict=True)
    return NsfHifiGAN(gen.to(device).eval(), mel_base=hparams.get('mel_base', '10'))


class SimplePhonemeTable:
    """Single-dictionary phoneme table: ids 1..N over the sorted phoneme set (AP and SP always present), 0 = padding."""

    def __init__(self, phonemes: Sequence[str]):
        names = sorted(set(phonemes) | {'AP', 'SP'})
        self._ids = {name: i + 1 for i, name in enumerate(names)}

    def __len__(self):
        return len(self._ids) + 1

    def is_cross_lingual(self, phone):
        return False

    def encode(self, sentence, lang=None):
        return [self._ids[p.split('/', maxsplit=1)[-1]] for p in sentence.strip().split()]


# ------------------------------------------------------------------------------------------------- speaker mix
def _mix_track(values, name, param, timestep, length, mode, device):
    """One speaker's proportion over the `length` tokens / frames: a constant, or a curve given as a string."""
    if not isinstance(values, str):
        assert values >= 0., f'Speaker mix checks failed.\nProportion of speaker \'{name}\' is negative.'
        return torch.full((1, length), fill_value=values, dtype=torch.float32, device=device)
    if mode == 'token':
        items = values.split()
        assert len(items) == length, ('Speaker mix checks failed. In dynamic token-level mix, '
                                      'number of proportion values must equal number of tokens.')
        track = np.array(items, 'float32')
    else:
        track = resample_align_curve(np.array(values.split(), 'float32'), original_timestep=float(param['spk_mix_timestep']),
                                     target_timestep=timestep, align_length=length)
    track = torch.from_numpy(track).to(device)[None]
    assert torch.all(track >= 0.), (f'Speaker mix checks failed.\nProportions of speaker \'{name}\' on some {mode}s '
                                    f'are negative.')
    return track


def load_speaker_mix(param_src: dict, summary_dst: dict, spk_map: Dict[str, int], timestep: float, device,
                     mix_mode: str = 'frame', mix_length: int = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> spk_mix_id [1, 1, N], spk_mix_value [1, T or 1, N] (normalised to sum 1 over the N speakers)."""
    assert mix_mode in ('token', 'frame')
    key = 'spk_mix' if mix_mode == 'frame' else 'ph_spk_mix'
    mix = param_src.get(key)
    if mix is None:
        assert len(spk_map) == 1, "This is a multi-speaker model. Please specify a speaker or speaker mix by --spk option."
        mix = {next(iter(spk_map)): 1.0}
    for name in mix:
        assert name in spk_map, f'Speaker \'{name}\' not found.'
    dynamic = len(mix) > 1 and any(isinstance(v, str) for v in mix.values())
    if len(mix) == 1:
        summary_dst['spk' if mix_mode == 'frame' else 'ph_spk'] = next(iter(mix))
    elif dynamic:
        summary_dst[key] = f"dynamic({'|'.join(mix)})"
    else:
        summary_dst[key] = 'static(' + '|'.join(f'{n}:{mix[n]:.3f}' for n in mix) + ')'
    ids = torch.LongTensor([spk_map[n] for n in mix]).to(device)[None, None]
    if dynamic:
        weights = torch.stack([_mix_track(v, n, param_src, timestep, mix_length, mix_mode, device) for n, v in mix.items()],
                              dim=2)
        total = weights.sum(dim=2, keepdim=True)
        assert torch.all(total > 0.), 'Speaker mix checks failed.\nProportions of speaker mix on some frames sum to zero.'
    else:
        for n, v in mix.items():
            assert v >= 0., f'Speaker mix checks failed.\nProportion of speaker \'{n}\' is negative.'
        weights = torch.FloatTensor(list(mix.values())).to(device)[None, None]
        total = weights.sum()
        assert total > 0., 'Speaker mix checks failed.\nProportions of speaker mix sum to zero.'
    return ids, weights / total


def load_ds(path) -> List[dict]:
    """A `.ds` project: a JSON list of segments (or one segment object)."""
    with open(path, 'r', encoding='utf8') as f:
        params = json.load(f)
    return params if isinstance(params, list) else [params]


# ----------------------------------------------------------------------------------------------------- harness
class AcousticHarness:
    """`model` = diffsinger_amd.toplevel.DiffSingerAcoustic (or any module with the reference's forward signature),
    `vocoder` = diffsinger_amd.vocoder.NsfHifiGAN or None (mel output only)."""

    def __init__(self, model, vocoder, phoneme_dictionary, spk_map: Optional[Dict[str, int]] = None,
                 lang_map: Optional[Dict[str, int]] = None, device=None):
        self.model, self.vocoder, self.phoneme_dictionary = model, vocoder, phoneme_dictionary
        self.spk_map, self.lang_map = spk_map or {}, lang_map or {}
        self.device = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
        self.timestep = hparams['hop_size'] / hparams['audio_sample_rate']
        self.variances_to_embed = {v for v in VARIANCE_NAMES if hparams.get(f'use_{v}_embed', False)}
        if hparams['use_spk_id']:
            assert isinstance(self.spk_map, dict) and len(self.spk_map) > 0, 'Invalid or empty speaker map!'
            assert len(self.spk_map) == len(set(self.spk_map.values())), 'Duplicate speaker id in speaker map!'

    # -- one segment -> model inputs --------------------------------------------------------------------------
    def _curve(self, param, key, step_key, length):
        return resample_align_curve(np.array(param[key].split(), np.float32), original_timestep=float(param[step_key]),
                                    target_timestep=self.timestep, align_length=length)

    def _tokens(self, param, batch):
        lang = param.get('lang')
        if lang is None:
            assert len(self.lang_map) <= 1, "This is a multilingual model. Please specify a language by --lang option."
        else:
            assert lang in self.lang_map, f'Unrecognized language name: \'{lang}\'.'
        phones = param['ph_seq'].split()
        if hparams.get('use_lang_id', False):
            def lang_id(p):
                if not self.phoneme_dictionary.is_cross_lingual(p):
                    return 0
                return self.lang_map[p.split('/', maxsplit=1)[0] if '/' in p else lang]
            batch['languages'] = torch.LongTensor([lang_id(p) for p in phones]).to(self.device)
        batch['tokens'] = torch.LongTensor([self.phoneme_dictionary.encode(param['ph_seq'], lang=lang)]).to(self.device)

    def _frames(self, param, batch):
        """Phoneme durations in seconds -> integer frame counts by rounding the cumulative boundaries, -> mel2ph."""
        seconds = torch.from_numpy(np.array(param['ph_dur'].split(), np.float32)).to(self.device)
        bounds = torch.round(torch.cumsum(seconds, dim=0) / self.timestep + 0.5).long()
        frames = torch.diff(bounds, dim=0, prepend=torch.zeros(1, dtype=torch.long, device=self.device))[None]
        batch['mel2ph'] = length_regulator(frames, batch['tokens'] == 0)
        return batch['mel2ph'].size(1)

    def _key_shift(self, param, length):
        lo, hi = hparams['augmentation_args']['random_pitch_shifting']['range']
        gender = param.get('gender', None)
        gender = 0. if gender is None else gender
        if isinstance(gender, (int, float, bool)):
            value = gender * hi if gender >= 0 else gender * abs(lo)
            return torch.FloatTensor([value]).to(self.device)[:, None], f'static({gender:.3f})'
        curve = self._curve(param, 'gender', 'gender_timestep', length)
        up = curve >= 0
        shift = curve * (up * hi + (1 - up) * abs(lo))
        return torch.clip(torch.from_numpy(shift.astype(np.float32)).to(self.device)[None], min=lo, max=hi), 'dynamic'

    def _speed(self, param, length):
        if param.get('velocity') is None:
            return torch.FloatTensor([1.]).to(self.device)[:, None], 'default'
        lo, hi = hparams['augmentation_args']['random_time_stretching']['range']
        curve = self._curve(param, 'velocity', 'velocity_timestep', length)
        return torch.clip(torch.from_numpy(curve.astype(np.float32)).to(self.device)[None], min=lo, max=hi), 'manual'

    def preprocess_input(self, param: dict, idx: int = 0, verbose: bool = False) -> Dict[str, torch.Tensor]:
        batch, summary = {}, {}
        self._tokens(param, batch)
        length = self._frames(param, batch)
        summary.update(tokens=batch['tokens'].size(1), frames=length, seconds='%.2f' % (length * self.timestep))
        if hparams['use_spk_id']:
            batch['spk_mix_id'], batch['spk_mix_value'] = load_speaker_mix(
                param, summary, self.spk_map, self.timestep, self.device, mix_mode='frame', mix_length=length)
        batch['f0'] = torch.from_numpy(self._curve(param, 'f0_seq', 'f0_timestep', length)).to(self.device)[None]
        for name in VARIANCE_NAMES:
            if name in self.variances_to_embed:
                batch[name] = torch.from_numpy(self._curve(param, name, f'{name}_timestep', length)).to(self.device)[None]
                summary[name] = 'manual'
        if hparams['use_key_shift_embed']:
            batch['key_shift'], summary['gender'] = self._key_shift(param, length)
        if hparams['use_speed_embed']:
            batch['speed'], summary['velocity'] = self._speed(param, length)
        if verbose:
            print(f'[{idx}]\t' + ', '.join(f'{k}: {v}' for k, v in summary.items()))
        return batch

    # -- model, vocoder ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_model(self, sample):
        kwargs = {v: sample.get(v) for v in self.variances_to_embed}
        if hparams['use_spk_id']:
            table = self.model.fs2.spk_embed(sample['spk_mix_id'])                       # [1, 1, N, H]
            kwargs['spk_mix_embed'] = torch.sum(table * sample['spk_mix_value'].unsqueeze(3), dim=2, keepdim=False)
        else:
            kwargs['spk_mix_embed'] = None
        out = self.model(sample['tokens'], languages=sample.get('languages'), mel2ph=sample['mel2ph'], f0=sample['f0'],
                         key_shift=sample.get('key_shift'), speed=sample.get('speed'), infer=True, **kwargs)
        return out.diff_out

    @torch.no_grad()
    def run_vocoder(self, spec, **kwargs):
        return self.vocoder.spec2wav_torch(spec, **kwargs)[None]

    # -- a whole project --------------------------------------------------------------------------------------
    @staticmethod
    def _seed(value):
        torch.manual_seed(value & 0xffff_ffff)
        torch.cuda.manual_seed_all(value & 0xffff_ffff)

    # -- several segments in one launch (ragged batch) --------------------------------------------------------------
    def _draw_noise(self, param, seed, t_len):
        """x_T of one segment, drawn exactly as the model would draw it when run alone (same seeding, same call)."""
        if 'seed' in param:
            self._seed(param['seed'])
        elif seed >= 0:
            self._seed(seed)
        d = self.model.diffusion
        return torch.randn(1, d.num_feats, d.out_dims, t_len, device=self.device)

    @torch.no_grad()
    def forward_model_batch(self, samples, noises):
        """`forward_model` for several segments at once: inputs zero-padded to the longest, per-segment lengths handed to
        the library (dsd_set_lengths), so every mel equals the one the segment gives alone.  -> list of [1, T_i, M]."""
        lens = [int(s['mel2ph'].size(1)) for s in samples]
        t_max, n = max(lens), len(samples)

        def pad_t(v, t_len):                         # [1, T_i or 1, ...] -> [1, t_max, ...]
            if v.size(1) == 1 and t_len > 1:
                v = v.expand(-1, t_len, *v.shape[2:])
            return torch.nn.functional.pad(v, [0, 0] * (v.dim() - 2) + [0, t_max - v.size(1)])

        l_max = max(int(s['tokens'].size(1)) for s in samples)
        cat = lambda key, width: torch.cat([torch.nn.functional.pad(s[key], [0, width - s[key].size(1)]) for s in samples])  # noqa: E731
        kwargs = {v: torch.cat([pad_t(s[v], t) for s, t in zip(samples, lens)]) for v in self.variances_to_embed}
        if hparams['use_spk_id']:
            mixes = []
            for s, t in zip(samples, lens):
                table = self.model.fs2.spk_embed(s['spk_mix_id'])
                mixes.append(pad_t(torch.sum(table * s['spk_mix_value'].unsqueeze(3), dim=2, keepdim=False), t))
            kwargs['spk_mix_embed'] = torch.cat(mixes)
        else:
            kwargs['spk_mix_embed'] = None
        for key in ('key_shift', 'speed'):
            kwargs[key] = torch.cat([pad_t(s[key], t) for s, t in zip(samples, lens)]) if samples[0].get(key) is not None else None
        languages = cat('languages', l_max) if samples[0].get('languages') is not None else None
        noise = torch.cat([torch.nn.functional.pad(z, [0, t_max - z.size(-1)]) for z in noises])
        out = self.model(cat('tokens', l_max), languages=languages, mel2ph=cat('mel2ph', t_max), f0=cat('f0', t_max),
                         infer=True, noise=noise, lengths=lens, **kwargs).diff_out
        return [out[i:i + 1, :lens[i]] for i in range(n)]

    def _batchable(self):
        """A ragged batch reproduces the one-by-one results when x_T is the sampler's only random draw."""
        d = getattr(self.model, 'diffusion', None)
        if d is None or not hasattr(self.model, 'fs2'):
            return False
        if hasattr(d, 'velocity_fn'):
            return True
        return hparams.get('diff_accelerator') in ('ddim', 'pndm', 'dpm-solver', 'unipc') and hparams.get('diff_speedup', 1) > 1

    def run_inference(self, params: List[dict], out_path=None, seed: int = -1, save_mel: bool = False, batch_size: int = 1,
                      out_dir=None, title: Optional[str] = None, num_runs: int = 1):
        """One pass over the segments of a project: returns the assembled waveform (or the list of mels) and, when
        `out_path` is given, writes it.  Each segment is placed at its `offset`; where it overlaps what is already
        there the two are cross-faded.  `batch_size` > 1 runs that many segments per launch of the acoustic model as
        a ragged batch (same mels as one by one - the reference's order of operations, `ds_acoustic.py:214-271`,
        `batch_size` = 1, is the default); the vocoder still takes them one at a time.
        With `out_dir` and `title` the reference's own calling convention applies: `num_runs` passes, written to
        `out_dir/title[-NNN].wav` (or `.mel.pt` with `save_mel`); the last pass is returned."""
        if out_dir is not None:
            assert title is not None, 'run_inference(out_dir=...) needs a title'
            suffix = '.mel.pt' if save_mel else '.wav'
            result = None
            for run in range(num_runs):
                name = f'{title}-{str(run).zfill(3)}{suffix}' if num_runs > 1 else title + suffix
                result = self.run_inference(params, out_path=pathlib.Path(out_dir) / name, seed=seed, save_mel=save_mel,
                                            batch_size=batch_size)
            return result
        batches = [self.preprocess_input(param, idx=i) for i, param in enumerate(params)]
        ready = {}
        if batch_size > 1 and self._batchable():
            noises = [self._draw_noise(p, seed, int(b['mel2ph'].size(1))) for p, b in zip(params, batches)]
            order = sorted(range(len(params)), key=lambda i: -int(batches[i]['mel2ph'].size(1)))     # similar lengths together
            for k in range(0, len(order), batch_size):
                group = order[k:k + batch_size]
                mels_g = self.forward_model_batch([batches[i] for i in group], [noises[i] for i in group])
                ready.update(zip(group, mels_g))
        mels, track, cursor = [], np.zeros(0), 0
        for i, (param, batch) in enumerate(zip(params, batches)):
            if i in ready:
                mel = ready[i]
                if not save_mel:        # leave the generator where a lone run of this segment leaves it for the vocoder's draws
                    self._draw_noise(param, seed, int(batch['mel2ph'].size(1)))
            else:
                if 'seed' in param:
                    self._seed(param['seed'])
                elif seed >= 0:
                    self._seed(seed)
                mel = self.forward_model(batch)
            if save_mel:
                mels.append({'offset': param.get('offset', 0.), 'mel': mel.cpu(), 'f0': batch['f0'].cpu()})
                continue
            wav = self.run_vocoder(mel, f0=batch['f0'])[0].cpu().numpy()
            gap = round(param.get('offset', 0) * hparams['audio_sample_rate']) - cursor
            if gap >= 0:
                track = np.concatenate((track, np.zeros(gap), wav))
            else:
                track = cross_fade(track, wav, cursor + gap)
            cursor += gap + wav.shape[0]
        result = mels if save_mel else track
        if out_path is not None:
            out_path = pathlib.Path(out_path)
            out_path.parent.mkdir(parents=True, exist_ok=True)
            if save_mel:
                torch.save(result, out_path)
            else:
                save_wav(result, out_path, hparams['audio_sample_rate'])
        return result


class PhonemeDictionary:
    """The phoneme <-> token-id table of a (possibly multilingual) model, built like `utils/phoneme_utils.py:10-176` from
    the pronunciation dictionaries (`word<TAB>ph ph ...` per line), the extra phonemes and the merged phoneme groups of the
    configuration: AP and SP always exist; with more than one language a dictionary phoneme is named `lang/ph` (an extra
    phoneme keeps the name it is given); ids run from 1 over the SORTED names, and the phonemes of a merged group share
    the id of the first of them met in that order; a phoneme is cross-lingual when its group spans languages."""

    def __init__(self, dictionaries: Dict[str, "pathlib.Path"], extra_phonemes: Optional[Sequence[str]] = None,
                 merged_groups: Optional[Sequence[Sequence[str]]] = None):
        names = {'AP', 'SP'}
        for ph in extra_phonemes or ():
            if '/' in ph:
                lang, short = ph.split('/', maxsplit=1)
                if lang not in dictionaries:
                    raise ValueError(f"Invalid phoneme tag '{ph}' in extra phonemes: unrecognized language name '{lang}'.")
                if short in names:
                    raise ValueError(f"Invalid phoneme tag '{ph}' in extra phonemes: short name conflicts with existing tag.")
            names.add(ph)
        self._multi_langs = len(dictionaries) > 1
        for lang, path in dictionaries.items():
            with open(path, 'r', encoding='utf8') as f:
                for line in f:
                    _, phones = line.strip().split('\t')
                    for ph in phones.split():
                        if '/' in ph:
                            raise ValueError(f"Invalid phoneme tag '{ph}' in dictionary '{path}': "
                                             f"should not contain the reserved character '/'.")
                        if ph not in names:
                            names.add(f'{lang}/{ph}' if self._multi_langs else ph)
        # merged groups: overlapping groups fuse (union-find over the phoneme names)
        parent: Dict[str, str] = {}

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for group in merged_groups or ():
            members = []
            for ph in group:
                if '/' in ph:
                    lang, short = ph.split('/', maxsplit=1)
                    if lang not in dictionaries:
                        raise ValueError(f"Invalid phoneme tag '{ph}' in merged group: unrecognized language name '{lang}'.")
                    member = ph if self._multi_langs else short
                else:
                    member = ph
                if member not in names:
                    raise ValueError(f"Invalid phoneme tag '{ph}' in merged group: not found in phoneme set.")
                members.append(member)
            if len(members) < 2:        # (a group that names one phoneme twice still counts as a group, as in the reference)
                continue
            for m in members:
                parent.setdefault(m, m)
            for m in members[1:]:
                parent[find(m)] = find(members[0])
        groups: Dict[str, List[str]] = {}
        for m in parent:
            groups.setdefault(find(m), []).append(m)
        self._phone_to_id: Dict[str, int] = {}
        self._id_to_phone: List = []
        cross = set()
        for ph in sorted(names):
            if ph in self._phone_to_id:
                continue
            idx = len(self._id_to_phone) + 1
            if ph in parent:
                members = sorted(groups[find(ph)])
                for alias in members:
                    self._phone_to_id[alias] = idx
                self._id_to_phone.append(tuple(members))
                if len({a.split('/', maxsplit=1)[0] if '/' in a else None for a in members}) > 1:
                    cross.update(a for a in members if '/' in a)
            else:
                self._phone_to_id[ph] = idx
                self._id_to_phone.append(ph)
        self._cross_lingual_phonemes = frozenset(cross)

    @property
    def vocab_size(self):
        return len(self._id_to_phone) + 1

    def __len__(self):
        return self.vocab_size

    @property
    def cross_lingual_phonemes(self):
        return self._cross_lingual_phonemes

    def is_cross_lingual(self, phone):
        return phone in self._cross_lingual_phonemes

    def encode_one(self, phone, lang=None):
        if '/' in phone:
            lang, phone = phone.split('/', maxsplit=1)
        if lang is None or not self._multi_langs or phone in self._phone_to_id:
            return self._phone_to_id[phone]
        return self._phone_to_id[phone if '/' in phone else f'{lang}/{phone}']

    def encode(self, sentence, lang=None):
        phones = sentence.strip().split() if isinstance(sentence, str) else sentence
        return [self.encode_one(p, lang=lang) for p in phones]

    def decode_one(self, idx, lang=None, scalar=True):
        if idx <= 0:
            return None
        phone = self._id_to_phone[idx - 1]
        if not scalar or isinstance(phone, str):
            return phone
        if lang is not None and self._multi_langs:
            for alias in phone:
                if alias.startswith(f'{lang}/'):
                    return alias
        return phone[0]

    def decode(self, ids, lang=None, scalar=True):
        return ' '.join(self.decode_one(i, lang=lang, scalar=scalar) for i in list(ids) if i >= 1)

    def dump(self, filename):
        with open(filename, 'w', encoding='utf8') as fp:
            json.dump(self._phone_to_id, fp, ensure_ascii=False, indent=2)


def load_phoneme_dictionary() -> PhonemeDictionary:
    """`utils/phoneme_utils.py:179-210`: `dictionary-<lang>.txt` in the work directory (else the configured path) for every
    language of `hparams['dictionaries']`, or the single `dictionary.txt` / `hparams['dictionary']`."""
    work = pathlib.Path(hparams['work_dir'])
    configured = hparams.get('dictionaries')
    if configured is not None:
        paths = {}
        for lang, fallback in configured.items():
            path = work / f'dictionary-{lang}.txt'
            path = path if path.exists() else pathlib.Path(fallback)
            if not path.exists():
                raise FileNotFoundError(f"Could not locate dictionary for language '{lang}'.")
            paths[lang] = path
    else:
        path = work / 'dictionary.txt'
        path = path if path.exists() else pathlib.Path(hparams['dictionary'])
        if not path.exists():
            raise FileNotFoundError("Could not locate dictionary file.")
        paths = {'default': path}
    return PhonemeDictionary(paths, extra_phonemes=hparams.get('extra_phonemes'),
                             merged_groups=hparams.get('merged_phoneme_groups'))


# --------------------------------------------------------------------------------------------------- project edits
_SHARP_NAMES = ('C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B')


def midi_to_note(midi) -> str:
    """60 -> 'C4', 61 -> 'C#4' (sharps, ASCII; C-1 = 0): the nearest semitone's name, like librosa.midi_to_note(unicode=False)."""
    n = int(round(float(midi)))
    return f'{_SHARP_NAMES[n % 12]}{n // 12 - 1}'


def trans_key(raw_data: List[dict], key: int) -> List[dict]:
    """`scripts/infer.py --key` (`utils/infer_utils.py:8-38`): every note of every segment moves by `key` semitones (names
    come back as sharps on the nearest semitone, rests stay), every value of `f0_seq` is multiplied by 2^(key/12) and
    rounded to 0.1 Hz.  Edits the segments in place and returns them."""
    from .variance_harness import note_to_midi
    missing = False
    for seg in raw_data:
        seg['note_seq'] = ' '.join(n if n == 'rest' else midi_to_note(int(round(note_to_midi(n))) + key)
                                   for n in seg['note_seq'].split(' '))
        if seg.get('f0_seq'):
            seg['f0_seq'] = ' '.join(str(round(float(v) * 2 ** (key / 12), 1)) for v in seg['f0_seq'].split(' '))
        else:
            missing = True
    if missing:
        print('Warning: parts of f0_seq do not exist, please freeze the pitch line in the editor.\r\n')
    return raw_data


def parse_commandline_spk_mix(mix: str) -> Dict[str, float]:
    """`scripts/infer.py --spk` (`utils/infer_utils.py:56-86`): "name", "a|b" or "a:0.3|b:0.5|c" -> proportions that sum to
    1: speakers without a number share what the given numbers leave of 1, then everything is normalised."""
    import re
    name, number = r'[0-9A-Za-z_-]+', r'\d+(\.\d+)?'
    single = rf'{name}(:{number})?'
    assert re.fullmatch(rf'{single}(\|{single})*', mix) is not None, f'Invalid mix pattern: {mix}'
    given, bare = {}, []
    for part in mix.split('|'):
        speaker = part.split(':')[0]
        assert speaker not in bare and speaker not in given, f'Duplicate speaker name: {speaker}'
        if ':' in part:
            given[speaker] = float(part.split(':')[1])
        else:
            bare.append(speaker)
    total = sum(given.values())
    assert total < 1 or len(bare) == 0, \
        'Proportion of all speakers should be specified if the sum of all given proportions are larger than 1.'
    for speaker in bare:
        given[speaker] = (1 - total) / len(bare)
    norm = sum(given.values())
    assert norm > 0, 'Sum of all proportions should be positive.'
    return {speaker: value / norm for speaker, value in given.items()}
