"""Host-side `.ds` harness against fixtures produced by the reference's own
`DiffSingerAcousticInfer.preprocess_input`, `LengthRegulator`, `resample_align_curve` and `cross_fade` (G11), plus the
`--depth/--steps` arithmetic of scripts/infer.py.  Integer outputs must match exactly, float32 ones bit for bit
(same numpy / torch CPU operations)."""
import json
import os

import numpy as np
import pytest
import torch

from diffsinger_amd import harness
from diffsinger_amd.hparams import hparams

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HP = dict(hop_size=512, audio_sample_rate=44100, use_spk_id=True, use_lang_id=False, use_energy_embed=True,
          use_key_shift_embed=True, use_speed_embed=True,
          augmentation_args=dict(random_pitch_shifting=dict(range=[-5.0, 5.0]), random_time_stretching=dict(range=[0.5, 2.0])))


@pytest.fixture
def hp():
    saved = dict(hparams)
    hparams.clear()
    hparams.update(HP)
    yield hparams
    hparams.clear()
    hparams.update(saved)


def test_preprocess_input_matches_reference(hp):
    g = np.load(os.path.join(GOLDEN, "g11_harness.npz"))
    segs = harness.load_ds(os.path.join(GOLDEN, "g11_segments.ds"))
    assert len(segs) == 3
    h = harness.AcousticHarness(None, None, harness.SimplePhonemeTable(["a", "b", "c", "d", "e"]),
                                spk_map={"alice": 0, "bob": 1, "carol": 2}, device="cpu")
    for i, seg in enumerate(segs):
        batch = h.preprocess_input(seg, idx=i)
        keys = {k[len(f"seg{i}_"):] for k in g.files if k.startswith(f"seg{i}_")}
        assert set(batch) == keys
        for k in keys:
            got, want = batch[k].numpy(), g[f"seg{i}_{k}"]
            assert got.shape == want.shape and got.dtype == want.dtype, (i, k)
            np.testing.assert_array_equal(got, want, err_msg=f"segment {i} {k}")
    # the mix proportions sum to one on every frame, padding-free mel2ph covers every token
    assert np.allclose(g["seg1_spk_mix_value"].sum(-1), 1.0, atol=1e-6)
    assert set(np.unique(g["seg1_mel2ph"])) == set(range(1, g["seg1_tokens"].shape[1] + 1))


def test_length_regulator_resample_cross_fade():
    g = np.load(os.path.join(GOLDEN, "g11_harness.npz"))
    dur, pad = torch.from_numpy(g["lr_dur"]), torch.from_numpy(g["lr_pad"]).bool()
    np.testing.assert_array_equal(harness.length_regulator(dur, pad).numpy(), g["lr_mel2ph"])
    np.testing.assert_array_equal(harness.length_regulator(dur, None, 1.3).numpy(), g["lr_mel2ph_alpha"])
    np.testing.assert_array_equal(harness.resample_align_curve(g["rs_points"], 0.01, 512 / 44100, 80), g["rs_long"])
    np.testing.assert_array_equal(harness.resample_align_curve(g["rs_points"], 0.01, 512 / 44100, 30), g["rs_short"])
    np.testing.assert_array_equal(harness.cross_fade(g["cf_a"], g["cf_b"], 820), g["cf_out"])


def test_speaker_mix_checks(hp):
    spk = {"alice": 0, "bob": 1}
    with pytest.raises(AssertionError, match="multi-speaker"):
        harness.load_speaker_mix({}, {}, spk, 0.0116, "cpu", mix_length=5)
    with pytest.raises(AssertionError, match="not found"):
        harness.load_speaker_mix({"spk_mix": {"zed": 1.0}}, {}, spk, 0.0116, "cpu", mix_length=5)
    with pytest.raises(AssertionError, match="negative"):
        harness.load_speaker_mix({"spk_mix": {"alice": -1.0, "bob": 2.0}}, {}, spk, 0.0116, "cpu", mix_length=5)
    with pytest.raises(AssertionError, match="sum to zero"):
        harness.load_speaker_mix({"spk_mix": {"alice": 0.0, "bob": 0.0}}, {}, spk, 0.0116, "cpu", mix_length=5)
    summary = {}
    ids, vals = harness.load_speaker_mix({"spk_mix": {"alice": 1.0, "bob": 3.0}}, summary, spk, 0.0116, "cpu", mix_length=5)
    assert ids.tolist() == [[[0, 1]]] and np.allclose(vals.numpy(), [[[0.25, 0.75]]])
    assert summary == {"spk_mix": "static(alice:1.000|bob:3.000)"}
    ids, vals = harness.load_speaker_mix({"ph_spk_mix": {"alice": "1 0 1", "bob": 1.0}}, {}, spk, 0.0116, "cpu",
                                         mix_mode="token", mix_length=3)
    assert tuple(vals.shape) == (1, 3, 2) and np.allclose(vals.numpy()[0, :, 0], [0.5, 0.0, 0.5])


def test_depth_and_steps_arithmetic():
    """scripts/infer.py:168-198 on the reference fork's acoustic.yaml values."""
    base = dict(use_shallow_diffusion=True, timesteps=1000, K_step=400, K_step_infer=400, diff_speedup=10)
    hp1 = harness.apply_depth_steps(dict(base))
    assert hp1["T_start"] == pytest.approx(0.6) and hp1["T_start_infer"] == pytest.approx(0.6)
    assert hp1["sampling_steps"] == 40 and hp1["time_scale_factor"] == 1000
    hp2 = harness.apply_depth_steps(dict(base), depth=0.3, steps=15)
    assert hp2["K_step_infer"] == 300 and hp2["T_start_infer"] == pytest.approx(0.7)
    assert hp2["diff_speedup"] == round(0.3 / 15 * 300) == 6 and hp2["sampling_steps"] == 15
    with pytest.raises(AssertionError, match="Depth"):
        harness.apply_depth_steps(dict(base), depth=0.5)
    hp3 = harness.apply_depth_steps(dict(use_shallow_diffusion=False, timesteps=1000, K_step=1000, K_step_infer=1000,
                                         pndm_speedup=20), steps=50)
    assert hp3["diff_speedup"] == 20 and hp3["sampling_steps"] == 50


def test_save_wav_roundtrip(tmp_path):
    from scipy.io import wavfile
    wav = np.sin(np.linspace(0, 40, 4410)) * 0.5
    harness.save_wav(wav, tmp_path / "a.wav", 44100)
    sr, data = wavfile.read(tmp_path / "a.wav")
    assert sr == 44100 and data.dtype == np.int16 and np.array_equal(data, (wav * 32767).astype(np.int16))
    table = harness.SimplePhonemeTable(["x", "y"])
    assert len(table) == 5 and table.encode("SP x zh/y AP") == [2, 3, 4, 1]


def test_load_ckpt_follows_the_reference_convention(tmp_path):
    """work_dir/model_ckpt_steps_<N>.ckpt, weights under state_dict with a `model.` prefix, highest N unless asked for
    another, old duplicate embeddings ignored, category checked, non-strict loading drops shape mismatches."""
    import torch.nn as nn
    from diffsinger_amd import harness

    class Net(nn.Module):
        category = 'acoustic'

        def __init__(self):
            super().__init__()
            self.a = nn.Linear(3, 2)

    def ckpt(step, scale, category='acoustic', extra=None):
        sd = {'model.a.weight': torch.full((2, 3), float(scale)), 'model.a.bias': torch.zeros(2),
              'model.fs2.encoder.embed_tokens.weight': torch.ones(4, 4), 'other.x': torch.ones(1)}
        sd.update(extra or {})
        torch.save({'state_dict': sd, 'category': category, 'global_step': step}, tmp_path / f'model_ckpt_steps_{step}.ckpt')

    ckpt(100, 1.0)
    ckpt(2000, 2.0)
    net = Net()
    assert harness.load_ckpt(net, tmp_path).name == 'model_ckpt_steps_2000.ckpt' and float(net.a.weight.detach()[0, 0]) == 2.0
    assert harness.load_ckpt(net, tmp_path, ckpt_steps=100).name == 'model_ckpt_steps_100.ckpt' and float(net.a.weight.detach()[0, 0]) == 1.0
    harness.load_ckpt(net, tmp_path / 'model_ckpt_steps_2000.ckpt')
    assert float(net.a.weight.detach()[0, 0]) == 2.0
    ckpt(3000, 3.0, category='variance')
    with pytest.raises(RuntimeError, match="Category mismatches"):
        harness.load_ckpt(net, tmp_path)
    ckpt(4000, 4.0, extra={'model.a.bias': torch.zeros(5)})
    with pytest.raises(RuntimeError):
        harness.load_ckpt(net, tmp_path)
    harness.load_ckpt(net, tmp_path, strict=False)
    assert float(net.a.weight.detach()[0, 0]) == 4.0
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(AssertionError, match="ckpt not found"):
        harness.load_ckpt(net, empty)


def test_load_vocoder_reads_config_and_generator_weights(tmp_path):
    """`config.json` beside the generator checkpoint, weights under 'generator' (models.py:18-33) - the openvpi release
    layout; the file is read with weights_only=True."""
    from diffsinger_amd import synth
    from diffsinger_amd.vocoder import NsfHifiGAN
    h = dict(synth.NSF_HIFIGAN_DEFAULT, num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4],
             upsample_initial_channel=64, resblock="2", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]])
    with open(tmp_path / "config.json", "w") as f:
        json.dump(h, f)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=9).items()}
    torch.save({"generator": sd}, tmp_path / "model.ckpt")
    hparams.clear()
    hparams.update(mel_base="e")
    voc = harness.load_vocoder(tmp_path / "model.ckpt", device="cpu")
    assert isinstance(voc, NsfHifiGAN) and voc.mel_base == "e"
    got = voc.model.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_load_config_resolves_the_base_config_chain(tmp_path):
    """base.yaml <- model.yaml (relative to the root, as the reference's `configs/...` paths are) <- exp/config.yaml ('.'-
    relative); later files win key by key through nested dicts; a saved complete config loads as is; overrides last."""
    import importlib
    hp_mod = importlib.import_module("diffsinger_amd.hparams")
    (tmp_path / "configs").mkdir()
    (tmp_path / "exp").mkdir()
    (tmp_path / "configs" / "base.yaml").write_text(
        "hidden_size: 256\nuse_pos_embed: true\nbackbone_args:\n  num_layers: 20\n  num_channels: 256\nK_step: 1000\n")
    (tmp_path / "configs" / "model.yaml").write_text(
        "base_config:\n  - configs/base.yaml\nbackbone_args:\n  num_channels: 512\nuse_rope: true\n")
    (tmp_path / "exp" / "local.yaml").write_text("K_step: 400\n")
    (tmp_path / "exp" / "config.yaml").write_text(
        "base_config: [configs/model.yaml, ./local.yaml, configs/base.yaml]\nbackbone_args:\n  dilation_cycle_length: 4\n")
    cfg = hp_mod.load_config("exp/config.yaml", root=tmp_path, overrides={"infer": True, "backbone_args": {"num_layers": 10}})
    assert cfg["hidden_size"] == 256 and cfg["use_rope"] is True and cfg["K_step"] == 400 and cfg["infer"] is True
    assert cfg["backbone_args"] == {"num_layers": 10, "num_channels": 512, "dilation_cycle_length": 4}
    assert hp_mod.hparams["backbone_args"]["num_channels"] == 512          # the process-global dict was refilled
    (tmp_path / "saved.yaml").write_text("hidden_size: 128\nspec_min: [-12.0]\n")
    assert hp_mod.load_config(tmp_path / "saved.yaml", update_global=False) == {"hidden_size": 128, "spec_min": [-12.0]}
    assert hp_mod.hparams["hidden_size"] == 256
    hp_mod.hparams.clear()


@pytest.mark.parametrize("tag", ["multi", "single"])
def test_phoneme_dictionary_vs_reference(tag, tmp_path):
    """harness.PhonemeDictionary against the reference's (G14: utils/phoneme_utils.py on two synthetic pronunciation
    dictionaries): ids, merged groups (overlapping and cross-lingual ones too), encode with and without a language, decode."""
    with open(os.path.join(GOLDEN, "g14_phoneme_dictionary.json"), encoding="utf8") as f:
        g = json.load(f)
    for lang, text in g["dicts"].items():
        (tmp_path / f"{lang}.txt").write_text(text, encoding="utf8")
    c = g["cases"][tag]
    cfg = c["config"]
    d = harness.PhonemeDictionary({l: tmp_path / f"{l}.txt" for l in cfg["langs"]}, extra_phonemes=cfg["extra"],
                                  merged_groups=cfg["merged"])
    assert len(d) == d.vocab_size == c["vocab_size"]
    assert d._phone_to_id == c["phone_to_id"]
    assert [list(p) if isinstance(p, tuple) else p for p in d._id_to_phone] == c["id_to_phone"]
    assert sorted(d.cross_lingual_phonemes) == c["cross_lingual"]
    assert all(d.is_cross_lingual(p) for p in c["cross_lingual"]) and not d.is_cross_lingual("SP")
    if tag == "multi":
        assert d.encode("a sh ir n ja/k EP AP", lang="zh") == c["encode_zh"]
        assert d.encode("a sh i N zh/b", lang="ja") == c["encode_ja"]
        assert [d.decode(range(1, d.vocab_size), lang=l) for l in (None, "zh", "ja")] == c["decode"]
    else:
        assert d.encode("a sh ir n y") == c["encode_zh"]
        assert [d.decode(range(1, d.vocab_size))] == c["decode"]
    got = [d.decode_one(i, scalar=False) for i in range(1, d.vocab_size)]
    assert [list(p) if isinstance(p, tuple) else p for p in got] == c["decode_groups"]
    d.dump(tmp_path / "phonemes.json")
    assert json.load(open(tmp_path / "phonemes.json", encoding="utf8")) == c["phone_to_id"]
    with pytest.raises(ValueError, match="unrecognized language"):
        harness.PhonemeDictionary({"zh": tmp_path / "zh.txt"}, extra_phonemes=["ko/x"])
    with pytest.raises(ValueError, match="not found in phoneme set"):
        harness.PhonemeDictionary({"zh": tmp_path / "zh.txt"}, merged_groups=[["a", "nope"]])


def test_load_phoneme_dictionary_prefers_the_work_directory(tmp_path):
    hparams.clear()
    (tmp_path / "work").mkdir()
    (tmp_path / "work" / "dictionary-zh.txt").write_text("a\ta\nba\tb a\n", encoding="utf8")
    (tmp_path / "ja.txt").write_text("ka\tk a\n", encoding="utf8")
    hparams.update(work_dir=str(tmp_path / "work"), dictionaries={"zh": "missing/zh.txt", "ja": str(tmp_path / "ja.txt")},
                   extra_phonemes=[], merged_phoneme_groups=[])
    d = harness.load_phoneme_dictionary()
    assert d.encode("a b", lang="zh") == [d._phone_to_id["zh/a"], d._phone_to_id["zh/b"]] and "ja/k" in d._phone_to_id
    hparams.update(dictionaries={"ko": "missing/ko.txt"})
    with pytest.raises(FileNotFoundError):
        harness.load_phoneme_dictionary()
    hparams.clear()


def test_project_edits_vs_reference(capsys):
    """`--spk` strings and `--key` shifts (G15: the reference's parse_commandline_spk_mix / trans_key; note NAMES come from
    this package's own note parser there - librosa is absent - and are pinned by the known answers below)."""
    with open(os.path.join(GOLDEN, "g15_infer_utils.json"), encoding="utf8") as f:
        g = json.load(f)
    for mix, want in g["mixes"].items():
        got = harness.parse_commandline_spk_mix(mix)
        assert list(got) == list(want) and all(got[k] == want[k] for k in want), mix
    for mix in g["bad"]:
        with pytest.raises(AssertionError):
            harness.parse_commandline_spk_mix(mix)
    import copy
    for key, want in g["shifted"].items():
        assert harness.trans_key(copy.deepcopy(g["project"]), int(key)) == want
    assert "parts of f0_seq do not exist" in capsys.readouterr().out
    # known answers for the names: twelve semitones to the octave from C-1 = 0, sharps, nearest semitone
    assert [harness.midi_to_note(m) for m in (0, 59, 60, 61, 69, 70.4, 70.6, 127)] == ["C-1", "B3", "C4", "C#4", "A4", "A#4", "B4", "G9"]
    seg = harness.trans_key([dict(note_seq="C4 rest A#3 Db4+20 B3", f0_seq="440.0")], 2)[0]
    assert seg["note_seq"] == "D4 rest C4 D#4 C#4" and seg["f0_seq"] == "493.9"
