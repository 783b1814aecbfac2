"""The variance `.ds` harness (diffsinger_amd/variance_harness.py) against G13: the reference's own
DiffSingerVarianceInfer run on the synthetic project of tests/variance_cases.py around a stand-in model
(tests/golden/make_golden.py g13_variance_harness).  Checked: every model input of every segment in three prediction
modes (integers exactly, floats to fp32 rounding), which predictors run and with which inputs, and the written project."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import variance_cases as vc
from diffsinger_amd import variance_harness as vh
from diffsinger_amd.harness import SimplePhonemeTable
from diffsinger_amd.hparams import hparams

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = {"auto": set(), "pitch_only": {"pitch"}, "dur_energy": {"dur", "energy"}}


def make(mode):
    hparams.clear()
    hparams.update(vc.HARNESS_HP)
    model = vc.FakeVarianceModel()
    h = vh.VarianceHarness(model, SimplePhonemeTable(vc.HARNESS_PHONES), predictions=MODES[mode], spk_map=vc.HARNESS_SPK,
                           device="cpu")
    return h, model


def test_project_file_is_the_generated_one():
    with open(os.path.join(GOLDEN, "g13_variance_segments.ds"), encoding="utf8") as f:
        assert json.load(f) == json.loads(json.dumps(vc.make_variance_segments()))


@pytest.mark.parametrize("mode", list(MODES))
def test_variance_harness_vs_reference(mode, tmp_path):
    g = np.load(os.path.join(GOLDEN, "g13_variance_harness.npz"))
    h, model = make(mode)
    assert np.allclose(h.smooth_kernel.numpy(), g["smooth_kernel"], rtol=1e-6, atol=0)
    segs = vc.make_variance_segments()
    seen = []
    orig = h.preprocess_input

    def spy(param, idx=0, load_dur=False, load_pitch=False, **kw):
        batch = orig(param, idx=idx, load_dur=load_dur, load_pitch=load_pitch, **kw)
        seen.append((load_dur, load_pitch, batch))
        return batch

    h.preprocess_input = spy
    runs = h.run_inference(copy.deepcopy(segs), out_dir=tmp_path, title=mode, seed=11)
    assert len(seen) == len(segs)
    for i, (ld, lp, batch) in enumerate(seen):
        assert [ld, lp] == g[f"{mode}_seg{i}_load"].tolist()
        keys = {k for k, v in batch.items() if v is not None}
        want_keys = {f[len(f"{mode}_seg{i}_"):] for f in g.files if f.startswith(f"{mode}_seg{i}_")} - {"load"}
        assert keys == want_keys
        for k in keys:
            got, want = batch[k].numpy(), g[f"{mode}_seg{i}_{k}"]
            assert got.shape == want.shape and got.dtype == want.dtype, k
            if got.dtype.kind in "iub":
                assert np.array_equal(got, want), k
            else:
                assert np.allclose(got, want, rtol=2e-6, atol=1e-6), k
    assert model.calls == json.loads(str(g[f"{mode}_calls"]), object_hook=lambda d: {
        k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()})
    with open(os.path.join(GOLDEN, f"g13_out_{mode}.ds"), encoding="utf8") as f:
        want = json.load(f)
    with open(tmp_path / f"{mode}.ds", encoding="utf8") as f:
        written = json.load(f)
    assert written == json.loads(json.dumps(runs[0]))
    assert len(written) == len(want)
    for a, b in zip(written, want):
        assert a.keys() == b.keys()
        for k in a:
            if isinstance(a[k], str) and k in ("ph_dur", "f0_seq", "energy", "breathiness"):
                x, y = np.array(a[k].split(), float), np.array(b[k].split(), float)
                assert x.shape == y.shape and np.allclose(x, y, rtol=0, atol=2e-4 if k == "f0_seq" else 2e-6), k
            else:
                assert a[k] == b[k], k


def test_note_names_and_pitch_conversions_known_answers():
    """librosa is absent from this image: these three follow their definitions; the anchors are A4 = 69 = 440 Hz, twelve
    semitones to the octave, C-1 = 0, a cent = 1/100 semitone."""
    assert vh.note_to_midi("A4") == 69 and vh.note_to_midi("C4") == 60 and vh.note_to_midi("C-1") == 0
    assert vh.note_to_midi("C#4") == 61 == vh.note_to_midi("Db4") == vh.note_to_midi("C♯4") == vh.note_to_midi("D♭4")
    assert vh.note_to_midi("B3") == 59 and vh.note_to_midi("c") == 12 and vh.note_to_midi("F##2") == 43
    assert vh.note_to_midi("F4+30") == pytest.approx(65.3) and vh.note_to_midi("A3-15") == pytest.approx(56.85)
    with pytest.raises(ValueError):
        vh.note_to_midi("H4")
    assert vh.hz_to_midi(440.0) == 69 and vh.hz_to_midi(880.0) == 81 and vh.hz_to_midi(220.0) == 57
    assert vh.midi_to_hz(69) == 440.0 and vh.midi_to_hz(57) == 220.0
    m = np.array([30.5, 60.0, 71.25, 100.0])
    assert np.allclose(vh.hz_to_midi(vh.midi_to_hz(m)), m, rtol=0, atol=1e-9)
    f0 = np.array([0.0, 100.0, 0.0, 0.0, 400.0, 0.0])
    filled, uv = vh.interp_f0(f0.copy())
    assert uv.tolist() == [True, False, True, True, False, True]
    assert np.allclose(filled, [100.0, 100.0, 100.0 * 4 ** (1 / 3), 100.0 * 4 ** (2 / 3), 400.0, 400.0])


def test_smoothing_keeps_length_and_constants():
    h, _ = make("auto")
    x = torch.full((1, 23), 61.5)
    assert torch.allclose(h.smooth(x), x, atol=1e-5)
    step = torch.cat([torch.full((1, 10), 60.0), torch.full((1, 10), 64.0)], dim=1)
    y = h.smooth(step)
    assert y.shape == step.shape and float(y[0, 0]) == pytest.approx(60.0, abs=1e-5) and float(y[0, -1]) == pytest.approx(64.0, abs=1e-5)
    assert (y[0, 1:] >= y[0, :-1] - 1e-6).all()
