"""-m gpu parity of the variance model (BASELINE config 5's callers: dsd_token_encode / dsd_predict_dur /
dsd_cond_assemble + the pitch and multi-variance denoisers) against the G12 fixtures generated from the reference's
own DiffSingerVariance (tests/golden/make_golden.py g12_variance_model) and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

import variance_cases as vc
from diffsinger_amd.hparams import hparams
from gpu_util import dev
from oracle import variance as ovar

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"


def build(tag, g=None):
    from diffsinger_amd.variance import DiffSingerVariance
    hp = vc.case_hparams(tag)
    hparams.clear()
    hparams.update(hp, infer=True)
    c = vc.CASES[tag]
    model = DiffSingerVariance(c["vocab"])
    shapes = vc.sorted_param_shapes(model.named_parameters())
    if g is not None:       # the reference's parameter names and shapes, as stored with the fixture
        ref = [str(s) for s in g[f"{tag}_params"]]
        assert [f"{n}:{'x'.join(map(str, sh))}" for n, sh in shapes.items()] == ref
    params = vc.synth_weights(shapes, c["seed"] + 1)
    res = model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    assert not res.unexpected_keys
    assert not set(res.missing_keys) & set(shapes), "a parameter was not loaded"
    return model.cuda().eval(), hp, params


def to_dev(inp):
    return {k: ({n: dev(a) for n, a in v.items()} if isinstance(v, dict) else dev(v)) for k, v in inp.items()}


def noises(hp, c):
    names = [n for n in ovar.VARIANCE_CHECKLIST if hp.get("predict_" + n)]
    from diffsinger_amd import synth
    seed, out = c["seed"] + 2, {}
    if hp["predict_pitch"]:
        out["pitch_noise"] = synth.synth_normal((c["bsz"], 1, hp["pitch_prediction_args"]["repeat_bins"], c["t_len"]), seed)
        seed += 1
    if names:
        rb = hp["variances_prediction_args"]["total_repeat_bins"] // len(names)
        out["variance_noise"] = synth.synth_normal((c["bsz"], len(names), rb, c["t_len"]), seed)
    return out, names


@pytest.mark.parametrize("tag", list(vc.CASES))
def test_variance_model_vs_golden(tag):
    g = np.load(os.path.join(GOLDEN, "g12_variance_model.npz"))
    model, hp, _ = build(tag, g)
    c = vc.CASES[tag]
    nz, names = noises(hp, c)
    with torch.no_grad():
        dur, pitch, var = model(infer=True, **to_dev(vc.case_inputs(tag)), **{k: dev(v) for k, v in nz.items()})
    if hp["predict_dur"]:
        want = g[f"{tag}_dur"]
        assert np.abs(dur.cpu().numpy() - want).max() < 5e-5 * max(1.0, np.abs(want).max())
    else:
        assert dur is None
    if hp["predict_pitch"]:
        want = g[f"{tag}_pitch"]
        assert tuple(pitch.shape) == want.shape
        assert np.abs(pitch.cpu().numpy() - want).max() < 2e-4 * max(1.0, np.abs(want).max())
    else:
        assert pitch is None
    assert list(var) == names
    for n in names:
        want = g[f"{tag}_{n}"]
        assert np.abs(var[n].cpu().numpy() - want).max() < 2e-4 * np.abs(want).max()
    for m in model.modules():
        if hasattr(m, "release_native"):
            m.release_native()


def test_regulators_and_encoders_vs_golden_and_oracle():
    """The pieces on their own: predicted durations -> aligned durations -> mel2ph exactly as the reference's
    RhythmRegulator / LengthRegulator produce them, encoder output and duration predictor against the oracle on a
    longer, ragged batch than the fixture's."""
    g = np.load(os.path.join(GOLDEN, "g12_variance_model.npz"))
    tag = "word_reflow"
    model, hp, params = build(tag, g)
    inp = vc.case_inputs(tag)
    with torch.no_grad():
        enc, dur = model.fs2(dev(inp["txt_tokens"]), midi=dev(inp["midi"]), ph2word=dev(inp["ph2word"]),
                             word_dur=dev(inp["word_dur"]))
        aligned = model.rr(dur, dev(inp["ph2word"]), dev(inp["word_dur"]))
        mel2ph = model.lr(aligned)
    assert np.array_equal(aligned.cpu().numpy(), g[f"{tag}_dur_aligned"])
    assert np.array_equal(mel2ph.cpu().numpy(), g[f"{tag}_mel2ph"])
    # a longer ragged batch: 3 utterances, 150 phonemes, against the oracle
    rng = np.random.Generator(np.random.PCG64(7))
    bsz, n_ph = 3, 150
    tokens = rng.integers(1, 30, (bsz, n_ph)).astype(np.int64)
    ph2word = np.zeros((bsz, n_ph), np.int64)
    for b, n in enumerate((150, 97, 1)):
        tokens[b, n:] = 0
        ph2word[b, :n] = np.cumsum(rng.random(n) < 0.4) + 1
    midi = rng.integers(30, 90, (bsz, n_ph)).astype(np.int64)
    word_dur = rng.integers(1, 40, (bsz, int(ph2word.max()))).astype(np.int64)
    want_enc, want_dur = ovar.fs2_variance_forward(ovar.sub(params, "fs2."), hp, tokens, midi, ph2word, word_dur=word_dur)
    with torch.no_grad():
        enc, dur = model.fs2(dev(tokens), midi=dev(midi), ph2word=dev(ph2word), word_dur=dev(word_dur))
    assert np.abs(enc.cpu().numpy() - want_enc).max() < 2e-4 * np.abs(want_enc).max()
    assert np.abs(dur.cpu().numpy() - want_dur).max() < 2e-4 * max(1.0, np.abs(want_dur).max())
    model.fs2.release_native()


def test_melody_encoder_vs_oracle():
    tag = "melody_ddim"
    model, hp, params = build(tag)
    inp = vc.case_inputs(tag)
    want = ovar.melody_encoder(ovar.sub(params, "melody_encoder."), hp, inp["note_midi"], inp["note_rest"], inp["note_dur"],
                               glide=inp["note_glide"])
    with torch.no_grad():
        got = model.melody_encoder(dev(inp["note_midi"]), dev(inp["note_rest"]), dev(inp["note_dur"]), glide=dev(inp["note_glide"]))
    keep = (inp["note_midi"] >= 0)[:, :, None]          # padding notes: out_proj's bias only, never gathered
    assert np.abs((got.cpu().numpy() - want) * keep).max() < 2e-4 * np.abs(want).max()
    model.melody_encoder.release_native()


@pytest.mark.parametrize("act", ("relu", "swish", "swiglu"))
def test_token_encoders_other_ffn_activations_vs_oracle(act):
    """`ffn_act` other than gelu (TransformerFFNLayer, common_layers.py:126-136) through the variance model's two token
    encoders - FastSpeech2Variance with its duration predictor and the MelodyEncoder (whose own `melody_encoder_args` may
    override it) - against the oracle, whose activations are pinned by the G8 fixtures `relu` / `swish` / `swiglu`."""
    from diffsinger_amd.variance import DiffSingerVariance
    tag = "melody_ddim"
    hp = vc.case_hparams(tag)
    hp.update(ffn_act=act, predict_dur=True)
    hp["melody_encoder_args"] = dict(hp["melody_encoder_args"], ffn_act="swiglu" if act == "relu" else act)
    hparams.clear()
    hparams.update(hp, infer=True)
    c = vc.CASES[tag]
    model = DiffSingerVariance(c["vocab"])
    shapes = vc.sorted_param_shapes(model.named_parameters())
    params = vc.synth_weights(shapes, c["seed"] + 1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model = model.cuda().eval()
    f1 = params["fs2.encoder.layers.0.op.ffn.ffn_1.weight"].shape[0]
    assert f1 == (8 if act == "swiglu" else 4) * hp["hidden_size"]
    inp = vc.case_inputs(tag)
    want = ovar.melody_encoder(ovar.sub(params, "melody_encoder."), hp, inp["note_midi"], inp["note_rest"], inp["note_dur"],
                               glide=inp["note_glide"])
    with torch.no_grad():
        got = model.melody_encoder(dev(inp["note_midi"]), dev(inp["note_rest"]), dev(inp["note_dur"]), glide=dev(inp["note_glide"]))
    keep = (inp["note_midi"] >= 0)[:, :, None]
    assert np.abs((got.cpu().numpy() - want) * keep).max() < 2e-4 * np.abs(want).max()
    rng = np.random.Generator(np.random.PCG64(17))
    bsz, n_ph = 2, 40
    tokens = rng.integers(1, c["vocab"], (bsz, n_ph)).astype(np.int64)
    tokens[1, 29:] = 0
    ph2word = np.zeros((bsz, n_ph), np.int64)
    for b, n in enumerate((40, 29)):
        ph2word[b, :n] = np.cumsum(rng.random(n) < 0.4) + 1
    midi = rng.integers(30, 90, (bsz, n_ph)).astype(np.int64)
    word_dur = rng.integers(1, 40, (bsz, int(ph2word.max()))).astype(np.int64)
    languages = rng.integers(1, 3, (bsz, n_ph)).astype(np.int64) * (tokens > 0)
    want_enc, want_dur = ovar.fs2_variance_forward(ovar.sub(params, "fs2."), hp, tokens, midi, ph2word, word_dur=word_dur,
                                                   languages=languages)
    with torch.no_grad():
        enc, dur = model.fs2(dev(tokens), midi=dev(midi), ph2word=dev(ph2word), word_dur=dev(word_dur), languages=dev(languages))
    assert np.abs(enc.cpu().numpy() - want_enc).max() < 2e-4 * np.abs(want_enc).max()
    assert np.abs(dur.cpu().numpy() - want_dur).max() < 2e-4 * max(1.0, np.abs(want_dur).max())
    model.fs2.release_native()
    model.melody_encoder.release_native()


def test_cond_assemble_matches_torch_and_rejects_bad_arguments():
    from diffsinger_amd import _lib
    from diffsinger_amd.variance import assemble
    rng = np.random.Generator(np.random.PCG64(3))
    bsz, t_len, hid = 3, 70, 192
    table = dev(rng.standard_normal((11, hid)).astype(np.float32))
    btable = dev(rng.standard_normal((bsz, 9, hid)).astype(np.float32))
    idx = dev(rng.integers(0, 11, (bsz, t_len)))
    bidx = dev(rng.integers(0, 11, (bsz, t_len)))           # 0 -> row -1 (zeros), 10 -> row 9 (out of range: zeros)
    rs = dev(rng.random((bsz, t_len)).astype(np.float32))
    s0 = dev(rng.standard_normal((bsz, t_len)).astype(np.float32))
    v0, v1 = (dev(rng.standard_normal(hid).astype(np.float32)) for _ in range(2))
    got = assemble(bsz, t_len, hid, [(table, idx, 0, 2.5), (btable, bidx, -1, 1.0, rs)], [(s0, v0), (None, v1)], torch.device("cuda", 0))
    padded = torch.nn.functional.pad(btable, [0, 0, 1, 1])
    want = 2.5 * table[idx] + rs[:, :, None] * torch.gather(padded, 1, bidx[:, :, None].expand(-1, -1, hid)) \
        + s0[:, :, None] * v0 + v1
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError, match="no CPU path"):
        assemble(bsz, t_len, hid, [], [(None, v0.cpu())], torch.device("cpu"))
    a = _lib.DsdAssembleArgs()
    assert _lib.lib().dsd_cond_assemble(a, None, None) != 0
    assert b"null argument" in _lib.lib().dsd_last_error(None)


def test_ds_project_completed_on_gpu(tmp_path):
    """`.ds` segments -> VarianceHarness -> DiffSingerVariance on the GPU -> completed `.ds`: durations, f0 and the two
    variance curves are filled in for the segment that lacks them, given curves are kept, a fixed seed reproduces."""
    import copy
    from diffsinger_amd.harness import SimplePhonemeTable
    from diffsinger_amd.variance import DiffSingerVariance
    from diffsinger_amd.variance_harness import VarianceHarness
    hp = vc.case_hparams("word_reflow")
    hp.update(vc.HARNESS_HP, hidden_size=256, use_melody_encoder=True, num_spk=3, infer=True)
    hparams.clear()
    hparams.update(hp)
    table = SimplePhonemeTable(vc.HARNESS_PHONES)
    model = DiffSingerVariance(len(table))
    shapes = vc.sorted_param_shapes(model.named_parameters())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in vc.synth_weights(shapes, 77).items()}, strict=False)
    model = model.cuda().eval()
    h = VarianceHarness(model, table, spk_map=vc.HARNESS_SPK, device="cuda")
    segs = vc.make_variance_segments()
    a = h.run_inference(copy.deepcopy(segs), out_dir=tmp_path, title="song", seed=5)[0]
    b = h.run_inference(copy.deepcopy(segs), seed=5)[0]
    assert a == b
    assert (tmp_path / "song.ds").exists()
    timestep = 512 / 44100
    for src, done in zip(segs, a):
        n_ph = len(src["ph_seq"].split())
        frames = len(done["breathiness"].split())
        assert len(done["ph_dur"].split()) == n_ph
        assert frames == len(done["breathiness"].split())
        if "ph_dur" not in src:     # predicted durations, rescaled word by word: never longer than the notes (a word whose
            total = sum(float(v) for v in done["ph_dur"].split())        # phones were all predicted 0 keeps 0 frames)
            assert 0 <= total <= (frames + 0.5 * n_ph) * timestep
        for key in ("ph_dur", "f0_seq", "energy"):
            if key in src:
                assert done[key] == src[key]          # given curves are not overwritten
        if "f0_seq" not in src:
            f0 = np.array(done["f0_seq"].split(), float)
            assert len(f0) == frames and np.isfinite(f0).all() and (f0 > 20).all() and (f0 < 5000).all()
    # segments with the same predictors and kinds of inputs (here the first and the last) share a launch as a ragged batch
    seen = []
    fmb = h.forward_model_batch
    h.forward_model_batch = lambda samples, noises: (seen.append(len(samples)), fmb(samples, noises))[1]
    c = h.run_inference(copy.deepcopy(segs), seed=5, batch_size=4)[0]
    assert seen == [2]
    for one, bat in zip(a, c):
        assert one.keys() == bat.keys()
        for key in one:
            if key in ("ph_dur", "f0_seq", "energy", "breathiness") and isinstance(one[key], str):
                x, y = np.array(one[key].split(), float), np.array(bat[key].split(), float)
                assert x.shape == y.shape and np.abs(x - y).max() <= 2e-3 * max(1.0, np.abs(x).max()), key
            else:
                assert one[key] == bat[key], key
    for m in model.modules():
        if hasattr(m, "release_native"):
            m.release_native()


def test_variance_model_ragged_batch_equals_alone():
    """Two segments of different length in one zero-padded batch with `lengths` (dsd_set_lengths) against the shorter one
    run alone: phoneme mode, melody encoder, pitch loop (DDIM) - padded tokens, notes and frames all present."""
    tag = "melody_ddim"
    model, hp, _ = build(tag)
    c = vc.CASES[tag]
    inp = vc.case_inputs(tag)
    n_short = 30
    lens = [c["t_len"], n_short]
    for key in ("mel2ph", "mel2note"):
        inp[key][1, n_short:] = 0
    nz, _ = noises(hp, c)
    with torch.no_grad():
        _, pitch, _ = model(infer=True, lengths=lens, **to_dev(inp), pitch_noise=dev(nz["pitch_noise"]))
        alone = {k: (v[1:2, :n_short] if v.shape[-1] == c["t_len"] and k not in ("txt_tokens", "midi", "ph2word", "ph_dur", "languages")
                     else v[1:2]) for k, v in inp.items()}
        _, pitch1, _ = model(infer=True, **to_dev(alone), pitch_noise=dev(nz["pitch_noise"][1:2, :, :, :n_short]))
    a, b = pitch[1:2, :n_short].cpu().numpy(), pitch1.cpu().numpy()
    assert a.shape == b.shape and np.abs(a - b).max() < 5e-5 * max(1.0, np.abs(b).max())
    for m in model.modules():
        if hasattr(m, "release_native"):
            m.release_native()


def test_example_script_completes_a_project_from_a_saved_experiment(tmp_path):
    """examples/ds_variance.py as a user would run it: config.yaml, model_ckpt_steps_<N>.ckpt, dictionary.txt and
    spk_map.json in an experiment directory, a `.ds` project in, the completed `.ds` out."""
    import json
    import subprocess
    import sys
    import yaml
    from diffsinger_amd.variance import DiffSingerVariance
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = tmp_path / "exp"
    exp.mkdir()
    hp = vc.case_hparams("word_reflow")
    hp.update(vc.HARNESS_HP, hidden_size=256, use_melody_encoder=True, num_spk=3)
    (exp / "config.yaml").write_text(yaml.safe_dump(hp))
    (exp / "dictionary.txt").write_text("ab\ta b\ncd\tc d\ne\te\n", encoding="utf8")
    (exp / "spk_map.json").write_text(json.dumps(vc.HARNESS_SPK))
    hparams.clear()
    hparams.update(hp, infer=True)
    model = DiffSingerVariance(8)
    shapes = vc.sorted_param_shapes(model.named_parameters())
    sd = dict(model.state_dict())
    sd.update({k: torch.from_numpy(v) for k, v in vc.synth_weights(shapes, 88).items()})
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "category": "variance"}, exp / "model_ckpt_steps_7.ckpt")
    proj = tmp_path / "song.ds"
    proj.write_text(json.dumps(vc.make_variance_segments()))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "ds_variance.py"), str(exp), str(proj), "-o", str(tmp_path / "out"),
                        "--seed", "5", "--batch-size", "4"], capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-2000:]
    done = json.loads((tmp_path / "out" / "song.ds").read_text(encoding="utf8"))
    assert len(done) == 4 and all("ph_dur" in s and "f0_seq" in s and "energy" in s and "breathiness" in s for s in done)
