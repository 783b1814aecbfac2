"""-m gpu parity of the NSF-HiFiGAN generator (dsd_vocode) against the fixtures
generated from the reference Generator (G10) and the numpy oracle.  Stated fp32 tolerance: 1e-4 of the waveform
range (oracle-vs-reference is <= 5e-5; the source's sin() of accumulated phases is the sensitive part)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import dev, rel_err  # noqa: E402
from oracle import vocoder as ov  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 5e-5        # measured <= 5.7e-6 (profiles/r02_parity.json)
GAIN = 0.7
OVER = {
    "default": dict(),
    "small_rb2": dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], upsample_initial_channel=64,
                      resblock="2", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16),
    "mini_nsf": dict(mini_nsf=True),
    "mini_small": dict(mini_nsf=True, num_mels=32, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                       upsample_initial_channel=128, resblock_kernel_sizes=[3, 7],
                       resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]], hop_size=32),
    "small_sigma": dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], upsample_initial_channel=64,
                        resblock="2", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16,
                        noise_sigma=0.3),
    "mini_sigma": dict(mini_nsf=True, num_mels=32, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                       upsample_initial_channel=128, resblock_kernel_sizes=[3, 7],
                       resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]], hop_size=32, noise_sigma=0.2),
}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"


def build(over, wseed):
    from diffsinger_amd.vocoder import Generator
    h = dict(synth.NSF_HIFIGAN_DEFAULT)
    h.update(over)
    params = synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=wseed, gain=GAIN)
    g = Generator(h)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return g.cuda().eval(), h, params


@pytest.mark.parametrize("tag", sorted(OVER))
def test_vocoder_vs_golden(tag):
    from diffsinger_amd.vocoder import NsfHifiGAN
    g = np.load(os.path.join(GOLDEN, "g10_vocoder.npz"))
    bsz, t_len, wseed, upp = (int(v) for v in g[f"{tag}_meta"])
    gen, h, _ = build(OVER[tag], wseed)
    mel = (synth.synth_normal((bsz, t_len, h["num_mels"]), wseed + 1) * 1.5 - 5.0).astype(np.float32)
    noise = synth.synth_normal((bsz, t_len * upp, 9), wseed + 3)
    pre = synth.synth_normal((bsz, h["upsample_initial_channel"], t_len), wseed + 4)
    wav = NsfHifiGAN(gen).spec2wav_torch(dev(mel), f0=dev(g[f"{tag}_f0"]), rand_ini=dev(g[f"{tag}_rand_ini"]),
                                         noise=dev(noise), pre_noise=dev(pre))
    want = g[f"{tag}_wav"].reshape(-1)
    assert rel_err(wav, want) < TOL
    gen.release_native()


@pytest.mark.parametrize("tag,bsz,t_len", [("small_rb2", 1, 1), ("small_rb2", 3, 130), ("default", 1, 33),
                                           ("mini_small", 1, 1), ("mini_small", 2, 2100),          # 2100 > one scan chunk
                                           ("small_sigma", 2, 70), ("mini_sigma", 1, 5)])
def test_vocoder_vs_oracle_sizes(tag, bsz, t_len):
    gen, h, params = build(OVER[tag], 410)
    upp = int(np.prod(h["upsample_rates"]))
    rng = np.random.Generator(np.random.PCG64(t_len))
    mel = (synth.synth_normal((bsz, h["num_mels"], t_len), 411) * 3.0 - 11.0).astype(np.float32)     # [B, M, T], ln-mel
    f0 = (150.0 * 2.0 ** rng.uniform(-1, 2, (bsz, t_len))).astype(np.float32)
    f0[:, ::7] = 0.0
    rand_ini = rng.random(9).astype(np.float32)
    noise = synth.synth_normal((bsz, t_len * upp, 9), 412)
    pre = synth.synth_normal((bsz, h["upsample_initial_channel"], t_len), 413)
    want = ov.generator_forward(params, h, mel, f0, rand_ini, noise, pre)
    if h.get("mini_nsf") and not h.get("noise_sigma"):      # deterministic source: the draws are not needed at all
        with torch.no_grad():
            assert torch.equal(gen(dev(mel), dev(f0)), gen(dev(mel), dev(f0), rand_ini=dev(rand_ini), noise=dev(noise)))
    with torch.no_grad():
        got = gen(dev(mel), dev(f0), rand_ini=dev(rand_ini), noise=dev(noise), pre_noise=dev(pre))
        got_t = gen(dev(np.ascontiguousarray(mel.transpose(0, 2, 1))).transpose(1, 2), dev(f0), rand_ini=dev(rand_ini),
                    noise=dev(noise), pre_noise=dev(pre))                       # [B, T, M] storage viewed as [B, M, T]
        drawn = gen(dev(mel), dev(f0))                      # device-side draws: same shape, finite, bounded by tanh
    assert tuple(got.shape) == (bsz, 1, t_len * upp)
    assert rel_err(got, want) < TOL
    assert torch.equal(got, got_t)
    assert torch.isfinite(drawn).all() and drawn.abs().max() <= 1.0
    gen.release_native()


def test_vocoder_errors_and_weight_norm_fold():
    from diffsinger_amd.vocoder import Generator
    gen, h, params = build(OVER["small_rb2"], 420)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU path"):
            gen(torch.zeros(1, 32, 4), torch.zeros(1, 4))
        with pytest.raises(ValueError):
            gen(torch.zeros(1, 31, 4).cuda(), torch.zeros(1, 4).cuda())
        assert tuple(gen(torch.zeros(0, 32, 4).cuda(), torch.zeros(0, 4).cuda()).shape) == (0, 1, 64)
    with pytest.raises(RuntimeError, match="rand_ini and noise are required"):      # C-ABI: only mini_nsf may omit the draws
        from diffsinger_amd import _lib
        import ctypes as C
        x = torch.zeros(1, 4, 32).cuda(); f = torch.zeros(1, 4).cuda(); o = torch.zeros(1, 64).cuda()
        hd = gen.native_handle(x.device)
        _lib.check(hd, _lib.lib().dsd_vocode(hd, C.c_void_p(x.data_ptr()), 1, 4, 128, 1, 32, C.c_void_p(f.data_ptr()), None, None,
                                             None, C.c_void_p(o.data_ptr()), None), "dsd_vocode")
    gs, _, _ = build(OVER["small_sigma"], 421)
    with pytest.raises(RuntimeError, match="pre_noise is required"):
        hd = gs.native_handle(x.device)
        _lib.check(hd, _lib.lib().dsd_vocode(hd, C.c_void_p(x.data_ptr()), 1, 4, 128, 1, 32, C.c_void_p(f.data_ptr()),
                                             C.c_void_p(f.data_ptr()), C.c_void_p(f.data_ptr()), None,
                                             C.c_void_p(o.data_ptr()), None), "dsd_vocode")
    gs.release_native()
    # a checkpoint that still carries weight norm: weight_g / weight_v pairs are folded on load
    sd = {}
    for k, v in params.items():
        t = torch.from_numpy(v)
        if k.endswith(".weight") and (k.startswith(("conv_pre", "ups.", "resblocks.", "conv_post"))):
            norm = t.flatten(1).norm(dim=1).reshape(-1, *([1] * (t.dim() - 1)))
            sd[k[:-6] + "weight_g"] = norm * 2.0
            sd[k[:-6] + "weight_v"] = t * 0.5 / 1.0
        else:
            sd[k] = t
    g2 = Generator(h)
    g2.load_state_dict(sd, strict=True)
    for k, v in params.items():
        if k.endswith(".weight") and k.startswith("ups."):
            assert torch.allclose(g2.state_dict()[k], torch.from_numpy(v) * 2.0, rtol=1e-5, atol=1e-7)
    gen.release_native()


def test_ds_harness_segments_to_waveform(tmp_path):
    """`.ds` project -> waveform through AcousticHarness: encoder, aux decoder, shallow loop and vocoder on the HIP
    library, speaker mix / key shift / speed / energy embeddings on, per-segment seeds, overlapping segments
    cross-faded.  Checked: determinism under the seeds, placement and length of the assembled track, the written wav."""
    from scipy.io import wavfile
    from diffsinger_amd import harness
    from diffsinger_amd.hparams import hparams
    from diffsinger_amd.toplevel import DiffSingerAcoustic
    from diffsinger_amd.vocoder import Generator, NsfHifiGAN
    saved = dict(hparams)
    hparams.clear()
    hparams.update(hop_size=512, audio_sample_rate=44100, hidden_size=256, enc_layers=2, enc_ffn_kernel_size=3, ffn_act="gelu",
                   dropout=0.1, num_heads=2, use_pos_embed=True, rel_pos=True, use_rope=True, use_spk_id=True, num_spk=3,
                   use_lang_id=False, num_lang=1, use_energy_embed=True, use_key_shift_embed=True, use_speed_embed=True,
                   augmentation_args=dict(random_pitch_shifting=dict(range=[-5.0, 5.0]),
                                          random_time_stretching=dict(range=[0.5, 2.0])),
                   schedule_type="linear", use_shallow_diffusion=True, diffusion_type="reflow", T_start=0.4, T_start_infer=0.4,
                   time_scale_factor=1000, sampling_algorithm="euler", sampling_steps=8, timesteps=1000, K_step=400,
                   K_step_infer=400, backbone_type="wavenet",
                   backbone_args=dict(num_layers=2, num_channels=64, dilation_cycle_length=2), spec_min=[-12.0], spec_max=[0.0],
                   shallow_diffusion_args=dict(aux_decoder_arch="convnext", val_gt_start=False,
                                               aux_decoder_args=dict(num_channels=64, num_layers=2, kernel_size=7)))
    try:
        table = harness.SimplePhonemeTable(["a", "b", "c", "d", "e"])
        model = DiffSingerAcoustic(len(table), 128)
        sd = dict(model.state_dict())
        sd.update({"fs2." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.fs2_acoustic_param_shapes(
            len(table), enc_layers=2, num_spk=3, variances=("energy",), key_shift=True, speed=True), seed=500).items()})
        sd.update({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.convnext_param_shapes(
            256, 128, num_channels=64, num_layers=2, prefix="aux_decoder.decoder."), seed=501).items()})
        sd.update({"diffusion.velocity_fn." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
            synth.backbone_param_shapes("wavenet", 128, 1, hidden_size=256, num_layers=2, num_channels=64,
                                        dilation_cycle_length=2), seed=502).items()})
        model.load_state_dict(sd, strict=True)
        model = model.cuda().eval()
        vh = dict(synth.NSF_HIFIGAN_DEFAULT, upsample_initial_channel=64)
        gen = Generator(vh)
        gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
            synth.nsf_hifigan_param_shapes(vh), seed=503, gain=GAIN).items()}, strict=True)
        h = harness.AcousticHarness(model, NsfHifiGAN(gen.cuda().eval()), table, spk_map={"alice": 0, "bob": 1, "carol": 2},
                                    device="cuda")
        segs = harness.load_ds(os.path.join(GOLDEN, "g11_segments.ds"))
        out = tmp_path / "proj.wav"
        track = h.run_inference(segs, out_path=out)
        again = h.run_inference(segs)
        assert np.array_equal(track, again)                       # per-segment seeds make the whole project reproducible
        frames = [h.preprocess_input(s)["mel2ph"].shape[1] for s in segs]
        last_start = round(segs[-1]["offset"] * 44100)
        assert track.shape[0] == last_start + frames[-1] * 512
        assert segs[1]["offset"] * 44100 < round(segs[0]["offset"] * 44100) + frames[0] * 512      # really overlapping
        assert np.isfinite(track).all() and np.abs(track).max() <= 1.0 and track.std() > 0
        sr, data = wavfile.read(out)
        assert sr == 44100 and data.shape[0] == track.shape[0]
        mels = h.run_inference(segs, save_mel=True)
        assert [tuple(m["mel"].shape) for m in mels] == [(1, f, 128) for f in frames]
        different = h.run_inference([dict(s, seed=s["seed"] + 1) for s in segs])
        assert not np.array_equal(track, different)
        # the reference's calling convention: out_dir / title / num_runs (ds_acoustic.py:214-271)
        last = h.run_inference(segs, out_dir=tmp_path / "runs", title="song", num_runs=2)
        assert sorted(p.name for p in (tmp_path / "runs").iterdir()) == ["song-000.wav", "song-001.wav"]
        assert np.array_equal(last, track)
        # all segments in ONE ragged batch of the acoustic model (dsd_set_lengths): the same mels, the same waveform
        batched_mels = h.run_inference(segs, save_mel=True, batch_size=len(segs))
        for a, b in zip(mels, batched_mels):
            # not bit-equal: the padded batch takes other tile shapes through the encoder's GEMMs (fp32 rounding, 1e-6),
            # which this random-weight model amplifies; a leak of padded frames would be an O(1) difference
            assert float((a["mel"] - b["mel"]).abs().max()) <= 3e-4 * float(a["mel"].abs().max())
        batched = h.run_inference(segs, batch_size=2)
        assert batched.shape == track.shape and np.abs(batched - track).max() < 2e-2 and \
            np.abs(batched - track).mean() < 1e-3
    finally:
        hparams.clear()
        hparams.update(saved)


def test_example_script_runs_a_saved_experiment(tmp_path):
    """examples/ds_to_wav.py as a user would run it: an experiment directory with config.yaml (base_config chain),
    model_ckpt_steps_<N>.ckpt (`model.` prefix), dictionary.txt and spk_map.json, a vocoder checkpoint with its config.json,
    a `.ds` project -> a wav file; everything found and loaded by the package's own loaders."""
    import subprocess
    import sys
    import yaml
    from scipy.io import wavfile
    from diffsinger_amd.toplevel import DiffSingerAcoustic
    from diffsinger_amd.hparams import hparams
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = tmp_path / "checkpoints" / "exp"
    exp.mkdir(parents=True)
    base = dict(hop_size=512, audio_sample_rate=44100, audio_num_mel_bins=128, hidden_size=256, enc_layers=2, enc_ffn_kernel_size=3,
                ffn_act="gelu", dropout=0.1, num_heads=2, use_pos_embed=True, rel_pos=True, use_rope=True, use_lang_id=False, num_lang=1,
                schedule_type="linear", timesteps=1000, spec_min=[-12.0], spec_max=[0.0], mel_base="e",
                augmentation_args=dict(random_pitch_shifting=dict(range=[-5.0, 5.0]), random_time_stretching=dict(range=[0.5, 2.0])))
    cfg = dict(base_config=["./base.yaml"], use_spk_id=True, num_spk=3, use_energy_embed=True, use_key_shift_embed=True,
               use_speed_embed=True, use_shallow_diffusion=True, diffusion_type="reflow", T_start=0.4, T_start_infer=0.4,
               time_scale_factor=1000, sampling_algorithm="euler", sampling_steps=8, K_step=400, K_step_infer=400,
               backbone_type="wavenet", backbone_args=dict(num_layers=2, num_channels=64, dilation_cycle_length=2),
               shallow_diffusion_args=dict(aux_decoder_arch="convnext", val_gt_start=False,
                                           aux_decoder_args=dict(num_channels=64, num_layers=2, kernel_size=7)))
    (exp / "base.yaml").write_text(yaml.safe_dump(base))
    (exp / "config.yaml").write_text(yaml.safe_dump(cfg))
    (exp / "dictionary.txt").write_text("aa\ta\nbab\tb a b\ncd\tc d\ne\te\n", encoding="utf8")
    (exp / "spk_map.json").write_text('{"alice": 0, "bob": 1, "carol": 2}')
    saved = dict(hparams)
    try:
        hparams.clear()
        hparams.update(base)
        hparams.update({k: v for k, v in cfg.items() if k != "base_config"})
        model = DiffSingerAcoustic(8, 128)                   # AP SP a b c d e + padding
        sd = dict(model.state_dict())
        sd.update({"fs2." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.fs2_acoustic_param_shapes(
            8, enc_layers=2, num_spk=3, variances=("energy",), key_shift=True, speed=True), seed=600).items()})
        sd.update({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.convnext_param_shapes(
            256, 128, num_channels=64, num_layers=2, prefix="aux_decoder.decoder."), seed=601).items()})
        sd.update({"diffusion.velocity_fn." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
            synth.backbone_param_shapes("wavenet", 128, 1, hidden_size=256, num_layers=2, num_channels=64,
                                        dilation_cycle_length=2), seed=602).items()})
    finally:
        hparams.clear()
        hparams.update(saved)
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "category": "acoustic"}, exp / "model_ckpt_steps_100.ckpt")
    torch.save({"state_dict": {"model." + k: v * 0 for k, v in sd.items()}, "category": "acoustic"}, exp / "model_ckpt_steps_20.ckpt")
    voc = tmp_path / "voc"
    voc.mkdir()
    vh = dict(synth.NSF_HIFIGAN_DEFAULT, upsample_initial_channel=64)
    (voc / "config.json").write_text(json.dumps(vh))
    torch.save({"generator": {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
        synth.nsf_hifigan_param_shapes(vh), seed=603, gain=GAIN).items()}}, voc / "model.ckpt")
    out = tmp_path / "song.wav"
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "ds_to_wav.py"), str(exp), os.path.join(GOLDEN, "g11_segments.ds"),
                        str(voc / "model.ckpt"), "-o", str(out), "--batch-size", "3", "--seed", "3"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "model_ckpt_steps_100.ckpt" in r.stdout          # the latest step, not the zeroed older one
    sr, data = wavfile.read(out)
    assert sr == 44100 and data.shape[0] > 44100 and np.abs(data.astype(np.float64)).max() > 0


def test_vocoder_random_draws_in_reference_order():
    """With noise_sigma > 0 the reference draws SineGen's initial phases (models.py:145), its additive noise (:165) and
    only then - after conv_pre - the noise_sigma normals (:272-273).  A seeded run with no injected tensor must consume
    the device generator in exactly that order and leave it where the reference leaves it."""
    gen, h, _ = build(OVER["small_sigma"], 430)
    bsz, t_len = 2, 40
    upp = int(np.prod(h["upsample_rates"]))
    mel = dev((synth.synth_normal((bsz, h["num_mels"], t_len), 431) * 3.0 - 11.0).astype(np.float32))
    f0 = dev(np.full((bsz, t_len), 220.0, np.float32))
    with torch.no_grad():
        torch.manual_seed(1234)
        seeded = gen(mel, f0)
        state_after = torch.cuda.get_rng_state()
        torch.manual_seed(1234)
        rand_ini = torch.rand(9, device="cuda")
        noise = torch.randn((bsz, t_len * upp, 9), device="cuda")
        pre = torch.randn((bsz, h["upsample_initial_channel"], t_len), device="cuda")
        assert torch.equal(torch.cuda.get_rng_state(), state_after)
        injected = gen(mel, f0, rand_ini=rand_ini, noise=noise, pre_noise=pre)
    assert torch.equal(seeded, injected)
    gen.release_native()
